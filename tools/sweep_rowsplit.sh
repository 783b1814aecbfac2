# A/B of the row-split pair (wn_rowsplit.hip) against the two GEMMs of gemm.hip on one box: DSD_ROWSPLIT=0/1
for spec in "1 512" "1 768" "1 900" "1 1000" "1 1024" "1 1100" "1 1536" "1 2048" "2 1000" "3 1000" "4 1000" "5 1000"; do
  set -- $spec
  for f in 0 1; do
    v=$(DSD_ROWSPLIT=$f python bench.py --batch $1 --frames $2 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])")
    echo "B=$1 T=$2 rowsplit=$f $v"
  done
done
