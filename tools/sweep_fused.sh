for b in 3 4 5 6 7 8 9 10 11 12 14 16 20 24; do
  for f in 0 1; do
    v=$(DSD_FUSED_LAYER=$f python bench.py --batch $b --steps 4 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_nfe'])")
    echo "B=$b fused=$f $v"
  done
done
