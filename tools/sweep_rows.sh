#!/bin/bash
# Rows per workgroup of the two-launch path (64: wn_rowsplit.hip / gemm.hip, 128 / 256: wn_rows.hip) over small batches, one box.
#   usage (GPU box): bash tools/sweep_rows.sh "2 3 4 5 6" [extra bench args]
bs=${1:-"2 3 4 5 6"}; shift
for b in $bs; do
  for rows in 0 128 256; do
    if [ $rows = 0 ]; then env_s="DSD_WN_PLAN=0"; else env_s="DSD_RS_ROWS=$rows DSD_FUSED_LAYER=0"; fi
    v=$(env $env_s python bench.py --batch $b --steps 6 --warmup 2 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_nfe'])")
    echo "B=$b rows=$rows $v"
  done
  v=$(python bench.py --batch $b --steps 6 --warmup 2 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_nfe'])")
  echo "B=$b default $v"
done
