#!/bin/bash
# Batch sweep of the headline workload (T = 1000) with the layer planner on (default) and off (DSD_WN_PLAN=0: one launch shape per
# layer, round 2's rule), on ONE box: frames/s per denoise step and ms per NFE.  -> profiles/r03_plan_sweep.txt
#   usage (GPU box): bash tools/sweep_plan.sh "8 9 10 ..." [extra bench args]
bs=${1:-"8 9 10 11 12 13 16 17 18 20 21 24"}; shift
for b in $bs; do
  for plan in 1 0; do
    v=$(DSD_WN_PLAN=$plan python bench.py --batch $b --steps 4 --warmup 2 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_nfe'])")
    echo "B=$b plan=$plan $v"
  done
done
