#!/bin/bash
# A/B of store cache policies of the row-split pair at B = 1: builds libdsdenoise with each combination ON THE GPU BOX
# (hipcc is there too) and runs the headline bench 3 times each.  usage: bash tools/ab_store.sh
for combo in "16 16" "16 0" "0 16" "0 0"; do
  set -- $combo
  DSD_EXTRA_HIPCC_FLAGS="-DDSD_ST_AUX=$1 -DDSD_ST_AUX_Z=$2" python -c "
from diffsinger_amd import build_native; build_native.build(force=True, verbose=False)" > /dev/null 2>&1
  for i in 1 2 3; do
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('x/skip aux=$1 z aux=$2', j['ms_per_step'])"
  done
done
