#!/bin/bash
# same-box A/B of the row-split weight-ring depth on the other grids the pair serves
for f in "-DDSD_RS_DEPTH=6" "-DDSD_RS_DEPTH=3"; do
  DSD_EXTRA_HIPCC_FLAGS="$f" python -c "
from diffsinger_amd import build_native; build_native.build(force=True, verbose=False)" > /dev/null 2>&1
  for cfg in "--batch 2" "--frames 2048" "--workload variance_reflow20" "--frames 900"; do
    for i in 1 2; do python bench.py $cfg --steps 12 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$f] $cfg', j['ms_per_step'])"; done
  done
done
