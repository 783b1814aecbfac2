#!/bin/bash
# Same-box scan over T (B = 1) of the tile-width choice for short utterances (run ON THE GPU BOX via gpurun):
# default rule / DSD_NARROW=0 (never 16-frame tiles: the row-split pair from T = 1 up) / DSD_NARROW=1 (always).
run() { python bench.py --frames $1 --steps 20 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | \
      python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('T=$1 DSD_NARROW=[$2]', j['ms_per_step'], j['value'])"; }
for T in "$@"; do
  unset DSD_NARROW; run $T default
  export DSD_NARROW=0; run $T 0
  export DSD_NARROW=1; run $T 1
done
