#!/bin/bash
# Same-box A/B of two builds (run ON THE GPU BOX via gpurun; hipcc is there): for each quoted flag set, rebuild libdsdenoise
# with DSD_EXTRA_HIPCC_FLAGS and run `bench.py <args>` three times.   usage: bash tools/ab_flags.sh "<flags A>" "<flags B>" -- <bench args>
sets=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do sets+=("$1"); shift; done
shift
for f in "${sets[@]}"; do
  DSD_EXTRA_HIPCC_FLAGS="$f" python -c "
from diffsinger_amd import build_native; build_native.build(force=True, verbose=False)" > /dev/null 2>&1 || { echo "build failed: $f"; exit 1; }
  for i in 1 2 3; do
    python bench.py "$@" --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$f]', j['ms_per_step'], j['value'])"
  done
done
