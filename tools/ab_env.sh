#!/bin/bash
# Same-box A/B of a runtime switch (run ON THE GPU BOX via gpurun): bash tools/ab_env.sh VAR "v1 v2 ..." -- <bench args>;
# three runs per value ("-" = unset).
var=$1; vals=$2; shift 3
for v in $vals; do
  if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
  for i in 1 2 3; do
    python bench.py "$@" --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$var=$v]', j['ms_per_step'], j['value'])"
  done
done
