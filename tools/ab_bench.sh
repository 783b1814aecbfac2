#!/bin/bash
# usage: ab_bench.sh <tag>  (run on the GPU box)
tag=$1
for cfg in "wavenet_dpm50 1" "wavenet_dpm50 8" "lynxnet_ddim100 8" "variance_reflow20 8" "wavenet_dpm50 2"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --workload $1 --batch $2 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/ab_${tag}_$1_$2.json 2>/dev/null
  python -c "import sys,json; j=json.loads(open('gpurun_out/ab_${tag}_$1_$2.json').read().strip().splitlines()[-1]); print('$tag', '$1', 'B=$2', j['value'], j['ms_per_step'])"
done
