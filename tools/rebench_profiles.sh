#!/bin/bash
# Run ON THE GPU BOX after profiles/traffic.json holds the round's entries: the plain bench line of every profiled configuration once
# more, so that its `roofline.frac` / `traffic` come from the committed rocprofv3 data -> gpurun_out/rebench_<tag>.json
# (copy over profiles/<tag>_bench.json).   usage: bash tools/rebench_profiles.sh r03
rt=${1:-r03}
run() { tag=$1; shift; timeout -k 10 300 python3 bench.py "$@" > gpurun_out/rebench_${tag}.json 2> gpurun_out/rebench_${tag}.err; tail -n 1 gpurun_out/rebench_${tag}.json | cut -c1-160; }
run ${rt}_wavenet_dpm50_b1 --workload wavenet_dpm50 --batch 1 --steps 20 --warmup 3
run ${rt}_wavenet_dpm50_b8 --workload wavenet_dpm50 --batch 8 --steps 6 --warmup 2
run ${rt}_wavenet_dpm50_ragged_b8 --workload wavenet_dpm50 --batch 8 --ragged --steps 6 --warmup 2
run ${rt}_lynxnet_ddim100_b8 --workload lynxnet_ddim100 --batch 8 --steps 3 --warmup 1
run ${rt}_variance_reflow20_b8 --workload variance_reflow20 --batch 8 --steps 6 --warmup 2
run ${rt}_wavenet_dpm50_bf16x3_b8 --workload wavenet_dpm50 --batch 8 --precision bf16x3 --steps 6 --warmup 2
run ${rt}_lynxnet_ddim100_bf16x3_b8 --workload lynxnet_ddim100 --batch 8 --precision bf16x3 --steps 3 --warmup 1
run ${rt}_wavenet_dpm50_b4 --workload wavenet_dpm50 --batch 4 --steps 8 --warmup 2
run ${rt}_acoustic_default_b1 --workload acoustic_default --batch 1 --steps 10 --warmup 2
