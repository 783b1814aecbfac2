set -e
B="python bench.py --workload lynxnet_ddim100 --batch 8 --precision bf16x3 --steps 2 --warmup 1 --no-cpu-baseline"
show() { python -c "
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], j['ms_per_step'], [(k['kernel'][12:], k['avg_launch_us']) for k in j['roofline']['kernels']])
" $1 $2; }
$B > gpurun_out/d0.json 2>/dev/null; show gpurun_out/d0.json base
cp diffsinger_amd/libdsdenoise.so /tmp/keep.so
for d in 1 2; do cp tools/diag/lib_d$d.so diffsinger_amd/libdsdenoise.so; $B > gpurun_out/d$d.json 2>/dev/null; show gpurun_out/d$d.json diag$d; done
cp /tmp/keep.so diffsinger_amd/libdsdenoise.so
