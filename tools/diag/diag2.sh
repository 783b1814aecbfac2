cp diffsinger_amd/libdsdenoise.so /tmp/keep.so
for d in 1 2; do cp tools/diag/lib_v$d.so diffsinger_amd/libdsdenoise.so; echo "== variant $d"; timeout -k 10 200 python -m pytest tests/test_gpu_bf16x3.py -x -q -k "wide_tiles" 2>&1 | tail -4; done
cp /tmp/keep.so diffsinger_amd/libdsdenoise.so
