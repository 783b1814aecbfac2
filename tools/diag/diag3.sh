cp diffsinger_amd/libdsdenoise.so /tmp/keep.so
for d in $@; do cp tools/diag/lib_v$d.so diffsinger_amd/libdsdenoise.so; echo "== variant $d"; python tools/diag/nd.py 1 256 2>&1 | grep -v amdgpu.ids | head -3; done
cp /tmp/keep.so diffsinger_amd/libdsdenoise.so
