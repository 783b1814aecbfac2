for f in "-DDSD_LX_PW1_MERGE=0" "-DDSD_LX_PW1_MERGE=1"; do
  DSD_EXTRA_HIPCC_FLAGS="$f" python -c "
from diffsinger_amd import build_native; build_native.build(force=True, verbose=False)" > /dev/null 2>&1
  for cfg in "512 1 1000" "512 8 1000" "1024 1 1000"; do for i in 1 2; do echo -n "[$f] "; python tools/time_lynx.py $cfg 2>/dev/null | tail -1; done; done
done
