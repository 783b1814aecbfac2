#!/bin/bash
# Run in the BUILD container: one gpurun call per profiler session (tools/collect_profiles.sh explains why), stopping at
# the first failure - never a retry.  Then condense into profiles/ with tools/summarize_profile.py.
#   usage: tools/collect_all.sh <tag> "<pass list>" <bench args...>
#   e.g.   tools/collect_all.sh r02_lynxnet_ddim100_b8 "FETCH_SIZE WRITE_SIZE MFMA" --workload lynxnet_ddim100 --batch 8 --steps 3 --warmup 1
set -o pipefail
tag=$1; passes=$2; shift 2
GPURUN=/usr/local/graft/bin/gpurun
for p in $passes; do
  echo "=== $tag: pass $p ($(date +%T))"
  $GPURUN --timeout ${DSD_GPURUN_TIMEOUT:-540} -- "DSD_SPLIT_KERNEL='$DSD_SPLIT_KERNEL' DSD_SPLIT_TIMED='$DSD_SPLIT_TIMED' bash tools/collect_profiles.sh $tag $p $*"
  rc=$?
  if [ $rc -ne 0 ]; then
    echo "=== $tag: pass $p FAILED (exit $rc) - stopping, not retrying"
    exit $rc
  fi
done
echo "=== $tag: all passes done"
