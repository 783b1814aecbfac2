#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: collects the rocprofv3 evidence behind bench.py's numbers
# into gpurun_out/prof_<tag>/ - one kernel-trace pass and separate --pmc passes (never combined with other
# trace domains).  tools/summarize_profile.py then condenses them into profiles/.
#   usage: tools/collect_profiles.sh <tag> "<dominant kernel name prefix>" <launches in the timing pass> <bench args...>
set -e -o pipefail
tag=$1; kname=$2; ntimed=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p "$out"
python3 bench.py "$@" > "$out/bench.json" 2> "$out/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- python3 bench.py "$@" --no-cpu-baseline > "$out/bench_under_rocprof.json" 2> "$out/trace.err"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$out/$c" -o c -- python3 bench.py "$@" --no-cpu-baseline --no-roofline --steps 2 --warmup 1 > /dev/null 2> "$out/$c.err"
  echo "pmc $c done"
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d "$out/MFMA" -o c -- python3 bench.py "$@" --no-cpu-baseline --no-roofline --steps 2 --warmup 1 > /dev/null 2> "$out/MFMA.err"
echo "pmc MFMA done"
# graph-replay launches vs the eager, event-carrying timing pass of the dominant kernel
python3 tools/trace_split.py "$out/trace" "$kname" "$ntimed" > "$out/trace_split.txt"
cat "$out/trace_split.txt"
# keep only what the summary needs (the traces are tens of MB)
find "$out" -name "*kernel_trace.csv" -delete
ls -R "$out" | head -40
