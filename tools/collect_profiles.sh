#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: ONE rocprofv3 session (or the plain bench run) per invocation,
# under its own timeout, into gpurun_out/prof_<tag>/.  tools/collect_all.sh (run in the build container) issues one
# gpurun call per pass and stops at the first failure; tools/summarize_profile.py then condenses the passes into
# profiles/.
#
# Why one session per call: in round 1 the second of two back-to-back --pmc sessions of ONE call stalled inside
# rocprofv3 / HSA start-up of the profiled child ("HSA version 8.20.0 initialized" was its last line, 0.7 s after the
# previous counter session had finished; the child never reached HIP initialisation - its libdrm "amdgpu.ids" warning,
# the first line of every other log, is missing).  A fresh box per session cannot inherit a counter session that is
# still being torn down, every session has a bound of its own (exit code 124 on expiry, no retry), and the PMC passes
# run the loop eagerly (--no-graph): counter collection serialises dispatches anyway, and it keeps hipGraph replay
# out of the profiler's interception path.
#
#   usage: tools/collect_profiles.sh <tag> <pass> <bench args...>
#   pass : both (= bench, then trace) | bench | trace | FETCH_SIZE | WRITE_SIZE | MFMA | split "<kernel name prefix>" <launches in the timing pass>
set -e -o pipefail
tag=$1; pass=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p "$out"
att=$(date +%H%M%S)        # every attempt keeps its own stderr (<pass>.err.<time>): a re-run never overwrites the evidence of a failure
LIMIT=${DSD_PROF_TIMEOUT:-420}
PMC_ARGS="--no-cpu-baseline --no-roofline --no-graph --steps 2 --warmup 1"
case "$pass" in
  both)      # the plain run, then the kernel trace (neither collects counters)
    timeout -k 10 "$LIMIT" python3 bench.py "$@" > "$out/bench.json" 2> "$out/bench.err.$att"
    tail -n 1 "$out/bench.json" | cut -c1-300
    timeout -k 10 "$LIMIT" rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- \
        python3 bench.py "$@" --no-cpu-baseline > "$out/bench_under_rocprof.json" 2> "$out/trace.err.$att"
    find "$out" -name "*kernel_trace.csv" -delete ;;
  bench)
    timeout -k 10 "$LIMIT" python3 bench.py "$@" > "$out/bench.json" 2> "$out/bench.err.$att"
    tail -n 1 "$out/bench.json" ;;
  trace)
    # rocprofv3 ... -- python3 bench.py: nothing between "--" and the program (no env / bash -c hop)
    timeout -k 10 "$LIMIT" rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- \
        python3 bench.py "$@" --no-cpu-baseline > "$out/bench_under_rocprof.json" 2> "$out/trace.err.$att"
    if [ -n "$DSD_SPLIT_KERNEL" ]; then
      python3 tools/trace_split.py "$out/trace" "$DSD_SPLIT_KERNEL" "${DSD_SPLIT_TIMED:-0}" > "$out/trace_split.txt"
      cat "$out/trace_split.txt"
    fi
    # keep only what the summary needs (the traces are tens of MB)
    find "$out" -name "*kernel_trace.csv" -delete ;;
  FETCH_SIZE|WRITE_SIZE)
    timeout -k 10 "$LIMIT" rocprofv3 --pmc "$pass" --output-format csv -d "$out/$pass" -o c -- \
        python3 bench.py "$@" $PMC_ARGS > "$out/$pass.out" 2> "$out/$pass.err.$att" ;;
  MFMA)
    timeout -k 10 "$LIMIT" rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv \
        -d "$out/MFMA" -o c -- python3 bench.py "$@" $PMC_ARGS > "$out/MFMA.out" 2> "$out/MFMA.err.$att" ;;
  *)
    echo "unknown pass $pass" >&2; exit 2 ;;
esac
echo "pass $pass of $tag done"
ls -R "$out" | head -30
