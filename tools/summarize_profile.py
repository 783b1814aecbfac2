#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/, scratch) into the small committed summaries under
profiles/.  Usage: python tools/summarize_profile.py <tag> <trace_dir> [<fetch_dir> <write_dir> [<mfma_dir>]]
                   [--traffic <workload/B../T..> <kernel name filter, or - for every dsd:: kernel with >= 1 % of the time>] [--bench <dir with bench*.json>]
(`tools/collect_profiles.sh <tag> ...` produces gpurun_out/prof_<tag>/{trace,FETCH_SIZE,WRITE_SIZE,MFMA}.)

FETCH_SIZE / WRITE_SIZE are in KiB per dispatch.  On gfx950 FETCH_SIZE tallies 128-B requests at 64 B, i.e.
it reports half the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM section): the summary stores the
raw value and the x2-corrected byte count.  Both counters sit on the fabric side of L2, so Infinity-Cache
hits are included - they bound HBM traffic from above."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
argv = sys.argv[1:]
traffic_key = traffic_kernel = bench_dir = None
if "--traffic" in argv:
    i = argv.index("--traffic")
    traffic_key, traffic_kernel = argv[i + 1], argv[i + 2]
    del argv[i:i + 3]
if "--bench" in argv:
    i = argv.index("--bench")
    bench_dir = argv[i + 1]
    del argv[i:i + 2]
tag, trace = argv[0], argv[1]
extra = argv[2:]
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)


def one(pattern):
    g = glob.glob(os.path.join(pattern, "*", "*" + "") if os.path.isdir(pattern) else pattern)
    return g


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    assert hits, (d, suffix)
    return hits[0]


stats = find(trace, "kernel_stats.csv")
shutil.copy(stats, os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
summary = {"tag": tag, "kernels": {}}
for r in csv.DictReader(open(stats)):
    if float(r["Percentage"]) < 0.05:
        continue
    summary["kernels"][r["Name"]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                     "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]),
                                     "pct": float(r["Percentage"])}


def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(find(d, "counter_collection.csv"))):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


for d in extra:
    for k, cs in counters(d).items():
        if k not in summary["kernels"]:
            continue
        for c, v in cs.items():
            summary["kernels"][k][c + "_mean"] = sum(v) / len(v)
for k, e in summary["kernels"].items():
    if "FETCH_SIZE_mean" in e and "WRITE_SIZE_mean" in e:
        e["fabric_bytes_per_launch_corrected"] = (2 * e["FETCH_SIZE_mean"] + e["WRITE_SIZE_mean"]) * 1024
    if "SQ_VALU_MFMA_BUSY_CYCLES_mean" in e and "GRBM_GUI_ACTIVE_mean" in e:
        # MFMA busy cycles are summed over 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs (profiled pass)
        e["mfma_busy_frac_profiled"] = e["SQ_VALU_MFMA_BUSY_CYCLES_mean"] / (1024 * e["GRBM_GUI_ACTIVE_mean"] / 8)
out = os.path.join(ROOT, "profiles", f"{tag}_summary.json")
json.dump(summary, open(out, "w"), indent=1)
print(out)

if bench_dir:
    for name in ("bench.json", "bench_under_rocprof.json"):
        src = os.path.join(bench_dir, name)
        if os.path.exists(src):
            line = [l for l in open(src) if l.startswith("{")][-1]
            open(os.path.join(ROOT, "profiles", f"{tag}_{name}"), "w").write(line)
if traffic_key:
    # every dsd:: kernel with at least 1 % of the traced time (traffic_kernel: an optional name filter, "-" = none): bench.py
    # weights them by launches; counters only where the PMC passes saw the kernel
    hits = [k for k, e in summary["kernels"].items() if "dsd::" in k and e["pct"] >= 1.0 and (traffic_kernel in ("-", "") or traffic_kernel in k)]
    assert hits, "no kernel matches"
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
    ent = {"source": f"profiles/{tag}_summary.json (rocprofv3 --kernel-trace --stats averages; --pmc FETCH_SIZE / WRITE_SIZE / MFMA in "
                     "separate eager passes; FETCH x2 gfx950 correction; fabric-side counters, Infinity-Cache hits included)",
           "kernels": {}}
    for k in hits:
        e = summary["kernels"][k]
        ent["kernels"][k] = {
            "rocprof_avg_ns": e["avg_ns"], "calls": e["calls"], "pct": e["pct"],
            "traffic_bytes_per_launch": int(round(e["fabric_bytes_per_launch_corrected"])) if "fabric_bytes_per_launch_corrected" in e else None,
            "fetch_size_kib_raw": e.get("FETCH_SIZE_mean"), "write_size_kib": e.get("WRITE_SIZE_mean"),
            "mfma_busy_frac_profiled": round(e["mfma_busy_frac_profiled"], 4) if "mfma_busy_frac_profiled" in e else None}
    if bench_dir and os.path.exists(os.path.join(bench_dir, "trace_split.txt")):
        shutil.copy(os.path.join(bench_dir, "trace_split.txt"), os.path.join(ROOT, "profiles", f"{tag}_trace_split.txt"))
    tj[traffic_key] = ent
    json.dump(tj, open(tpath, "w"), indent=1)
    print(tpath, traffic_key, len(hits), "kernels")
