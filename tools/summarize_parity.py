#!/usr/bin/env python3
"""Condense gpurun_out/parity.jsonl (every comparison of a `pytest -m gpu` run: tests/gpu_util.py) into
profiles/<tag>_parity.json: per test the worst max-abs / max-ref and rms / rms-ref over its comparisons, next to the
tolerance it was held to.  Usage: python tools/summarize_parity.py <tag> [path]"""
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "parity.jsonl")
acc = collections.OrderedDict()
for line in open(path):
    r = json.loads(line)
    e = acc.setdefault(r["test"], {"comparisons": 0, "max_rel": 0.0, "rms_rel": 0.0, "tol": None, "rms_tol": None})
    e["comparisons"] += 1
    e["max_rel"] = max(e["max_rel"], r["max_rel"])
    e["rms_rel"] = max(e["rms_rel"], r["rms_rel"])
    if r.get("tol") is not None:
        e["tol"] = r["tol"] if e["tol"] is None else min(e["tol"], r["tol"])
        e["rms_tol"] = r["rms_tol"] if e["rms_tol"] is None else min(e["rms_tol"], r["rms_tol"])
for e in acc.values():
    e["max_rel"] = float(f"{e['max_rel']:.3e}")
    e["rms_rel"] = float(f"{e['rms_rel']:.3e}")
out = os.path.join(ROOT, "profiles", f"{tag}_parity.json")
json.dump({"source": "pytest tests -m gpu on one MI355X (tests/gpu_util.py records every comparison)",
           "metric": "max_rel = max|a-b| / max|ref|, rms_rel = rms(a-b) / rms(ref); worst over the test's comparisons",
           "tests": acc}, open(out, "w"), indent=1)
worst = sorted(acc.items(), key=lambda kv: -kv[1]["max_rel"])[:12]
print(out, len(acc), "tests")
for k, e in worst:
    print(f"  {e['max_rel']:.2e} {e['rms_rel']:.2e}  tol {e['tol']}  {k}")
