#!/usr/bin/env python3
"""Batch and frame sweep of the headline workload with the current build -> gpurun_out/<tag>_sweep.json (GPU box only;
copy it to profiles/).  Each point is one `bench.py` run (its JSON line, minus the CPU baseline); see DESIGN.md section 6.
Usage: python tools/sweep.py [tag, default r03]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
points = [("wavenet_dpm50", b, 1000) for b in (1, 2, 3, 4, 5, 6, 8, 9, 12, 16, 24)] + [("wavenet_dpm50_ragged", 8, 0), ("wavenet_dpm50_ragged8", 8, 0)] + \
         [("wavenet_dpm50", 1, t) for t in (128, 256, 512, 768, 1100, 1536, 2048, 4096)] + \
         [("lynxnet_ddim100", b, 1000) for b in (1, 8)] + [("variance_reflow20", b, 1000) for b in (1, 8)] + \
         [("acoustic_default", 1, 1000), ("acoustic_wav", 1, 1000)] + \
         [("wavenet_dpm50_bf16x3", b, 1000) for b in (8, 16, 24)] + [("variance_reflow20_bf16x3", 8, 1000)] + \
         [("lynxnet_ddim100_bf16x3", b, 1000) for b in (1, 2, 8)] + [("acoustic_default_bf16x3", 1, 1000)]
out = []
for wl, b, t in points:
    steps = 10 if b * t <= 4000 else 4
    base = wl.replace("_ragged8", "").replace("_ragged", "").replace("_bf16x3", "")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", base, "--batch", str(b), "--steps", str(steps),
           "--warmup", "2", "--no-cpu-baseline"]
    if wl.endswith("_ragged"):          # BASELINE config 4 as specified: rank 0's shard of the 64-utterance, 8-rank partition
        cmd += ["--ragged"]
    elif wl.endswith("_ragged8"):       # round 2's stand-in: 8 lengths drawn (mean 736 frames)
        cmd += ["--ragged", "--ragged-world", "0"]
    else:
        cmd += ["--frames", str(t)]
    if wl.endswith("_bf16x3"):
        cmd += ["--precision", "bf16x3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    try:
        j = json.loads(line)
    except Exception:
        print(f"{wl} B={b} T={t}: FAILED\n{r.stderr[-400:]}", flush=True)
        continue
    rf = j.get("roofline", {})
    row = dict(workload=wl, batch=b, frames=t, value=j["value"], ms_per_step=j["ms_per_step"], rtf=j.get("rtf"),
               path_tflops=j.get("path_tflops"), path_mfma_frac=j.get("path_mfma_frac"), kernel_avg_launch_us=rf.get("avg_launch_us"),
               kernel_frac=rf.get("frac_events"), kernel=(rf.get("kernel") or "")[:40], frames_mean=j["config"].get("frames"),
               plan=rf.get("plan"))
    out.append(row)
    print(json.dumps(row), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", f"{tag}_sweep.json"), "w") as f:
    json.dump(out, f, indent=1)
