#!/usr/bin/env python3
"""Diagnostic (GPU box): repeatability of the wide-row kernels and where they differ from the 64-row pair, per 32-frame tile."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diffsinger_amd import synth
from gpu_util import dev, make_backbone, set_hp
set_hp()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
T = int(sys.argv[3]) if len(sys.argv) > 3 else 256
args = dict(num_layers=L, num_channels=256, dilation_cycle_length=4)
x = dev(synth.synth_normal((B, 1, 128, T), 21)); cond = dev(synth.synth_normal((B, 256, T), 22))
t = dev((np.arange(B) * 211.5 + 3.25).astype(np.float32))
def run(rows, n=4, rows_out=0):
    for k in ("DSD_RS_ROWS", "DSD_FUSED_LAYER", "DSD_RS_ROWS_OUT"): os.environ.pop(k, None)
    if rows:
        os.environ["DSD_RS_ROWS"] = str(rows); os.environ["DSD_FUSED_LAYER"] = "0"
    if rows_out:
        os.environ["DSD_RS_ROWS_OUT"] = str(rows_out)
    net, _ = make_backbone("wavenet", 128, 1, args, 42)
    outs = []
    with torch.no_grad():
        for _ in range(n):
            outs.append(net(x, t, cond).cpu().numpy())
    print("rows", rows, net.stats())
    net.release_native()
    return outs
ref = run(0)[0]
for rows, rows_out in ((128, 0), (256, 0), (256, 128), (128, 256)):
    outs = run(rows, 4, rows_out)
    print("conv rows", rows, "out rows", rows_out or rows)
    for i, o in enumerate(outs):
        d = np.abs(o - ref).max(axis=(1, 2))            # [B, T]
        dt = d.reshape(B, -1, 32).max(-1) if T % 32 == 0 else d
        print(f"rows {rows} run {i}: max |diff vs 64-row| {np.abs(o - ref).max():.3e}  per tile:", np.array2string(dt, precision=1, max_line_width=200))
    print(f"rows {rows}: run-to-run max diff", max(np.abs(o - outs[0]).max() for o in outs))
