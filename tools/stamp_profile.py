#!/usr/bin/env python3
"""Diagnostic (never shipped, never timed): builds libdsdenoise with -DDSD_STAMPS, runs one backbone
evaluation and prints, per GEMM variant, where a workgroup's wave 0 spends its cycles
(cdna_hip_programming.md section 7, in-kernel stamps).  Usage on the GPU box: python tools/stamp_profile.py [B] [T]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "diffsinger_amd", "csrc")
OUT = os.path.join(ROOT, "diffsinger_amd", "libdsdenoise_stamps.so")
srcs = [os.path.join(CSRC, f) for f in ("gemm.hip", "aux_kernels.hip", "encoder_kernels.hip", "vocoder_kernels.hip", "tconv.hip", "api.hip")]
if not os.path.exists(OUT) or any(os.path.getmtime(s) > os.path.getmtime(OUT) for s in srcs):
    extra = [a for a in sys.argv[1:] if a.startswith("-D")]
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DDSD_STAMPS",
                    "-w", "-shared", "-o", OUT] + extra + srcs, check=True)
if "--build-only" in sys.argv:
    sys.exit(0)

import numpy as np
import torch
from diffsinger_amd import _lib
_lib.LIB_PATH = OUT
from diffsinger_amd import synth
from diffsinger_amd.hparams import hparams
hparams.update(hidden_size=256)
from diffsinger_amd.backbones import build_backbone

args = [a for a in sys.argv[1:] if not a.startswith("-")]
B = int(args[0]) if args else 1
T = int(args[1]) if len(args) > 1 else 1000
bargs = dict(num_layers=20, num_channels=256, dilation_cycle_length=4)
net = build_backbone(128, 1, "wavenet", bargs)
sd = synth.synth_state_dict(synth.backbone_param_shapes("wavenet", 128, 1, **bargs), 42)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net = net.cuda().eval()
x = torch.randn(B, 1, 128, T, device="cuda")
c = torch.randn(B, 256, T, device="cuda")
t = torch.full((B,), 500.0, device="cuda")
with torch.no_grad():
    for _ in range(3):
        net(x, t, c)
torch.cuda.synchronize()
buf = np.zeros((8, 4096, 8), dtype=np.uint64)
rc = _lib.lib().dsd_dbg_read_stamps(buf.ctypes.data_as(C.c_void_p))
assert rc == 0
names = {0: "BIAS_ACT", 1: "GATE (conv)", 2: "RESSKIP (outproj)", 3: "LINCOMB (tail2)"}
labels = ["start->first loads issued", "chunk-0 land + LDS write", "ring prologue drain", "barrier", "K loop (all chunks)", "epilogue"]
for v, nm in names.items():
    st = buf[v].astype(np.int64)
    live = st[:, 0] > 0
    if not live.any():
        continue
    st = st[live]
    d = np.diff(st[:, :7], axis=1)
    span = st[:, 6].max() - st[:, 0].min()
    print(f"{nm}: {live.sum()} WGs; kernel span {span} cyc; start skew {st[:,0].max()-st[:,0].min()} cyc; "
          f"mean WG life {(st[:,6]-st[:,0]).mean():.0f} cyc")
    for lb, m, mx in zip(labels, d.mean(axis=0), d.max(axis=0)):
        print(f"    {lb:28s} mean {m:8.0f}  max {mx:8.0f}")

buf2 = np.zeros((8, 4096, 16), dtype=np.uint64)
assert _lib.lib().dsd_dbg_read_stamps2(buf2.ctypes.data_as(C.c_void_p)) == 0
lab2 = ["chunk0: issue + 1 chunk of MFMA steps", "wait staged loads", "transform + LDS write", "barrier",
        "chunk1: issue + MFMA steps", "wait staged loads", "transform + LDS write", "barrier"]
print("first chunk pair of the pipelined K loop (wave 0):")
for v, nm in names.items():
    st = buf2[v].astype(np.int64)
    live = (st[:, 0] > 0) & (st[:, 8] > 0)
    if not live.any():
        continue
    d = np.diff(st[live][:, :9], axis=1)
    print(f"  {nm}")
    for lb, m, mx in zip(lab2, d.mean(axis=0), d.max(axis=0)):
        print(f"    {lb:40s} mean {m:8.0f}  max {mx:8.0f}")
