#!/usr/bin/env python3
"""Diagnostic (never shipped, never timed): builds libdsdenoise with -DDSD_STAMPS, runs LYNXNet evaluations on a batched grid
and prints where wave 0 of an lx_pw1p_kernel workgroup spends its cycles.  Usage on the GPU box: python tools/stamp_lynx.py [B] [T]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "diffsinger_amd", "csrc")
OUT = os.path.join(ROOT, "diffsinger_amd", "libdsdenoise_stamps.so")
from diffsinger_amd import build_native
srcs = [os.path.join(CSRC, f) for f in build_native.SOURCES]
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DDSD_STAMPS", "-w", "-shared", "-o", OUT] + srcs, check=True)
import numpy as np
import torch
from diffsinger_amd import _lib
_lib.LIB_PATH = OUT
from diffsinger_amd import synth
from diffsinger_amd.hparams import hparams
hparams.update(hidden_size=256)
from diffsinger_amd.backbones import build_backbone

args = [a for a in sys.argv[1:] if not a.startswith("-")]
B = int(args[0]) if args else 8
T = int(args[1]) if len(args) > 1 else 1000
bargs = dict(num_layers=6, num_channels=1024, expansion_factor=2, kernel_size=31, activation="PReLU", strong_cond=True)
net = build_backbone(128, 1, "lynxnet", bargs)
sd = synth.synth_state_dict(synth.backbone_param_shapes("lynxnet", 128, 1, **bargs), 42)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net = net.cuda().eval()
x = torch.randn(B, 1, 128, T, device="cuda")
c = torch.randn(B, 256, T, device="cuda")
t = torch.full((B,), 500.0, device="cuda")
with torch.no_grad():
    for _ in range(8):
        net(x, t, c)
torch.cuda.synchronize()
buf = np.zeros((4096, 8), dtype=np.uint64)
assert _lib.lib().dsd_dbg_read_lx_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
st = buf.astype(np.int64)
st = st[st[:, 0] > 0]
life = st[:, 4] - st[:, 0]
print(f"lx_pw1p_kernel: {len(st)} workgroups, 8 row tiles each; mean life {life.mean():.0f} cycles (min {life.min()}, max {life.max()})")
print(f"    prologue (statistics merge, 128 KiB tile staged, barrier) {np.mean(st[:, 1] - st[:, 0]):10.0f}")
print(f"    K walks, sum of 8 (MFMA floor 8 x 131,072)              {st[:, 2].mean():10.0f}   per row tile {st[:, 2].mean() / 8:8.0f}")
print(f"    SwiGLU epilogues, sum of 8                               {st[:, 3].mean():10.0f}   per row tile {st[:, 3].mean() / 8:8.0f}")
