#!/usr/bin/env python3
"""Diagnostic (never shipped, never timed): builds libdsdenoise with -DDSD_STAMPS, runs backbone evaluations with the
fused WaveNet layer kernel (wn_layer.hip) and prints where wave 0 of a workgroup spends its cycles, plus the in-kernel
clock (s_memtime / s_memrealtime).  Usage on the GPU box: python tools/stamp_wn_layer.py [B] [T] [-DNAME ...]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "diffsinger_amd", "csrc")
OUT = os.path.join(ROOT, "diffsinger_amd", "libdsdenoise_stamps.so")
from diffsinger_amd.build_native import SOURCES  # noqa: E402
srcs = [os.path.join(CSRC, f) for f in SOURCES]
extra = [a for a in sys.argv[1:] if a.startswith("-D")]
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DDSD_STAMPS",
                "-w", "-shared", "-o", OUT] + extra + srcs, check=True)
if "--build-only" in sys.argv:
    sys.exit(0)
ROWSPLIT = "--rowsplit" in sys.argv           # the B = 1 pair of wn_rowsplit.hip instead of the fused kernel
if not ROWSPLIT:
    os.environ["DSD_FUSED_LAYER"] = "1"

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
CH = int(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--channels=")), 256))      # --channels=192 --cycle=5
CYC = int(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--cycle=")), 4))
bargs = dict(num_layers=CYC * 2, num_channels=CH, dilation_cycle_length=CYC)       # the LAST layer (the one stamped) has the widest dilation
net = build_backbone(128, 1, "wavenet", bargs)
sd = synth.synth_state_dict(synth.backbone_param_shapes("wavenet", 128, 1, **bargs), 42)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net = net.cuda().eval()
x = torch.randn(B, 1, 128, T, device="cuda")
c = torch.randn(B, 256, T, device="cuda")
t = torch.full((B,), 500.0, device="cuda")
with torch.no_grad():
    for _ in range(20):         # the stamps of the LAST layer launch survive; the chip is warm by then
        net(x, t, c)
torch.cuda.synchronize()
if "--edge" in sys.argv:        # whole sampler loops: every evaluation but the last fuses the next one's input projection
    from diffsinger_amd.diffusion import GaussianDiffusion
    hparams.update(schedule_type="linear", use_shallow_diffusion=False, diff_accelerator="ddim", diff_speedup=500, K_step_infer=1000)
    dd = GaussianDiffusion(128, 1, backbone_type="wavenet", backbone_args=bargs, spec_min=[-12.0], spec_max=[0.0])
    dd.denoise_fn.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    dd = dd.cuda().eval()
    dd.use_graph = False
    cond = torch.randn(B, T, 256, device="cuda")
    dd(cond, infer=True)
    torch.cuda.synchronize()
if "--edge" in sys.argv:
    # wn_edge_kernel, the launch around the layers (batched grids): the last launch of each kind survives
    buf3 = np.zeros((2, 4096, 16), dtype=np.uint64)
    assert _lib.lib().dsd_dbg_read_edge_stamps(buf3.ctypes.data_as(C.c_void_p)) == 0
    labels = ["prologue: loads issued, solver sums formed -> LDS", "skip tile / sqrt(L) -> LDS, barrier", "product 1 (C x C)",
              "relu + bias -> h tile, barrier", "product 2 (F*M x C)", "outputs = sums + c eps, x' -> LDS, row stores",
              "barrier", "product 3 (C x F*M)", "relu + bias -> tile -> row stores"]
    for kind, n in ((1, 10), (0, 7)):
        st = buf3[kind].astype(np.int64)
        st = st[st[:, 0] > 0]
        if not len(st):
            continue
        d = np.diff(st[:, :n], axis=1)
        life = st[:, n - 1] - st[:, 0]
        print(f"wn_edge_kernel {'with' if kind else 'without'} the next input projection: {len(st)} workgroups; mean life {life.mean():.0f} cycles")
        for lb, m, mn, mx in zip(labels, d.mean(axis=0), d.min(axis=0), d.max(axis=0)):
            print(f"    {lb:58s} mean {m:8.0f}  min {mn:8.0f}  max {mx:8.0f}")
    sys.exit(0)
if ROWSPLIT:
    buf2 = np.zeros((2, 4096, 40), dtype=np.uint64)
    assert _lib.lib().dsd_dbg_read_rs_stamps(buf2.ctypes.data_as(C.c_void_p)) == 0
    names = [("conv + FiLM + gate", ["tile decode -> loads issued (x, FiLM, 5 weight blocks)",
                                     "x tile landed, FiLM + mask, LDS write, 2 barriers", "K walk (48 x 8 MFMA per wave)",
                                     "LDS transpose, gate, z store", "-"]),
             ("out-proj + residual / skip", ["tile decode -> loads issued (z, 5 weight blocks, bias)",
                                             "z tile landed, LDS write, barrier", "K walk (16 x 8 MFMA per wave)",
                                             "LDS transpose, residual / skip, stores", "-"])]
    for k, (nm, labels) in enumerate(names):
        st = buf2[k].astype(np.int64)
        st = st[st[:, 0] > 0]
        d = np.diff(st[:, :6], axis=1)
        life = st[:, 5] - st[:, 0]
        real = (st[:, 9] - st[:, 8]) * 10e-9
        print(f"{nm}: {len(st)} workgroups; mean life {life.mean():.0f} cyc; in-kernel clock "
              f"{np.median(life / np.maximum(real, 1e-12) / 1e9):.3f} GHz -> {np.median(real) * 1e6:.2f} us per workgroup")
        for lb, m, mn, mx in zip(labels, d.mean(axis=0), d.min(axis=0), d.max(axis=0)):
            print(f"    {lb:62s} mean {m:8.0f}  min {mn:8.0f}  max {mx:8.0f}")
        if k == 0 and (st[:, 10:34] > 0).all():          # the conv walk step by step (wave 0; the stamp itself costs an lgkmcnt(0))
            w = np.diff(np.concatenate([st[:, 2:3], st[:, 10:34]], axis=1), axis=1).mean(axis=0)
            print("    conv walk, cycles per local step (256 = the MFMAs of two waves): " + " ".join(f"{v:.0f}" for v in w))
        if k == 0 and (st[:, 7] > 0).all():
            e = st[:, 7] - st[:, 6]
            print(f"    kernel entry -> every argument in SGPRs (one batch of scalar loads): mean {e.mean():6.0f}  min {e.min():6.0f}  max {e.max():6.0f};"
                  f"  then {np.mean(st[:, 0] - st[:, 7]):6.0f} until the first phase stamp")
    sys.exit(0)
buf = np.zeros((4096, 10), dtype=np.uint64)
rc = _lib.lib().dsd_dbg_read_wn_stamps(buf.ctypes.data_as(C.c_void_p))
assert rc == 0
st = buf.astype(np.int64)
live = st[:, 0] > 0
st = st[live]
labels = ["tile decode -> loads issued (x, FiLM, cond-proj, weights)", "x tile landed, FiLM + mask, LDS write, barrier",
          "GEMM 1 (conv, 3072 MFMA per wave)", "gate + barrier + z -> LDS + barrier", "GEMM 2 (out-proj, 1024 MFMA per wave)",
          "epilogue (LDS transpose, residual / skip, stores)"]
d = np.diff(st[:, :7], axis=1)
life = st[:, 6] - st[:, 0]
real = (st[:, 9] - st[:, 8]) * 10e-9          # s_memrealtime ticks at 100 MHz
clk = life / np.maximum(real, 1e-12) / 1e9
print(f"{live.sum()} workgroups; kernel span {st[:, 6].max() - st[:, 0].min()} cyc; start skew {st[:, 0].max() - st[:, 0].min()} cyc; "
      f"mean life {life.mean():.0f} cyc (min {life.min()}, max {life.max()}); in-kernel clock {np.median(clk):.3f} GHz "
      f"-> {np.median(real) * 1e6:.1f} us per workgroup")
for lb, m, mn, mx in zip(labels, d.mean(axis=0), d.min(axis=0), d.max(axis=0)):
    print(f"    {lb:62s} mean {m:8.0f}  min {mn:8.0f}  max {mx:8.0f}")
