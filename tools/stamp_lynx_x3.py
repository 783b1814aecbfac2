#!/usr/bin/env python3
"""Diagnostic (never shipped, never timed): where wave 0 of an lx_x3_kernel workgroup spends its cycles.  The stamped library is
built HERE (tools/stamp_lynx_x3.py --build, hipcc -DDSD_STAMPS -> tools/diag/lib_stamps.so, which travels with the snapshot);
on the GPU box: python tools/stamp_lynx_x3.py [B] [T]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "diffsinger_amd", "csrc")
EXTRA = [a for a in sys.argv[1:] if a.startswith("-D")]
TAG = next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--tag=")), "")
OUT = os.path.join(ROOT, "tools", "diag", f"lib_stamps{TAG}.so")
if "--build" in sys.argv:
    from diffsinger_amd import build_native
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    objs = []
    for f in build_native.SOURCES:
        o = os.path.join(CSRC, f.replace(".hip", ".o"))
        if f == "lynx_x3.hip":
            o = f"/tmp/lynx_x3_stamps{TAG}.o"
            subprocess.run([build_native.HIPCC] + build_native.FLAGS + build_native.FILE_FLAGS.get(f, []) + ["-DDSD_STAMPS"] + EXTRA + ["-c", os.path.join(CSRC, f), "-o", o], check=True)
        objs.append(o)
    subprocess.run([build_native.HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", OUT] + objs, check=True)
    print(OUT)
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
B = int(args[0]) if args else 8
T = int(args[1]) if len(args) > 1 else 1000
bargs = dict(num_layers=6, num_channels=1024, expansion_factor=2, kernel_size=31, activation="PReLU", strong_cond=True)
net = build_backbone(128, 1, "lynxnet", bargs)
sd = synth.synth_state_dict(synth.backbone_param_shapes("lynxnet", 128, 1, **bargs), 42)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net = net.cuda().eval()
net.set_precision("bf16x3")
x = torch.randn(B, 1, 128, T, device="cuda")
c = torch.randn(B, 256, T, device="cuda")
t = torch.full((B,), 500.0, device="cuda")
with torch.no_grad():
    for _ in range(8):
        net(x, t, c)
torch.cuda.synchronize()
buf = np.zeros((2, 4096, 8), dtype=np.uint64)
assert _lib.lib().dsd_dbg_read_x3_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
for mode, nm in ((0, "pw1"), (1, "pw2")):
    st = buf[mode].astype(np.int64)
    st = st[st[:, 0] > 0]
    if not len(st):
        continue
    life = st[:, 4] - st[:, 0]
    print(f"lx_x3_kernel {nm}: {len(st)} workgroups; mean life {life.mean():.0f} cycles (min {life.min()}, max {life.max()}); "
          f"launch span {st[:, 4].max() - st[:, 0].min()} cycles")
    print(f"    ring fill issued, LayerNorm statistics / bias table            {np.mean(st[:, 1] - st[:, 0]):9.0f}")
    print(f"    phase 0 staged (loads, split, LDS writes), barrier             {np.mean(st[:, 2] - st[:, 1]):9.0f}")
    print(f"    K walk incl. phase switches                                    {np.mean(st[:, 3] - st[:, 2]):9.0f}")
    if (st[:, 6] > 0).all():
        print(f"        of which the LAST phase switch (barrier, stage, barrier)   {np.mean(st[:, 6] - st[:, 5]):9.0f}")
    print(f"    epilogue                                                       {np.mean(st[:, 4] - st[:, 3]):9.0f}")
