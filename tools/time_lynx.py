#!/usr/bin/env python3
"""ms per backbone evaluation of a LYNXNet (GPU box).  usage: python tools/time_lynx.py C B T"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsinger_amd import synth
from diffsinger_amd.hparams import hparams
hparams.update(hidden_size=256)
from diffsinger_amd.backbones import build_backbone
C, B, T = (int(v) for v in sys.argv[1:4])
bargs = dict(num_layers=6, num_channels=C, expansion_factor=2, kernel_size=31, activation="PReLU", strong_cond=False)
net = build_backbone(128, 1, "lynxnet", bargs)
sd = synth.synth_state_dict(synth.backbone_param_shapes("lynxnet", 128, 1, **bargs), 42)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net = net.cuda().eval()
x = torch.randn(B, 1, 128, T, device="cuda"); c = torch.randn(B, 256, T, device="cuda"); t = torch.full((B,), 500.0, device="cuda")
with torch.no_grad():
    for _ in range(10):
        net(x, t, c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        net(x, t, c)
    torch.cuda.synchronize()
print(f"LYNXNet C={C} B={B} T={T}: {(time.perf_counter() - t0) / 50 * 1e3:.4f} ms per evaluation")
