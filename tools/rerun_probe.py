"""Diagnostic (GPU box): one LYNXNet evaluation in split-bf16 mode run four times on the same inputs - prints which kernels ran and,\nper 16-frame group, how far the runs differ (0 everywhere = repeatable).  Usage: python tools/rerun_probe.py [DSD_X3_WIDE] [T]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
from diffsinger_amd import synth
from gpu_util import dev, make_backbone, set_hp
os.environ["DSD_LYNX_RESIDENT"] = "1"; os.environ["DSD_X3_WIDE"] = sys.argv[1] if len(sys.argv) > 1 else "1"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 211
set_hp()
args = dict(num_layers=2, num_channels=1024, expansion_factor=2, kernel_size=31, activation="PReLU", strong_cond=True)
net, _ = make_backbone("lynxnet", 128, 1, args, 42)
net.set_precision("bf16x3")
x = dev(synth.synth_normal((2, 1, 128, T), 1)); c = dev(synth.synth_normal((2, 256, T), 2)); t = dev(np.array([100.0, 700.0], np.float32))
outs = []
with torch.no_grad():
    for i in range(4):
        outs.append(net(x, t, c).clone())
torch.cuda.synchronize()
net.kernel_timing(True)
with torch.no_grad(): net(x, t, c)
print([k["name"] for k in net.kernel_classes()])
for i in range(1, 4):
    d = (outs[i] - outs[0]).abs()[:, 0]          # [B, M, T]
    print(f"run {i} vs 0: max {d.max().item():.3e}; per item {[f'{v:.1e}' for v in d.amax(dim=(1,2)).tolist()]}")
    per_t = d.amax(dim=(0, 1)).cpu().numpy()
    print("   per 16-frame group:", " ".join(f"{per_t[j:j+16].max():.0e}" for j in range(0, T, 16)))
