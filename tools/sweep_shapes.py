#!/usr/bin/env python3
"""Shape sweep of the WaveNet backbone: the nine configurations of tests/test_gpu_config_sweep.py (wn0 ... wn8: channel
counts 32 ... 512, dilation cycles up to 6, n_feats > 1, hidden sizes other than 256), each at B = 8, T = 1000, one
evaluation timed with HIP events over a replayed hipGraph -> gpurun_out/<tag>_shapes.json (GPU box only; copy to profiles/).
`frac` = algorithmic flops of one evaluation (in-projection, L x (dilated conv + 1x1 out), skip and output projections; the
hoisted conditioner projection is not in the loop) / time / the fp32-MFMA peak (157.3 TF, MI355X_MICROARCH.md).
Usage: python tools/sweep_shapes.py [tag, default r03] [batch, default 8] [frames, default 1000]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from diffsinger_amd import synth  # noqa: E402
from diffsinger_amd.backbones import build_backbone  # noqa: E402
from gpu_util import set_hp  # noqa: E402

PEAK_TF = 157.3
SHAPES = [
    # (in_dims, n_feats, C, L, cycle, hidden)  - tests/test_gpu_config_sweep.py WAVENET_SWEEP, the grid replaced by B x T
    (128, 1, 256, 6, 6, 256), (128, 1, 256, 5, 5, 256), (128, 1, 256, 5, 5, 256), (64, 1, 256, 4, 4, 256), (24, 2, 192, 4, 4, 256),
    (20, 3, 96, 3, 2, 128), (80, 1, 128, 3, 3, 192), (128, 1, 512, 2, 2, 256), (8, 1, 32, 2, 1, 64),
    # widths seen in the wild beside the fork's 256 / 192
    (128, 1, 384, 20, 4, 256), (128, 1, 512, 20, 4, 256), (128, 1, 128, 20, 4, 256),
]


def flops_per_eval(in_dims, n_feats, c, nl, bsz, t_len):
    fm = in_dims * n_feats
    per_frame = 2 * fm * c + nl * (2 * 3 * c * 2 * c + 2 * c * 2 * c) + 2 * c * c + 2 * c * fm
    return per_frame * bsz * t_len


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    bsz = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    t_len = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    dev = torch.device("cuda", 0)
    rows = []
    for i, (in_dims, n_feats, c, nl, cyc, hidden) in enumerate(SHAPES):
        set_hp(hidden_size=hidden)
        args = dict(num_layers=nl, num_channels=c, dilation_cycle_length=cyc)
        shapes = synth.backbone_param_shapes("wavenet", in_dims, n_feats, hidden_size=hidden, **args)
        params = synth.synth_state_dict(shapes, seed=100 + c + nl)
        net = build_backbone(in_dims, n_feats, "wavenet", args)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
        net = net.to(dev).eval()
        x = torch.from_numpy(synth.synth_normal((bsz, n_feats, in_dims, t_len), 31)).to(dev)
        cond = torch.from_numpy(synth.synth_normal((bsz, hidden, t_len), 32)).to(dev)
        t = torch.from_numpy((np.arange(bsz) * 97.5 + 3.0).astype(np.float32)).to(dev)
        with torch.no_grad():
            for _ in range(3):
                net(x, t, cond)
            torch.cuda.synchronize()
            n = 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                net(x, t, cond)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            net.kernel_timing(True)
            for _ in range(8):
                net(x, t, cond)
            torch.cuda.synchronize()
            classes = net.kernel_classes()
            net.kernel_timing(False)
        st = net.stats()
        fl = flops_per_eval(in_dims, n_feats, c, nl, bsz, t_len)
        row = dict(shape=f"wn{i}", in_dims=in_dims, n_feats=n_feats, C=c, L=nl, cycle=cyc, hidden=hidden, batch=bsz, frames=t_len,
                   ms_per_eval=round(ms, 4), tflops=round(fl / ms / 1e9, 2), frac=round(fl / ms / 1e9 / PEAK_TF, 4),
                   kernels_per_nfe=st.get("kernels_per_nfe"),
                   kernels=[(k["name"], round(k["mean_ms"] * 1e3, 2)) for k in classes])
        rows.append(row)
        print(json.dumps(row), flush=True)
        net.release_native()
        del net
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"{tag}_shapes.json"), "w") as f:
        json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
