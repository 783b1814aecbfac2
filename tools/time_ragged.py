#!/usr/bin/env python3
"""Several segments of one project through the denoise loop: one by one (what the reference does) against ONE ragged
batch (dsd_set_lengths).  Same results per segment; GPU box only."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from diffsinger_amd import synth  # noqa: E402
from diffsinger_amd.diffusion import GaussianDiffusion  # noqa: E402
from diffsinger_amd.hparams import hparams  # noqa: E402

LENS = [1000, 930, 850, 760, 700, 640, 560, 480]
hparams.clear()
hparams.update(hidden_size=256, schedule_type="linear", use_shallow_diffusion=False, infer=False, diff_accelerator="dpm-solver",
               diff_speedup=20, K_step_infer=1000)
args = dict(num_layers=20, num_channels=256, dilation_cycle_length=4)
d = GaussianDiffusion(128, 1, timesteps=1000, k_step=1000, backbone_type="wavenet", backbone_args=args, spec_min=[-12.0],
                      spec_max=[0.0])
d.denoise_fn.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(
    synth.backbone_param_shapes("wavenet", 128, 1, hidden_size=256, **args), seed=42).items()}, strict=True)
d = d.cuda().eval()
t_max, n = max(LENS), len(LENS)
cond = torch.from_numpy(synth.synth_normal((n, t_max, 256), 0)).cuda()
noise = torch.from_numpy(synth.synth_normal((n, 1, 128, t_max), 1)).cuda()
singles = [(cond[i:i + 1, :t].contiguous(), noise[i:i + 1, :, :, :t].contiguous()) for i, t in enumerate(LENS)]


def one_by_one():
    return [d(c, infer=True, noise=z) for c, z in singles]


def ragged():
    return d(cond, infer=True, noise=noise, lengths=LENS)


with torch.no_grad():
    a, b = one_by_one(), ragged()
    worst = max(float((x - b[i:i + 1, :t]).abs().max() / x.abs().max()) for i, (x, t) in enumerate(zip(a, LENS)))
    res = {}
    def with_policy(policy):
        def run():
            d.use_graph = policy
            try:
                return one_by_one()
            finally:
                d.use_graph = "lazy"
        return run

    for name, fn in (("one by one (lazy hipGraph: the default)", one_by_one), ("one by one, a capture per segment", with_policy(True)),
                     ("one by one, no hipGraph", with_policy(False)), ("ragged batch", ragged)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / 5
frames = sum(LENS) * 50
print(f"{n} segments, {sum(LENS)} frames, DPM-Solver++ 50 NFE, WaveNet 20x256; max rel difference ragged vs alone {worst:.2e}")
for name, dt in res.items():
    print(f"  {name}: {dt * 1e3:.1f} ms  {frames / dt / 1e6:.2f} M frames/s per denoise step")
print(f"  ragged batch vs one by one: {res['one by one (lazy hipGraph: the default)'] / res['ragged batch']:.2f}x")
