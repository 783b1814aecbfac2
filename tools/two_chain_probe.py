"""Feasibility probe: does running TWO independent half-length sampling chains on two HIP streams hide the per-kernel
prologue/epilogue latency that bounds the B=1 WaveNet loop (all workgroups of one launch move through load -> MFMA ->
store in lockstep)?  Times  (a) one B=1, T=1000 chain,  (b) one B=1, T=Th chain alone,
(c) two T=Th chains concurrently (two handles, two streams).  Th = 576 models a time split with a 76-frame halo."""
import sys
import time

import torch

sys.path.insert(0, ".")
from diffsinger_amd import synth  # noqa: E402
from diffsinger_amd.hparams import hparams  # noqa: E402
from diffsinger_amd.diffusion import GaussianDiffusion  # noqa: E402

dev = torch.device("cuda", 0)
bargs = dict(num_layers=20, num_channels=256, dilation_cycle_length=4)
params = synth.synth_state_dict(synth.backbone_param_shapes("wavenet", 128, 1, hidden_size=256, **bargs), seed=42)
hparams.clear()
hparams.update(hidden_size=256, schedule_type="linear", use_shallow_diffusion=False, infer=False,
               diff_accelerator="dpm-solver", diff_speedup=20, K_step_infer=1000)


def make():
    d = GaussianDiffusion(128, 1, timesteps=1000, k_step=1000, backbone_type="wavenet", backbone_args=bargs,
                          spec_min=[-12.0], spec_max=[0.0])
    d.denoise_fn.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    d = d.to(dev).eval()
    d.use_graph = True
    return d


def inputs(t_len, seed):
    c = torch.from_numpy(synth.synth_normal((1, t_len, 256), seed)).to(dev)
    n = torch.from_numpy(synth.synth_normal((1, 1, 128, t_len), seed + 1)).to(dev)
    return c, n


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


d0, d1 = make(), make()
s0, s1 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
for th in (1000, 576, 512):
    c0, n0 = inputs(th, 1)
    c1, n1 = inputs(th, 3)

    def one():
        with torch.cuda.stream(s0):
            d0(c0, infer=True, noise=n0)

    def two():
        with torch.cuda.stream(s0):
            d0(c0, infer=True, noise=n0)
        with torch.cuda.stream(s1):
            d1(c1, infer=True, noise=n1)

    a = timed(one)
    b = timed(two)
    print(f"T={th}: one chain {a:.3f} ms, two concurrent chains {b:.3f} ms  (ratio {b / a:.3f})", flush=True)
