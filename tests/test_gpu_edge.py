"""wn_edge.hip (skip projection -> output projection + solver update -> the next evaluation's input projection in ONE launch)
against the three gemm.hip launches it replaces, forced on (DSD_EDGE=1, read per call) on grids far smaller than it is
selected for, over every sampler's program shape: 1-3 outputs per evaluation, up to four state terms per output, shallow
starts, the pitch (64 bins, dilation 16) and multi-variance (2 x 24 bins, C = 192) shapes, ragged batches - and the programs
it must leave to the GEMM path (ancestral DDPM: caller-noise terms).  The model term joins a sum last in the edge kernel, so
the two paths may differ by rounding only: <= 2e-6 of the output range over a whole loop."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from diffsinger_amd import synth  # noqa: E402
from gpu_util import check, dev, load_synth, set_hp, synth_params  # noqa: E402

ARGS = dict(num_layers=4, num_channels=256, dilation_cycle_length=4)


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    yield
    os.environ.pop("DSD_EDGE", None)
    set_hp()


class _edge:
    def __init__(self, v):
        self.v = v

    def __enter__(self):
        os.environ["DSD_EDGE"] = self.v

    def __exit__(self, *a):
        os.environ.pop("DSD_EDGE", None)


def _pair(cls, in_dims, n_feats, args, **kw):
    params = synth_params("wavenet", in_dims, n_feats, args, 42)
    out = []
    for _ in range(2):
        d = cls(in_dims, n_feats, backbone_type="wavenet", backbone_args=args, **kw)
        net = d.denoise_fn if hasattr(d, "denoise_fn") else d.velocity_fn
        load_synth(net, params)
        out.append(d.cuda().eval())
    return out


SAMPLERS = [dict(diff_accelerator="ddim", diff_speedup=100), dict(diff_accelerator="pndm", diff_speedup=100),
            dict(diff_accelerator="dpm-solver", diff_speedup=50), dict(diff_accelerator="unipc", diff_speedup=50)]


@pytest.mark.parametrize("hp", SAMPLERS, ids=lambda h: h["diff_accelerator"])
def test_edge_kernel_vs_three_gemms_ddpm_family(hp):
    from diffsinger_amd.diffusion import GaussianDiffusion
    set_hp(K_step_infer=1000, **hp)
    on, off = _pair(GaussianDiffusion, 128, 1, ARGS, spec_min=[-12.0], spec_max=[0.0])
    bsz = 1 if hp["diff_accelerator"] == "pndm" else 3
    cond = dev(synth.synth_normal((bsz, 150, 256), 40))
    noise = dev(synth.synth_normal((bsz, 1, 128, 150), 41))
    with _edge("0"):
        want = off(cond, infer=True, noise=noise)
        assert off.denoise_fn.stats()["kernels_per_nfe"] in (2 * 4 + 3, 4 + 3)           # (4 + 3: DSD_FUSED_LAYER=1 forced runs)
    with _edge("1"):
        got = [on(cond, infer=True, noise=noise) for _ in range(3)]          # eager, then graph replays
        assert on.denoise_fn.stats()["kernels_per_nfe"] in (2 * 4 + 1, 4 + 1), on.denoise_fn.stats()
    check(got[0], want.cpu().numpy(), 2e-6, what=("edge kernel vs three GEMMs", hp["diff_accelerator"]))
    assert torch.equal(got[0], got[1]) and torch.equal(got[1], got[2])
    on.denoise_fn.release_native()
    off.denoise_fn.release_native()


@pytest.mark.parametrize("algo", ["euler", "rk4"])
def test_edge_kernel_vs_three_gemms_reflow_pitch_and_variance(algo):
    from diffsinger_amd.diffusion import MultiVarianceRectifiedFlow, PitchRectifiedFlow
    set_hp(sampling_algorithm=algo, sampling_steps=6)
    pargs = dict(num_layers=6, num_channels=256, dilation_cycle_length=5)          # dilation 16 in layer 4
    on, off = _pair(lambda i, f, **kw: PitchRectifiedFlow(vmin=-8.0, vmax=8.0, cmin=-12.0, cmax=12.0, repeat_bins=64, **kw), 64, 1, pargs)
    cond = dev(synth.synth_normal((2, 133, 256), 50))
    noise = dev(synth.synth_normal((2, 1, 64, 133), 51))
    with _edge("0"):
        want = off(cond, infer=True, noise=noise)
    with _edge("1"):
        got = on(cond, infer=True, noise=noise)
    check(got, want.cpu().numpy(), 2e-6, what=("pitch reflow", algo))
    on.velocity_fn.release_native()
    off.velocity_fn.release_native()
    vargs = dict(num_layers=3, num_channels=192, dilation_cycle_length=3)
    mk = lambda i, f, **kw: MultiVarianceRectifiedFlow(ranges=[(-96.0, -12.0), (-96.0, -20.0)], clamps=[(-96.0, 0.0), (-96.0, 0.0)],  # noqa: E731
                                                       repeat_bins=24, **kw)
    on, off = _pair(mk, 24, 2, vargs)
    noise = dev(synth.synth_normal((2, 2, 24, 133), 52))
    with _edge("0"):
        want = off(cond, infer=True, noise=noise)
    with _edge("1"):
        got = on(cond, infer=True, noise=noise)
    for g, w in zip(got, want):
        check(g, w.cpu().numpy(), 2e-6, what=("multi-variance reflow", algo))
    on.velocity_fn.release_native()
    off.velocity_fn.release_native()


def test_edge_kernel_ragged_shallow_and_ancestral_fallback():
    from diffsinger_amd.diffusion import GaussianDiffusion
    set_hp(K_step_infer=400, diff_accelerator="dpm-solver", diff_speedup=40, use_shallow_diffusion=True)
    on, off = _pair(GaussianDiffusion, 128, 1, ARGS, timesteps=1000, k_step=400, spec_min=[-12.0], spec_max=[0.0])
    cond = dev(synth.synth_normal((3, 200, 256), 60))
    noise = dev(synth.synth_normal((3, 1, 128, 200), 61))
    src = dev(-6.0 + 2.0 * synth.synth_normal((3, 200, 128), 62))
    lens = [200, 77, 130]
    with _edge("0"):
        want = off(cond, src_spec=src, infer=True, noise=noise, lengths=lens)
    with _edge("1"):
        got = on(cond, src_spec=src, infer=True, noise=noise, lengths=lens)
    for b, n in enumerate(lens):
        check(got[b, :n], want[b, :n].cpu().numpy(), 2e-6, what=("ragged shallow dpm-solver, item", b))
    # ancestral DDPM injects caller noise into the update: those evaluations stay on the GEMM path (the last one, at t = 0,
    # has no noise term and takes the edge kernel)
    set_hp(K_step_infer=20, diff_accelerator="ddim", diff_speedup=1, use_shallow_diffusion=True)      # speed-up 1: ancestral
    step_noise = dev(synth.synth_normal((20, 3, 1, 128, 200), 63))
    with _edge("0"):
        want = off(cond, src_spec=src, infer=True, noise=noise, step_noise=step_noise)
    with _edge("1"):
        got = on(cond, src_spec=src, infer=True, noise=noise, step_noise=step_noise)
    check(got, want.cpu().numpy(), 2e-6, what="ancestral DDPM, 20 steps with injected noise")
    on.denoise_fn.release_native()
    off.denoise_fn.release_native()


# the shapes added in round 3: C = 128 with every F*M the kernel has, and 80 mel bins (five 16-row blocks over four waves: the
# last wave's second block does not exist) at every C
WIDE_SHAPES = [(128, 128), (128, 80), (128, 64), (128, 48), (256, 80), (192, 80)]


@pytest.mark.parametrize("shape", WIDE_SHAPES, ids=lambda s: f"c{s[0]}_fm{s[1]}")
def test_edge_kernel_c128_and_80_bins_vs_three_gemms_and_oracle(shape):
    from diffsinger_amd.diffusion import GaussianDiffusion
    from oracle import backbones as ob
    c, fm = shape
    n_feats = 2 if fm == 48 else 1
    in_dims = fm // n_feats
    args = dict(num_layers=3, num_channels=c, dilation_cycle_length=3)
    set_hp(K_step_infer=1000, diff_accelerator="dpm-solver", diff_speedup=100)
    kw = dict(spec_min=[-12.0], spec_max=[0.0])
    if n_feats == 1:
        on, off = _pair(GaussianDiffusion, in_dims, 1, args, **kw)
        net_on, net_off = on.denoise_fn, off.denoise_fn
        cond = dev(synth.synth_normal((3, 150, 256), 70))
        noise = dev(synth.synth_normal((3, 1, in_dims, 150), 71))
        lens = [150, 61, 97]
        with _edge("0"):
            want = off(cond, infer=True, noise=noise)
            want_r = off(cond, infer=True, noise=noise, lengths=lens)
        with _edge("1"):
            got = on(cond, infer=True, noise=noise)
            assert net_on.stats()["kernels_per_nfe"] in (2 * 3 + 1, 3 + 1), net_on.stats()
            got_r = on(cond, infer=True, noise=noise, lengths=lens)
        check(got, want.cpu().numpy(), 2e-6, what=("edge kernel vs three GEMMs", shape))
        for b, n in enumerate(lens):
            check(got_r[b, :n], want_r[b, :n].cpu().numpy(), 2e-6, what=("edge kernel vs three GEMMs, ragged item", shape, b))
    else:
        from diffsinger_amd.diffusion import MultiVarianceRectifiedFlow
        set_hp(sampling_algorithm="euler", sampling_steps=6)
        mk = lambda i, f, **k2: MultiVarianceRectifiedFlow(ranges=[(-96.0, -12.0), (-96.0, -20.0)], clamps=[(-96.0, 0.0), (-96.0, 0.0)],  # noqa: E731
                                                           repeat_bins=24, **k2)
        on, off = _pair(mk, 24, 2, args)
        net_on, net_off = on.velocity_fn, off.velocity_fn
        cond = dev(synth.synth_normal((2, 133, 256), 50))
        noise = dev(synth.synth_normal((2, 2, 24, 133), 52))
        with _edge("0"):
            want = off(cond, infer=True, noise=noise)
        with _edge("1"):
            got = on(cond, infer=True, noise=noise)
        for g, w in zip(got, want):
            check(g, w.cpu().numpy(), 2e-6, what=("multi-variance reflow", shape))
    # one evaluation of the backbone itself against the oracle with the edge kernel forced
    params = synth_params("wavenet", in_dims, n_feats, args, 42)
    x = synth.synth_normal((2, n_feats, in_dims, 77), 81)
    cnd = synth.synth_normal((2, 256, 77), 82)
    t = np.array([12.0, 700.5], np.float32)
    with _edge("1"):
        with torch.no_grad():
            out = net_on(dev(x), dev(t), dev(cnd))
    check(out, ob.wavenet_forward(params, x, t, cnd, dilation_cycle_length=3), 1.5e-5, what=("edge forced, one evaluation vs oracle", shape))
    net_on.release_native()
    net_off.release_native()
