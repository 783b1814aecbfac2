"""Parity of the HIP path (through the C-ABI) against the numpy oracle and the committed golden
fixtures generated from the reference.  Run on the MI355X box: `pytest tests -m gpu`.

Stated fp32 tolerances, held by BOTH max |a - b| / max |ref| and rms(a - b) / rms(ref) (gpu_util.check; the second cannot
be flattered by the ~260 peaks of the random-weight sampler fixtures on a [-12, 0] mel range):
  * one backbone evaluation            <= 2e-5    (measured worst case 6.1e-6: T = 4128; typical 1-3e-6)
  * full sampler runs (<= 120 NFE)     <= 1.5e-5  (measured worst case 1.8e-6: 50-NFE DPM-Solver++ at B = 8; the headline
                                                   50-NFE run 1.7e-6) - i.e. at most ~10x what profiles/r02_parity.json
                                                   records, so a drift of one order of magnitude fails
Every comparison is appended to gpurun_out/parity.jsonl; tools/summarize_parity.py writes profiles/<round>_parity.json.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from diffsinger_amd import synth  # noqa: E402
from gpu_util import check, dev, load_synth, make_backbone, rel_err, set_hp, synth_params  # noqa: E402
from oracle import backbones as ob  # noqa: E402
from oracle import diffusion as od  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL_NFE = 2e-5
TOL_SAMPLER = 1.5e-5


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    set_hp()


WN = {
    "wn_acoustic": (128, 1, dict(num_layers=20, num_channels=256, dilation_cycle_length=4)),
    "wn_pitch": (64, 1, dict(num_layers=20, num_channels=256, dilation_cycle_length=5)),
    "wn_multivar": (24, 2, dict(num_layers=10, num_channels=192, dilation_cycle_length=4)),
    "wn_small": (32, 1, dict(num_layers=4, num_channels=64, dilation_cycle_length=2)),
    "wn_c250": (32, 1, dict(num_layers=5, num_channels=250, dilation_cycle_length=3)),
}
LX = {
    "lx_default": (128, 1, dict(num_layers=6, num_channels=512, expansion_factor=2, kernel_size=31,
                                activation="PReLU", strong_cond=False)),
    "lx_acoustic1024": (128, 1, dict(num_layers=6, num_channels=1024, expansion_factor=2, kernel_size=31,
                                     activation="PReLU", strong_cond=True)),
    "lx_silu": (64, 1, dict(num_layers=2, num_channels=128, expansion_factor=2, kernel_size=31,
                            activation="SiLU", strong_cond=False)),
    "lx_relu": (24, 2, dict(num_layers=2, num_channels=128, expansion_factor=1, kernel_size=7,
                            activation="ReLU", strong_cond=True)),
}


def _golden_backbone_cases(kind, name, table, prefix):
    in_dims, n_feats, args = table[name]
    g = load(prefix + name)
    set_hp()
    net, params = make_backbone(kind, in_dims, n_feats, args, int(g["weight_seed"]))
    assert synth.state_dict_digest(params) == str(g["digest"])
    ci = 0
    while f"c{ci}_meta" in g:
        bsz, t_len, xs, cs, _ = (int(v) for v in g[f"c{ci}_meta"])
        x = synth.synth_normal((bsz, n_feats, in_dims, t_len), xs)
        cond = synth.synth_normal((bsz, 256, t_len), cs)
        t = g[f"c{ci}_t"]
        with torch.no_grad():
            out = net(dev(x), dev(t), dev(cond))
        torch.cuda.synchronize()
        assert tuple(out.shape) == g[f"c{ci}_out"].shape
        check(out, g[f"c{ci}_out"], TOL_NFE, what=(name, ci))
        ci += 1
    assert ci > 0
    net.release_native()


@pytest.mark.parametrize("name", sorted(WN))
def test_wavenet_single_nfe_vs_golden(name):
    _golden_backbone_cases("wavenet", name, WN, "g2_")


@pytest.mark.parametrize("name", sorted(LX))
def test_lynxnet_single_nfe_vs_golden(name):
    _golden_backbone_cases("lynxnet", name, LX, "g3_")


@pytest.mark.parametrize("bsz,t_len", [(1, 1), (1, 31), (3, 64), (2, 257), (1, 1000)])
def test_wavenet_vs_oracle_ragged_sizes(bsz, t_len):
    """Edge sizes: T = 1, T < dilation, T not a multiple of any tile, exact tile multiples, headline T."""
    set_hp()
    args = dict(num_layers=20, num_channels=256, dilation_cycle_length=4)
    net, params = make_backbone("wavenet", 128, 1, args, 42)
    x = synth.synth_normal((bsz, 1, 128, t_len), 11)
    cond = synth.synth_normal((bsz, 256, t_len), 12)
    t = (np.arange(bsz) * 333.25 + 7.5).astype(np.float32)
    want = ob.wavenet_forward(params, x, t, cond, dilation_cycle_length=4)
    with torch.no_grad():
        out = net(dev(x), dev(t), dev(cond))
        out2 = net(dev(x), dev(t), dev(cond))              # second call: cached cond, same result
    check(out, want, TOL_NFE)
    assert torch.equal(out, out2)
    net.release_native()


def test_wavenet_intermediate_state_does_not_leak_between_calls():
    """The handle is reused across shapes and inputs: results must depend on the inputs only."""
    set_hp()
    args = dict(num_layers=4, num_channels=64, dilation_cycle_length=2)
    net, params = make_backbone("wavenet", 32, 1, args, 45)
    outs = []
    for bsz, t_len in ((2, 50), (1, 200), (2, 50)):
        x = synth.synth_normal((bsz, 1, 32, t_len), 5)
        cond = synth.synth_normal((bsz, 256, t_len), 6)
        t = np.full((bsz,), 123.0, np.float32)
        with torch.no_grad():
            outs.append(net(dev(x), dev(t), dev(cond)).cpu().numpy())
        want = ob.wavenet_forward(params, x, t, cond, dilation_cycle_length=2)
        check(outs[-1], want, TOL_NFE)
    np.testing.assert_array_equal(outs[0], outs[2])
    net.release_native()


def test_lynxnet_vs_oracle_ragged():
    set_hp()
    args = dict(num_layers=3, num_channels=256, expansion_factor=2, kernel_size=31, activation="PReLU",
                strong_cond=True)
    net, params = make_backbone("lynxnet", 128, 1, args, 56)
    for bsz, t_len in ((1, 5), (2, 333)):
        x = synth.synth_normal((bsz, 1, 128, t_len), 21)
        cond = synth.synth_normal((bsz, 256, t_len), 22)
        t = np.array([500.5], np.float32)           # [1] step broadcast over the batch (reflow.py:135)
        want = ob.lynxnet_forward(params, x, t, cond, activation="PReLU", strong_cond=True)
        with torch.no_grad():
            out = net(dev(x), dev(t), dev(cond))
        check(out, want, TOL_NFE)
    net.release_native()


# ------------------------------------------------------------------------------------------------
# samplers through the coarse boundary (GaussianDiffusion.forward / RectifiedFlow.forward)
# ------------------------------------------------------------------------------------------------
SN = dict(in_dims=32, n_feats=1, args=dict(num_layers=4, num_channels=64, dilation_cycle_length=2), wseed=45)

GD_CASES = {
    "ddim10": (dict(diff_accelerator="ddim", diff_speedup=10, K_step_infer=1000), 1000, False),
    "ddim100": (dict(diff_accelerator="ddim", diff_speedup=100, K_step_infer=1000), 1000, False),
    "pndm20": (dict(diff_accelerator="pndm", diff_speedup=20, K_step_infer=1000), 1000, False),
    "dpm20": (dict(diff_accelerator="dpm-solver", diff_speedup=50, K_step_infer=1000), 1000, False),
    "dpm50": (dict(diff_accelerator="dpm-solver", diff_speedup=20, K_step_infer=1000), 1000, False),
    "dpm5": (dict(diff_accelerator="dpm-solver", diff_speedup=200, K_step_infer=1000), 1000, False),
    "unipc20": (dict(diff_accelerator="unipc", diff_speedup=50, K_step_infer=1000), 1000, False),
    "unipc50": (dict(diff_accelerator="unipc", diff_speedup=20, K_step_infer=1000), 1000, False),
    "ddpm_shallow20": (dict(diff_accelerator="ddim", diff_speedup=1, K_step_infer=20), 400, True),
    "dpm_shallow": (dict(diff_accelerator="dpm-solver", diff_speedup=20, K_step_infer=400), 400, True),
    "ddim_shallow": (dict(diff_accelerator="ddim", diff_speedup=10, K_step_infer=200), 400, True),
}


def _gd(k_step, kind="wavenet", in_dims=None, args=None, wseed=None):
    from diffsinger_amd.diffusion import GaussianDiffusion
    in_dims = in_dims or SN["in_dims"]
    args = args or SN["args"]
    wseed = wseed or SN["wseed"]
    d = GaussianDiffusion(in_dims, 1, timesteps=1000, k_step=k_step, backbone_type=kind, backbone_args=args,
                          spec_min=[-12.0], spec_max=[0.0])
    load_synth(d.denoise_fn, synth_params(kind, in_dims, 1, args, wseed))
    return d.cuda().eval()


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("tag", sorted(GD_CASES))
def test_gaussian_diffusion_samplers_vs_golden(tag, use_graph):
    hp, k_step, shallow = GD_CASES[tag]
    g = load("g5_samplers")
    bsz, t_len, nseed, n_randn, _, _ = (int(v) for v in g[f"{tag}_meta"])
    set_hp(use_shallow_diffusion=shallow, **hp)
    d = _gd(k_step)
    d.use_graph = use_graph
    cond = dev(synth.synth_normal((bsz, t_len, 256), nseed + 500))
    src = None
    if shallow:
        src = dev((synth.synth_normal((bsz, t_len, 32), nseed + 501) * 1.5 - 6.0).astype(np.float32))
    noise = dev(synth.synth_normal((bsz, 1, 32, t_len), nseed))
    step_noise = None
    if n_randn > 1:
        step_noise = dev(np.stack([synth.synth_normal((bsz, 1, 32, t_len), nseed + 1 + i) for i in range(n_randn - 1)]))
    out = d(cond, src_spec=src, infer=True, noise=noise, step_noise=step_noise)
    out_again = d(cond, src_spec=src, infer=True, noise=noise, step_noise=step_noise)   # graph replay / cache
    check(out, g[f"{tag}_out"], TOL_SAMPLER, what=tag)
    assert torch.equal(out, out_again)
    d.denoise_fn.release_native()


RF_CASES = {"rf_euler20": ("euler", 0.0, False), "rf_rk2_20": ("rk2", 0.0, False), "rf_rk4_20": ("rk4", 0.0, False),
            "rf_rk5_20": ("rk5", 0.0, False), "rf_euler_shallow": ("euler", 0.4, True)}


@pytest.mark.parametrize("tag", sorted(RF_CASES))
def test_rectified_flow_samplers_vs_golden(tag):
    from diffsinger_amd.diffusion import RectifiedFlow
    algo, t_start, shallow = RF_CASES[tag]
    g = load("g5_samplers")
    bsz, t_len, nseed, _, steps, _ = (int(v) for v in g[f"{tag}_meta"])
    set_hp(use_shallow_diffusion=shallow, sampling_algorithm=algo, sampling_steps=steps, T_start_infer=t_start)
    r = RectifiedFlow(32, 1, t_start=t_start, time_scale_factor=1000, backbone_type="wavenet",
                      backbone_args=SN["args"], spec_min=[-12.0], spec_max=[0.0])
    load_synth(r.velocity_fn, synth_params("wavenet", 32, 1, SN["args"], SN["wseed"]))
    r = r.cuda().eval()
    cond = dev(synth.synth_normal((bsz, t_len, 256), nseed + 500))
    src = None
    if shallow:
        src = dev((synth.synth_normal((bsz, t_len, 32), nseed + 501) * 1.5 - 6.0).astype(np.float32))
    out = r(cond, src_spec=src, infer=True, noise=dev(synth.synth_normal((bsz, 1, 32, t_len), nseed)))
    check(out, g[f"{tag}_out"], TOL_SAMPLER, what=tag)
    r.velocity_fn.release_native()


def test_full_size_wavenet_dpm20_vs_golden():
    g = load("g5_full_dpm20")
    bsz, t_len, nseed, _, cseed = (int(v) for v in g["meta"])
    set_hp(diff_accelerator="dpm-solver", diff_speedup=50, K_step_infer=1000)
    d = _gd(1000, in_dims=128, args=dict(num_layers=20, num_channels=256, dilation_cycle_length=4), wseed=42)
    out = d(dev(synth.synth_normal((bsz, t_len, 256), cseed)), infer=True,
            noise=dev(synth.synth_normal((bsz, 1, 128, t_len), nseed)))
    check(out, g["out"], TOL_SAMPLER)
    d.denoise_fn.release_native()


def test_lynxnet_samplers_vs_golden():
    from diffsinger_amd.diffusion import RectifiedFlow
    g = load("g5_lynx")
    bsz, t_len, s_ddim, s_rf, cseed, wseed = (int(v) for v in g["meta"])
    largs = dict(num_layers=3, num_channels=256, expansion_factor=2, kernel_size=31, activation="PReLU",
                 strong_cond=True)
    cond = dev(synth.synth_normal((bsz, t_len, 256), cseed))
    set_hp(diff_accelerator="ddim", diff_speedup=50, K_step_infer=1000)
    d = _gd(1000, kind="lynxnet", in_dims=128, args=largs, wseed=wseed)
    out = d(cond, infer=True, noise=dev(synth.synth_normal((bsz, 1, 128, t_len), s_ddim)))
    check(out, g["ddim20_out"], TOL_SAMPLER)
    d.denoise_fn.release_native()
    set_hp(sampling_algorithm="euler", sampling_steps=10)
    r = RectifiedFlow(128, 1, backbone_type="lynxnet", backbone_args=largs, spec_min=[-12.0], spec_max=[0.0])
    load_synth(r.velocity_fn, synth_params("lynxnet", 128, 1, largs, wseed))
    r = r.cuda().eval()
    out2 = r(cond, infer=True, noise=dev(synth.synth_normal((bsz, 1, 128, t_len), s_rf)))
    check(out2, g["rf_euler10_out"], TOL_SAMPLER)
    r.velocity_fn.release_native()


def test_headline_config_properties_full_size():
    """BASELINE config 2 at full size (20x256 WaveNet, DPM-Solver++ 1000->50, B=1, T=1000), checked through
    size-independent properties instead of the (slow) oracle: determinism, batch independence (utterances do
    not interact), and time-locality (the receptive field of one NFE is +-75 frames)."""
    set_hp(diff_accelerator="dpm-solver", diff_speedup=20, K_step_infer=1000)
    d = _gd(1000, in_dims=128, args=dict(num_layers=20, num_channels=256, dilation_cycle_length=4), wseed=42)
    t_len = 1000
    cond = dev(synth.synth_normal((2, t_len, 256), 900))
    noise = dev(synth.synth_normal((2, 1, 128, t_len), 901))
    out_b2 = d(cond, infer=True, noise=noise)
    out_b1 = d(cond[1:2].contiguous(), infer=True, noise=noise[1:2].contiguous())
    assert torch.isfinite(out_b2).all()
    assert rel_err(out_b1, out_b2[1:2].cpu().numpy()) < 1e-6       # same arithmetic per utterance
    net = d.denoise_fn
    x = dev(synth.synth_normal((1, 1, 128, t_len), 902))
    c1 = dev(synth.synth_normal((1, 256, t_len), 903))
    t = dev(np.array([412.0], np.float32))
    with torch.no_grad():
        y1 = net(x, t, c1)
        x2 = x.clone()
        x2[..., 600:] += 1.0                                         # perturb frames >= 600
        y2 = net(x2, t, c1)
    assert torch.equal(y1[..., :600 - 75], y2[..., :600 - 75])      # outside the receptive field: identical
    assert not torch.equal(y1[..., 600:], y2[..., 600:])
    net.release_native()


def test_headline_config_full_size_vs_oracle():
    """BASELINE config 2 exactly (20x256 WaveNet, DPM-Solver++ 1000->50, B=1, T=1000) against the numpy oracle
    run on the host cores (~50 backbone evaluations); tolerance: TOL_SAMPLER = 1.5e-5 (max and RMS) after 50 solver steps."""
    set_hp(diff_accelerator="dpm-solver", diff_speedup=20, K_step_infer=1000)
    args = dict(num_layers=20, num_channels=256, dilation_cycle_length=4)
    d = _gd(1000, in_dims=128, args=args, wseed=42)
    t_len = 1000
    cond = synth.synth_normal((1, t_len, 256), 0)
    noise = synth.synth_normal((1, 1, 128, t_len), 1)
    out = d(dev(cond), infer=True, noise=dev(noise))
    params = synth_params("wavenet", 128, 1, args, 42)
    fn = lambda x, t, c: ob.wavenet_forward(params, x, t, c, dilation_cycle_length=4)
    o = od.GaussianDiffusion(fn, 128, 1, spec_min=[-12.0], spec_max=[0.0])
    want = o.forward(cond, noise, diff_accelerator="dpm-solver", diff_speedup=20, K_step_infer=1000)
    check(out, want, TOL_SAMPLER)
    d.denoise_fn.release_native()


def test_long_utterance_single_nfe_vs_oracle():
    """T = 4128 (the longest .ds segment in the reference's samples), B = 2: many tiles per
    utterance, 64-frame-tile kernels."""
    set_hp()
    args = dict(num_layers=20, num_channels=256, dilation_cycle_length=4)
    net, params = make_backbone("wavenet", 128, 1, args, 42)
    bsz, t_len = 2, 4128
    x = synth.synth_normal((bsz, 1, 128, t_len), 31)
    cond = synth.synth_normal((bsz, 256, t_len), 32)
    t = np.array([17.0, 940.5], np.float32)
    want = ob.wavenet_forward(params, x, t, cond, dilation_cycle_length=4)
    with torch.no_grad():
        out = net(dev(x), dev(t), dev(cond))
    check(out, want, TOL_NFE)
    net.release_native()


def test_reference_error_behaviour():
    from diffsinger_amd.diffusion import RectifiedFlow
    set_hp(diff_accelerator="nope", diff_speedup=10, K_step_infer=1000)
    d = _gd(1000)
    cond = dev(synth.synth_normal((1, 8, 256), 1))
    with pytest.raises(ValueError, match="Unsupported acceleration algorithm"):
        d(cond, infer=True)
    set_hp(diff_accelerator="ddim", diff_speedup=7, K_step_infer=1000)
    with pytest.raises(AssertionError, match="factor of diffusion depth"):
        d(cond, infer=True)
    set_hp(diff_accelerator="pndm", diff_speedup=20, K_step_infer=1000)
    with pytest.raises(RuntimeError):
        d(dev(synth.synth_normal((2, 8, 256), 1)), infer=True)
    with pytest.raises(NotImplementedError):
        d(cond, gt_spec=torch.zeros(1, 8, 32).cuda(), infer=False)
    d.denoise_fn.release_native()
    set_hp(use_shallow_diffusion=True, diff_accelerator="ddim", diff_speedup=10, K_step_infer=100)
    d2 = _gd(400)
    with pytest.raises(AssertionError, match="Missing shallow diffusion source"):
        d2(cond, infer=True)
    d2.denoise_fn.release_native()
    set_hp(sampling_algorithm="nope", sampling_steps=4)
    r = RectifiedFlow(32, 1, backbone_type="wavenet", backbone_args=SN["args"], spec_min=[-12.0], spec_max=[0.0])
    load_synth(r.velocity_fn, synth_params("wavenet", 32, 1, SN["args"], SN["wseed"]))
    with pytest.raises(ValueError, match="Unsupported algorithm for Rectified Flow"):
        r.cuda()(cond, infer=True)
    r.velocity_fn.release_native()


def test_variance_wrappers_gpu_vs_oracle():
    """PitchDiffusion / MultiVarianceDiffusion end to end (norm -> sample -> denorm/mean/clamp), config-5 shapes."""
    from diffsinger_amd.diffusion import MultiVarianceRectifiedFlow, PitchDiffusion
    args = dict(num_layers=4, num_channels=64, dilation_cycle_length=2)
    bsz, t_len = 2, 40
    cond = synth.synth_normal((bsz, t_len, 256), 70)
    # pitch, DDIM
    set_hp(diff_accelerator="ddim", diff_speedup=100, K_step_infer=1000)
    p = PitchDiffusion(vmin=-8.0, vmax=8.0, cmin=-12.0, cmax=12.0, repeat_bins=64, backbone_type="wavenet",
                       backbone_args=args)
    params = synth_params("wavenet", 64, 1, args, 71)
    load_synth(p.denoise_fn, params)
    p = p.cuda().eval()
    noise = synth.synth_normal((bsz, 1, 64, t_len), 72)
    out = p(dev(cond), infer=True, noise=dev(noise))
    fn = lambda x, t, c: ob.wavenet_forward(params, x, t, c, dilation_cycle_length=2)
    nf, smin, smax = od.repetitive_spec_ranges(-8.0, 8.0)
    o = od.GaussianDiffusion(fn, 64, nf, spec_min=smin, spec_max=smax)
    xo = o.inference(np.ascontiguousarray(np.swapaxes(cond, 1, 2)), noise, diff_speedup=100, diff_accelerator="ddim",
                     K_step_infer=1000)
    want = od.pitch_denorm(o, xo, -12.0, 12.0)
    assert tuple(out.shape) == want.shape == (bsz, t_len)
    check(out, want, TOL_SAMPLER, what="PitchDiffusion, DDIM 10 steps (values clamp to [-12, 12])")
    p.denoise_fn.release_native()
    # multi-variance (F = 2), reflow euler
    set_hp(sampling_algorithm="euler", sampling_steps=8)
    ranges, clamps = [(-96.0, -12.0), (-96.0, -20.0)], [(-96.0, 0.0), None]
    m = MultiVarianceRectifiedFlow(ranges=ranges, clamps=clamps, repeat_bins=24, backbone_type="wavenet",
                                   backbone_args=args)
    params2 = synth_params("wavenet", 24, 2, args, 73)
    load_synth(m.velocity_fn, params2)
    m = m.cuda().eval()
    noise2 = synth.synth_normal((bsz, 2, 24, t_len), 74)
    outs = m(dev(cond), infer=True, noise=dev(noise2))
    fn2 = lambda x, t, c: ob.wavenet_forward(params2, x, t, c, dilation_cycle_length=2)
    nf, smin, smax = od.repetitive_spec_ranges([r[0] for r in ranges], [r[1] for r in ranges])
    orf = od.RectifiedFlow(fn2, 24, nf, spec_min=smin, spec_max=smax)
    xo = orf.inference(np.ascontiguousarray(np.swapaxes(cond, 1, 2)), noise2, sampling_algorithm="euler", sampling_steps=8)
    want = od.multivar_denorm(orf, xo, clamps)
    assert len(outs) == 2
    for i, (a, w) in enumerate(zip(outs, want)):
        check(a, w, TOL_SAMPLER, what=("MultiVarianceRectifiedFlow, euler 8", i))
    m.velocity_fn.release_native()


def test_config3_lynxnet_full_width_ddim_vs_oracle():
    """BASELINE config 3's network and sampler at full width (LYNXNet 6x1024, k=31, strong_cond, PReLU; DDIM) with a
    batch of 8 utterances, against the numpy oracle.  The loop is shortened to 10 of the 100 DDIM steps (speed-up
    100) and T to 200 frames so that the host-side oracle finishes in seconds; tolerance TOL_SAMPLER = 1.5e-5."""
    largs = dict(num_layers=6, num_channels=1024, expansion_factor=2, kernel_size=31, activation="PReLU", strong_cond=True)
    set_hp(diff_accelerator="ddim", diff_speedup=100, K_step_infer=1000)
    d = _gd(1000, kind="lynxnet", in_dims=128, args=largs, wseed=77)
    bsz, t_len = 8, 200
    cond = synth.synth_normal((bsz, t_len, 256), 70)
    noise = synth.synth_normal((bsz, 1, 128, t_len), 71)
    out = d(dev(cond), infer=True, noise=dev(noise))
    params = synth_params("lynxnet", 128, 1, largs, 77)
    fn = lambda x, t, c: ob.lynxnet_forward(params, x, t, c, activation="PReLU", strong_cond=True)   # noqa: E731
    o = od.GaussianDiffusion(fn, 128, 1, spec_min=[-12.0], spec_max=[0.0])
    want = o.forward(cond, noise, diff_accelerator="ddim", diff_speedup=100, K_step_infer=1000)
    check(out, want, TOL_SAMPLER)
    d.denoise_fn.release_native()


def test_config4_per_gpu_batch_vs_oracle():
    """BASELINE config 4's per-GPU share (8 utterances of the 20x256 WaveNet, T=1000: the 64-frame-tile kernels) with
    DPM-Solver++ shortened to 10 steps so that the host-side oracle finishes in seconds; tolerance TOL_SAMPLER = 1.5e-5."""
    set_hp(diff_accelerator="dpm-solver", diff_speedup=100, K_step_infer=1000)
    args = dict(num_layers=20, num_channels=256, dilation_cycle_length=4)
    d = _gd(1000, in_dims=128, args=args, wseed=42)
    bsz, t_len = 8, 1000
    cond = synth.synth_normal((bsz, t_len, 256), 40)
    noise = synth.synth_normal((bsz, 1, 128, t_len), 41)
    out = d(dev(cond), infer=True, noise=dev(noise))
    params = synth_params("wavenet", 128, 1, args, 42)
    fn = lambda x, t, c: ob.wavenet_forward(params, x, t, c, dilation_cycle_length=4)   # noqa: E731
    o = od.GaussianDiffusion(fn, 128, 1, spec_min=[-12.0], spec_max=[0.0])
    want = o.forward(cond, noise, diff_accelerator="dpm-solver", diff_speedup=100, K_step_infer=1000)
    check(out, want, TOL_SAMPLER)
    d.denoise_fn.release_native()


def test_config5_variance_full_size_vs_oracle():
    """BASELINE config 5's two denoisers at the sizes of configs/variance.yaml (pitch: 20x256 WaveNet, cycle 5, 64 repeat
    bins; variances: 10x192, cycle 4, 48 bins for 2 parameters), rectified flow, euler 20, against the numpy oracle."""
    from diffsinger_amd.diffusion import MultiVarianceRectifiedFlow, PitchRectifiedFlow
    bsz, t_len = 2, 211
    cond = synth.synth_normal((bsz, t_len, 256), 80)
    cond_t = np.ascontiguousarray(np.swapaxes(cond, 1, 2))
    set_hp(sampling_algorithm="euler", sampling_steps=20)
    # pitch
    pargs = dict(num_layers=20, num_channels=256, dilation_cycle_length=5)
    p = PitchRectifiedFlow(vmin=-8.0, vmax=8.0, cmin=-12.0, cmax=12.0, repeat_bins=64, backbone_type="wavenet",
                           backbone_args=pargs)
    params = synth_params("wavenet", 64, 1, pargs, 81)
    load_synth(p.velocity_fn, params)
    p = p.cuda().eval()
    noise = synth.synth_normal((bsz, 1, 64, t_len), 82)
    out = p(dev(cond), infer=True, noise=dev(noise))
    fn = lambda x, t, c: ob.wavenet_forward(params, x, t, c, dilation_cycle_length=5)
    nf, smin, smax = od.repetitive_spec_ranges(-8.0, 8.0)
    o = od.RectifiedFlow(fn, 64, nf, spec_min=smin, spec_max=smax)
    want = od.pitch_denorm(o, o.inference(cond_t, noise, sampling_algorithm="euler", sampling_steps=20), -12.0, 12.0)
    assert tuple(out.shape) == want.shape == (bsz, t_len)
    check(out, want, TOL_SAMPLER, what="config 5 pitch, B = 2, T = 211")
    p.velocity_fn.release_native()
    # energy + breathiness
    vargs = dict(num_layers=10, num_channels=192, dilation_cycle_length=4)
    ranges, clamps = [(-96.0, -12.0), (-96.0, -20.0)], [(-96.0, 0.0), (-96.0, 0.0)]
    m = MultiVarianceRectifiedFlow(ranges=ranges, clamps=clamps, repeat_bins=24, backbone_type="wavenet",
                                   backbone_args=vargs)
    params2 = synth_params("wavenet", 24, 2, vargs, 83)
    load_synth(m.velocity_fn, params2)
    m = m.cuda().eval()
    noise2 = synth.synth_normal((bsz, 2, 24, t_len), 84)
    outs = m(dev(cond), infer=True, noise=dev(noise2))
    fn2 = lambda x, t, c: ob.wavenet_forward(params2, x, t, c, dilation_cycle_length=4)
    nf, smin, smax = od.repetitive_spec_ranges([r[0] for r in ranges], [r[1] for r in ranges])
    orf = od.RectifiedFlow(fn2, 24, nf, spec_min=smin, spec_max=smax)
    want2 = od.multivar_denorm(orf, orf.inference(cond_t, noise2, sampling_algorithm="euler", sampling_steps=20), clamps)
    assert len(outs) == 2
    for i, (a, w) in enumerate(zip(outs, want2)):
        check(a, w, TOL_SAMPLER, what=("config 5 variances, B = 2, T = 211", i))
    m.velocity_fn.release_native()


def test_config1_full_size_wavenet_pndm50_vs_golden():
    """BASELINE configs[0] (the reference's own CPU-runnable case): 20x256 WaveNet, one utterance, PNDM 1000 -> 50
    (ddpm.py:149-204,323-347), against the fixture generated from the reference."""
    g = load("g5_config1_pndm50")
    bsz, t_len, nseed, _, cseed = (int(v) for v in g["meta"])
    set_hp(diff_accelerator="pndm", diff_speedup=20, K_step_infer=1000)
    d = _gd(1000, in_dims=128, args=dict(num_layers=20, num_channels=256, dilation_cycle_length=4), wseed=42)
    out = d(dev(synth.synth_normal((bsz, t_len, 256), cseed)), infer=True,
            noise=dev(synth.synth_normal((bsz, 1, 128, t_len), nseed)))
    check(out, g["out"], TOL_SAMPLER)
    d.denoise_fn.release_native()


# ---------------------------------------------------------------------------------------------------------------
# G16: the runtime inputs of the ONNX deployment twins (deployment/modules/diffusion.py:105-161, rectified_flow.py:37-68)
# ---------------------------------------------------------------------------------------------------------------

GD_ONNX = ["gd_steps30", "gd_steps7", "gd_depth037_steps11", "gd_depth06_steps50", "gd_depth1_steps20",
           "gd_depth0012_steps20"]
RF_ONNX = ["rf_steps20", "rf_depth05_steps13", "rf_depth09_steps9", "rf_depth1_steps10", "rf_depth0_steps5"]


@pytest.mark.parametrize("tag", GD_ONNX)
def test_g16_gaussian_diffusion_onnx_twin_vs_golden(tag):
    g = load("g16_onnx_twins")
    t_len, nseed, n_randn, k_step, shallow, steps = (int(v) for v in g[f"{tag}_meta"])
    depth = float(g[f"{tag}_depth"])
    set_hp(use_shallow_diffusion=bool(shallow))
    d = _gd(k_step)
    cond = dev(synth.synth_normal((1, t_len, 256), nseed + 500))
    src = None if depth < 0 else dev((synth.synth_normal((1, t_len, 32), nseed + 501) * 1.5 - 6.0).astype(np.float32))
    noise = dev(synth.synth_normal((1, 1, 32, t_len), nseed))
    step_noise = None
    if n_randn > 1:
        step_noise = dev(np.stack([synth.synth_normal((1, 1, 32, t_len), nseed + 1 + i) for i in range(n_randn - 1)]))
    out = d.forward_onnx(cond, x_start=src, depth=None if depth < 0 else torch.tensor(depth), steps=steps, noise=noise,
                         step_noise=step_noise)
    check(out, g[f"{tag}_out"], 5e-6, what=tag)
    # any `steps` is legal here; the hparams-driven path asserts divisibility instead (ddpm.py:225)
    d.denoise_fn.release_native()


@pytest.mark.parametrize("tag", RF_ONNX)
def test_g16_rectified_flow_onnx_twin_vs_golden(tag):
    from diffsinger_amd.diffusion import RectifiedFlow
    g = load("g16_onnx_twins")
    t_len, nseed, _, shallow, steps = (int(v) for v in g[f"{tag}_meta"])
    depth, t_start = float(g[f"{tag}_depth"]), float(g[f"{tag}_tstart"])
    set_hp(use_shallow_diffusion=bool(shallow))
    r = RectifiedFlow(SN["in_dims"], 1, t_start=t_start, time_scale_factor=1000, backbone_type="wavenet",
                      backbone_args=SN["args"], spec_min=[-12.0], spec_max=[0.0])
    load_synth(r.velocity_fn, synth_params("wavenet", SN["in_dims"], 1, SN["args"], SN["wseed"]))
    r = r.cuda().eval()
    cond = dev(synth.synth_normal((1, t_len, 256), nseed + 500))
    src = None if depth < 0 else dev((synth.synth_normal((1, t_len, 32), nseed + 501) * 1.5 - 6.0).astype(np.float32))
    noise = dev(synth.synth_normal((1, 1, 32, t_len), nseed))
    out = r.forward_onnx(cond, x_end=src, depth=None if depth < 0 else torch.tensor(depth), steps=steps, noise=noise)
    check(out, g[f"{tag}_out"], 2e-6, what=tag)
    r.velocity_fn.release_native()


# ---------------------------------------------------------------------------------------------------------------
# full-length runs of the remaining BASELINE configs
# ---------------------------------------------------------------------------------------------------------------
def test_config3_lynxnet_properties_full_size():
    """BASELINE config 3 at full size (LYNXNet 6x1024, k = 31, strong_cond, PReLU; DDIM 1000 -> 100 NFE; B = 8, T = 1000)
    through size-independent properties: determinism, batch independence, and time-locality outside the +-90-frame
    receptive field of one evaluation (6 layers x 15 frames of the depthwise k = 31)."""
    largs = dict(num_layers=6, num_channels=1024, expansion_factor=2, kernel_size=31, activation="PReLU", strong_cond=True)
    set_hp(diff_accelerator="ddim", diff_speedup=10, K_step_infer=1000)
    d = _gd(1000, kind="lynxnet", in_dims=128, args=largs, wseed=77)
    bsz, t_len = 8, 1000
    cond = dev(synth.synth_normal((bsz, t_len, 256), 72))
    noise = dev(synth.synth_normal((bsz, 1, 128, t_len), 73))
    out = d(cond, infer=True, noise=noise)
    assert tuple(out.shape) == (bsz, t_len, 128) and torch.isfinite(out).all()
    assert torch.equal(out, d(cond, infer=True, noise=noise))                      # deterministic (graph replay included)
    one = d(cond[5:6].contiguous(), infer=True, noise=noise[5:6].contiguous())      # B = 1: other tile widths, same utterance
    check(one, out[5:6].cpu().numpy(), 5e-6, what="item 5 alone vs in the batch of 8")
    net = d.denoise_fn
    x = dev(synth.synth_normal((1, 1, 128, t_len), 74))
    c1 = cond[:1].transpose(1, 2).contiguous()
    t = dev(np.array([412.0], np.float32))
    with torch.no_grad():
        y1 = net(x, t, c1)
        x2 = x.clone()
        x2[..., 600:] += 1.0
        y2 = net(x2, t, c1)
    assert torch.equal(y1[..., :600 - 90], y2[..., :600 - 90])
    assert not torch.equal(y1[..., 600:], y2[..., 600:])
    net.release_native()


def test_config4_per_gpu_share_all_50_steps_vs_oracle():
    """BASELINE config 4's per-GPU share at FULL length: 8 utterances x 1000 frames, 20x256 WaveNet, DPM-Solver++ 1000 -> 50
    (50 NFE) - the fused residual-layer kernel (wn_layer.hip) - against the numpy oracle (about half a minute of host time),
    and one of the utterances alone (the two-GEMM kernels of gemm.hip) against the same."""
    set_hp(diff_accelerator="dpm-solver", diff_speedup=20, K_step_infer=1000)
    args = dict(num_layers=20, num_channels=256, dilation_cycle_length=4)
    d = _gd(1000, in_dims=128, args=args, wseed=42)
    bsz, t_len = 8, 1000
    cond = synth.synth_normal((bsz, t_len, 256), 40)
    noise = synth.synth_normal((bsz, 1, 128, t_len), 41)
    out = d(dev(cond), infer=True, noise=dev(noise))
    forced = os.environ.get("DSD_FUSED_LAYER")                      # diagnostic override of the per-shape choice
    stats = d.denoise_fn.stats()
    assert forced == "0" or stats["kernels_per_nfe"] in (20 + 1, 20 + 3), stats      # one launch per layer: the fused path really ran
    params = synth_params("wavenet", 128, 1, args, 42)
    fn = lambda x, t, c: ob.wavenet_forward(params, x, t, c, dilation_cycle_length=4)   # noqa: E731
    o = od.GaussianDiffusion(fn, 128, 1, spec_min=[-12.0], spec_max=[0.0])
    want = o.forward(cond, noise, diff_accelerator="dpm-solver", diff_speedup=20, K_step_infer=1000)
    check(out, want, TOL_SAMPLER, what="8 x 1000 frames, 50 NFE, fused layers")
    one = d(dev(cond[3:4]), infer=True, noise=dev(noise[3:4]))
    assert forced == "1" or os.environ.get("DSD_FUSED16") == "1" or d.denoise_fn.stats()["kernels_per_nfe"] in (2 * 20 + 1, 2 * 20 + 3)
    check(one, want[3:4], TOL_SAMPLER, what="utterance 3 alone, two GEMMs per layer")
    d.denoise_fn.release_native()


def test_schedule_follows_loaded_buffers():
    """A checkpoint's schedule buffers win over the constructor's (ddpm.py reads the registered buffers): load a state
    dict with another beta schedule, then DDIM must follow the LOADED tables - against the oracle built from them."""
    from diffsinger_amd import schedule
    set_hp(diff_accelerator="ddim", diff_speedup=50, K_step_infer=1000)
    d = _gd(1000)
    betas = schedule.cosine_beta_schedule(1000)
    other = schedule.DDPMTables(betas)
    sd = d.state_dict()
    for name in schedule.DDPMTables.NAMES:
        sd[name] = torch.from_numpy(getattr(other, name).copy())
    cond = synth.synth_normal((1, 40, 256), 11)
    noise = synth.synth_normal((1, 1, 32, 40), 12)
    before = d(dev(cond), infer=True, noise=dev(noise))            # caches a program for the linear schedule
    d.load_state_dict(sd, strict=True)
    after = d(dev(cond), infer=True, noise=dev(noise))
    params = synth_params("wavenet", SN["in_dims"], 1, SN["args"], SN["wseed"])
    fn = lambda x, t, c: ob.wavenet_forward(params, x, t, c, dilation_cycle_length=SN["args"]["dilation_cycle_length"])   # noqa: E731
    o = od.GaussianDiffusion(fn, 32, 1, spec_min=[-12.0], spec_max=[0.0], betas=betas)
    want = o.forward(cond, noise, diff_accelerator="ddim", diff_speedup=50, K_step_infer=1000)
    check(after, want, TOL_SAMPLER, what="DDIM on the loaded cosine schedule")
    assert not torch.allclose(before, after)
    d.denoise_fn.release_native()
