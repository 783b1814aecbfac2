"""The numpy oracle against the golden fixtures generated from the imported reference
(tests/golden/make_golden.py).  This is what pins the oracle: SURVEY.md section 8(c), G1-G6.

Tolerances are stated per test.  Reference = torch 2.10 CPU (oneDNN) fp32; oracle = numpy fp32
(OpenBLAS); both accumulate in fp32 in different orders, so agreement is to a few 1e-6 relative
for one backbone evaluation and degrades with the number of solver steps.
"""
import os
from collections import OrderedDict

import numpy as np
import pytest

from diffsinger_amd import synth
from oracle import backbones as ob
from oracle import diffusion as od

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def synth_params(kind, in_dims, n_feats, args, seed, hidden=256):
    shapes = synth.backbone_param_shapes(kind, in_dims, n_feats, hidden_size=hidden, **args)
    return synth.synth_state_dict(shapes, seed=seed)


# --------------------------------------------------------------------------- G1
def test_g1_sinusoidal_pos_emb():
    g = load("g1_posemb")
    for dim in (256, 192, 512):
        got_l = ob.sinusoidal_pos_emb(g["t_long"], dim)
        got_f = ob.sinusoidal_pos_emb(g["t_float"], dim)
        # arguments reach ~1e3 rad; one ulp of the fp32 frequency table moves sin() by ~6e-5
        np.testing.assert_allclose(got_l, g[f"long_{dim}"], atol=2e-4, rtol=0)
        np.testing.assert_allclose(got_f, g[f"float_{dim}"], atol=2e-4, rtol=0)


# --------------------------------------------------------------------------- G2
WN = {
    "wn_acoustic": (128, 1, dict(num_layers=20, num_channels=256, dilation_cycle_length=4)),
    "wn_pitch": (64, 1, dict(num_layers=20, num_channels=256, dilation_cycle_length=5)),
    "wn_multivar": (24, 2, dict(num_layers=10, num_channels=192, dilation_cycle_length=4)),
    "wn_small": (32, 1, dict(num_layers=4, num_channels=64, dilation_cycle_length=2)),
    "wn_c250": (32, 1, dict(num_layers=5, num_channels=250, dilation_cycle_length=3)),
}


@pytest.mark.parametrize("name", sorted(WN))
def test_g2_wavenet_single_nfe(name):
    in_dims, n_feats, args = WN[name]
    g = load("g2_" + name)
    params = synth_params("wavenet", in_dims, n_feats, args, int(g["weight_seed"]))
    assert synth.state_dict_digest(params) == str(g["digest"])
    ci = 0
    while f"c{ci}_meta" in g:
        bsz, t_len, xs, cs, _ = (int(v) for v in g[f"c{ci}_meta"])
        x = synth.synth_normal((bsz, n_feats, in_dims, t_len), xs)
        cond = synth.synth_normal((bsz, 256, t_len), cs)
        t = g[f"c{ci}_t"]
        want_inter = f"c{ci}_x_after_0" in g
        res = ob.wavenet_forward(params, x, t, cond, dilation_cycle_length=args["dilation_cycle_length"],
                                 return_intermediates=want_inter)
        out, inter = res if want_inter else (res, {})
        assert out.shape == g[f"c{ci}_out"].shape
        # tolerance: 2e-5 of the output range for one 20-layer evaluation
        assert rel_err(out, g[f"c{ci}_out"]) < 2e-5, (name, ci)
        for k, v in inter.items():
            if f"c{ci}_{k}" in g:
                assert rel_err(v, g[f"c{ci}_{k}"]) < 2e-5, (name, ci, k)
        ci += 1
    assert ci > 0


# --------------------------------------------------------------------------- G3
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


@pytest.mark.parametrize("name", sorted(LX))
def test_g3_lynxnet_single_nfe(name):
    in_dims, n_feats, args = LX[name]
    g = load("g3_" + name)
    params = synth_params("lynxnet", in_dims, n_feats, args, int(g["weight_seed"]))
    assert synth.state_dict_digest(params) == str(g["digest"])
    ci = 0
    while f"c{ci}_meta" in g:
        bsz, t_len, xs, cs, _ = (int(v) for v in g[f"c{ci}_meta"])
        x = synth.synth_normal((bsz, n_feats, in_dims, t_len), xs)
        cond = synth.synth_normal((bsz, 256, t_len), cs)
        out = ob.lynxnet_forward(params, x, g[f"c{ci}_t"], cond, activation=args["activation"],
                                 strong_cond=args["strong_cond"])
        assert rel_err(out, g[f"c{ci}_out"]) < 2e-5, (name, ci)
        ci += 1
    assert ci > 0


# --------------------------------------------------------------------------- G4
def _dummy_gd(**kw):
    return od.GaussianDiffusion(None, 32, 1, spec_min=[-12.0], spec_max=[0.0], **kw)


def test_g4_ddpm_buffers():
    g = load("g4_schedules")
    d = _dummy_gd()
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
              "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
              "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1",
              "posterior_mean_coef2"):
        np.testing.assert_array_equal(getattr(d, k), g[k], err_msg=k)       # float64 numpy -> fp32: bit-exact
    dc = _dummy_gd(schedule_type="cosine")
    np.testing.assert_array_equal(dc.betas, g["cosine_betas"])
    np.testing.assert_array_equal(dc.alphas_cumprod, g["cosine_alphas_cumprod"])


@pytest.mark.parametrize("tag,n_keep,steps", [("full", 1000, 50), ("full20", 1000, 20), ("shallow", 400, 20)])
def test_g4_noise_schedule_vp(tag, n_keep, steps):
    g = load("g4_schedules")
    d = _dummy_gd()
    ns = od.NoiseScheduleVP(d.betas[:n_keep], clip=True)
    nu = od.NoiseScheduleVP(d.betas[:n_keep], clip=False)
    # cumsum order differs between torch and numpy: a few ulp on values of magnitude <= 2.6
    np.testing.assert_allclose(ns.log_alpha_array, g[f"{tag}_log_alpha_array"], rtol=2e-6, atol=0)
    np.testing.assert_allclose(nu.log_alpha_array, g[f"{tag}_unipc_log_alpha_array"], rtol=2e-6, atol=0)
    np.testing.assert_allclose(ns.t_array, g[f"{tag}_t_array"], rtol=2e-7, atol=0)
    ts = od.torch_linspace_f32(1.0, 1.0 / ns.total_N, steps + 1)
    np.testing.assert_allclose(ts, g[f"{tag}_timesteps"], rtol=2e-7, atol=1e-9)
    lam = np.array([ns.marginal_lambda(t) for t in ts])
    alpha = np.array([ns.marginal_alpha(t) for t in ts])
    sigma = np.array([ns.marginal_std(t) for t in ts])
    mt = np.array([ns.model_time(t) for t in ts])
    np.testing.assert_allclose(alpha, g[f"{tag}_alpha"], rtol=3e-6)
    # sigma / lambda are ill-conditioned in fp32 near t -> 0 (1 - exp(2 log_alpha) with log_alpha ~ -5e-5):
    # the reference itself is only good to ~1e-3 relative there, so is any fp32 restatement.
    np.testing.assert_allclose(sigma, g[f"{tag}_sigma"], rtol=2e-3)
    np.testing.assert_allclose(lam, g[f"{tag}_lambda"], rtol=0, atol=2e-3)
    np.testing.assert_allclose(mt, g[f"{tag}_model_t"], rtol=0, atol=1e-3)


# --------------------------------------------------------------------------- G5
SN_ARGS = dict(num_layers=4, num_channels=64, dilation_cycle_length=2)


def _sampler_net():
    params = synth_params("wavenet", 32, 1, SN_ARGS, 45)
    return lambda x, t, c: ob.wavenet_forward(params, x, t, c, dilation_cycle_length=2)


GD_CASES = {
    # tag: (hp, k_step, shallow, tolerance relative to the output range)
    "ddim10": (dict(diff_accelerator="ddim", diff_speedup=10, K_step_infer=1000), 1000, False, 2e-4),
    "ddim100": (dict(diff_accelerator="ddim", diff_speedup=100, K_step_infer=1000), 1000, False, 1e-4),
    "pndm20": (dict(diff_accelerator="pndm", diff_speedup=20, K_step_infer=1000), 1000, False, 2e-4),
    "dpm20": (dict(diff_accelerator="dpm-solver", diff_speedup=50, K_step_infer=1000), 1000, False, 2e-4),
    "dpm50": (dict(diff_accelerator="dpm-solver", diff_speedup=20, K_step_infer=1000), 1000, False, 2e-4),
    "dpm5": (dict(diff_accelerator="dpm-solver", diff_speedup=200, K_step_infer=1000), 1000, False, 1e-4),
    "unipc20": (dict(diff_accelerator="unipc", diff_speedup=50, K_step_infer=1000), 1000, False, 2e-4),
    "unipc50": (dict(diff_accelerator="unipc", diff_speedup=20, K_step_infer=1000), 1000, False, 2e-4),
    "ddpm_shallow20": (dict(diff_accelerator="ddim", diff_speedup=1, K_step_infer=20), 400, True, 1e-4),
    "dpm_shallow": (dict(diff_accelerator="dpm-solver", diff_speedup=20, K_step_infer=400), 400, True, 2e-4),
    "ddim_shallow": (dict(diff_accelerator="ddim", diff_speedup=10, K_step_infer=200), 400, True, 1e-4),
}


@pytest.mark.parametrize("tag", sorted(GD_CASES))
def test_g5_gaussian_diffusion_samplers(tag):
    hp, k_step, shallow, tol = GD_CASES[tag]
    g = load("g5_samplers")
    bsz, t_len, nseed, n_randn, _, _ = (int(v) for v in g[f"{tag}_meta"])
    d = od.GaussianDiffusion(_sampler_net(), 32, 1, timesteps=1000, k_step=k_step, spec_min=[-12.0],
                             spec_max=[0.0], use_shallow_diffusion=shallow)
    cond = synth.synth_normal((bsz, t_len, 256), nseed + 500)
    src = None
    if shallow:
        src = (synth.synth_normal((bsz, t_len, 32), nseed + 501) * 1.5 - 6.0).astype(np.float32)
    noise = synth.synth_normal((bsz, 1, 32, t_len), nseed)
    step_noise = [synth.synth_normal((bsz, 1, 32, t_len), nseed + 1 + i) for i in range(n_randn - 1)]
    out = d.forward(cond, noise, src_spec=src, step_noise=step_noise, **hp)
    assert out.shape == g[f"{tag}_out"].shape
    assert rel_err(out, g[f"{tag}_out"]) < tol, tag


RF_CASES = {
    "rf_euler20": ("euler", 20, 0.0, False, 5e-5),
    "rf_rk2_20": ("rk2", 20, 0.0, False, 5e-5),
    "rf_rk4_20": ("rk4", 20, 0.0, False, 5e-5),
    "rf_rk5_20": ("rk5", 20, 0.0, False, 5e-5),
    "rf_euler_shallow": ("euler", 20, 0.4, True, 5e-5),
}


@pytest.mark.parametrize("tag", sorted(RF_CASES))
def test_g5_rectified_flow_samplers(tag):
    algo, steps, t_start, shallow, tol = RF_CASES[tag]
    g = load("g5_samplers")
    bsz, t_len, nseed, _, _, _ = (int(v) for v in g[f"{tag}_meta"])
    r = od.RectifiedFlow(_sampler_net(), 32, 1, t_start=t_start, time_scale_factor=1000, spec_min=[-12.0],
                         spec_max=[0.0], use_shallow_diffusion=shallow)
    cond = synth.synth_normal((bsz, t_len, 256), nseed + 500)
    src = None
    if shallow:
        src = (synth.synth_normal((bsz, t_len, 32), nseed + 501) * 1.5 - 6.0).astype(np.float32)
    noise = synth.synth_normal((bsz, 1, 32, t_len), nseed)
    out = r.forward(cond, noise, src_spec=src, T_start_infer=t_start, sampling_algorithm=algo,
                    sampling_steps=steps)
    assert rel_err(out, g[f"{tag}_out"]) < tol, tag


def test_g5_full_size_wavenet_dpm20():
    g = load("g5_full_dpm20")
    bsz, t_len, nseed, _, cseed = (int(v) for v in g["meta"])
    args = dict(num_layers=20, num_channels=256, dilation_cycle_length=4)
    params = synth_params("wavenet", 128, 1, args, 42)
    fn = lambda x, t, c: ob.wavenet_forward(params, x, t, c, dilation_cycle_length=4)
    d = od.GaussianDiffusion(fn, 128, 1, spec_min=[-12.0], spec_max=[0.0])
    out = d.forward(synth.synth_normal((bsz, t_len, 256), cseed), synth.synth_normal((bsz, 1, 128, t_len), nseed),
                    diff_accelerator="dpm-solver", diff_speedup=50, K_step_infer=1000)
    assert rel_err(out, g["out"]) < 2e-4


def test_g5_lynxnet_samplers():
    g = load("g5_lynx")
    bsz, t_len, s_ddim, s_rf, cseed, wseed = (int(v) for v in g["meta"])
    largs = dict(num_layers=3, num_channels=256, expansion_factor=2, kernel_size=31,
                 activation="PReLU", strong_cond=True)
    params = synth_params("lynxnet", 128, 1, largs, wseed)
    fn = lambda x, t, c: ob.lynxnet_forward(params, x, t, c, activation="PReLU", strong_cond=True)
    cond = synth.synth_normal((bsz, t_len, 256), cseed)
    d = od.GaussianDiffusion(fn, 128, 1, spec_min=[-12.0], spec_max=[0.0])
    out = d.forward(cond, synth.synth_normal((bsz, 1, 128, t_len), s_ddim),
                    diff_accelerator="ddim", diff_speedup=50, K_step_infer=1000)
    assert rel_err(out, g["ddim20_out"]) < 2e-4
    r = od.RectifiedFlow(fn, 128, 1, spec_min=[-12.0], spec_max=[0.0])
    out2 = r.forward(cond, synth.synth_normal((bsz, 1, 128, t_len), s_rf),
                     sampling_algorithm="euler", sampling_steps=10)
    assert rel_err(out2, g["rf_euler10_out"]) < 5e-5


def test_unsupported_algorithms_raise():
    d = od.GaussianDiffusion(_sampler_net(), 32, 1, spec_min=[-12.0], spec_max=[0.0])
    x = np.zeros((1, 1, 32, 4), np.float32)
    c = np.zeros((1, 256, 4), np.float32)
    with pytest.raises(ValueError):
        d.inference(c, x, diff_speedup=10, diff_accelerator="nope")
    with pytest.raises(AssertionError):
        d.inference(c, x, diff_speedup=7, diff_accelerator="ddim")
    r = od.RectifiedFlow(_sampler_net(), 32, 1, spec_min=[-12.0], spec_max=[0.0])
    with pytest.raises(ValueError):
        r.inference(c, x, sampling_algorithm="nope")


# --------------------------------------------------------------------------- G6
def test_g6_norm_denorm_wrappers():
    g = load("g6_wrappers")
    d = od.GaussianDiffusion(None, 32, 1, spec_min=g["gd_smin"].tolist(), spec_max=g["gd_smax"].tolist())
    np.testing.assert_allclose(d.norm_spec(g["gd_mel"]), g["gd_norm"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(d.denorm_spec(g["gd_mel"]), g["gd_denorm"], rtol=1e-6, atol=1e-5)

    nf, smin, smax = od.repetitive_spec_ranges(-8.0, 8.0)
    p = od.GaussianDiffusion(None, 64, nf, spec_min=smin, spec_max=smax)
    np.testing.assert_allclose(od.pitch_norm(p, g["pitch_in"], 64, -12.0, 12.0), g["pitch_norm"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(od.pitch_denorm(p, g["pitch_x"], -12.0, 12.0), g["pitch_denorm"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(od.pitch_norm(p, g["pitch_in"], 64, -12.0, 12.0), g["rf_pitch_norm"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(od.pitch_denorm(p, g["pitch_x"], -12.0, 12.0), g["rf_pitch_denorm"], rtol=1e-5, atol=1e-5)

    ranges, clamps = [(-96.0, -12.0), (-96.0, -20.0)], [(-96.0, 0.0), None]
    nf, smin, smax = od.repetitive_spec_ranges([r[0] for r in ranges], [r[1] for r in ranges])
    m = od.GaussianDiffusion(None, 24, nf, spec_min=smin, spec_max=smax)
    np.testing.assert_allclose(od.multivar_norm(m, [g["mv_in0"], g["mv_in1"]], 24, clamps), g["mv_norm"],
                               rtol=1e-6, atol=1e-6)
    den = od.multivar_denorm(m, g["mv_x"], clamps)
    np.testing.assert_allclose(den[0], g["mv_denorm0"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(den[1], g["mv_denorm1"], rtol=1e-5, atol=1e-4)

    nf, smin, smax = od.repetitive_spec_ranges(-96.0, -12.0)
    m1 = od.GaussianDiffusion(None, 48, nf, spec_min=smin, spec_max=smax)
    np.testing.assert_allclose(od.multivar_norm(m1, [g["mv_in0"]], 48, [(-96.0, 0.0)]), g["mv1_norm"],
                               rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(od.multivar_denorm(m1, g["mv1_x"], [(-96.0, 0.0)])[0], g["mv1_denorm0"],
                               rtol=1e-5, atol=1e-4)


# --------------------------------------------------------------------------- G7 (section 8(f) rank 1)
from oracle import aux_decoder as oa  # noqa: E402

AUX_TAGS = ("default", "small", "k5")


def aux_params(g, tag):
    hsz, m, c, nl, ks, bsz, t_len, wseed = (int(v) for v in g[f"{tag}_meta"])
    shapes = synth.convnext_param_shapes(hsz, m, num_channels=c, num_layers=nl, kernel_size=ks, prefix="decoder.")
    return synth.synth_state_dict(shapes, seed=wseed), (hsz, m, bsz, t_len, wseed)


@pytest.mark.parametrize("tag", AUX_TAGS)
def test_g7_convnext_aux_decoder(tag):
    """ConvNeXt aux decoder restatement vs reference AuxDecoderAdaptor; tolerance 2e-5 of the output range."""
    g = load("g7_aux_decoder")
    params, (hsz, m, bsz, t_len, wseed) = aux_params(g, tag)
    cond = synth.synth_normal((bsz, t_len, hsz), wseed + 100)
    raw = oa.aux_adaptor_forward(params, cond, m, 1, g[f"{tag}_smin"], g[f"{tag}_smax"], infer=False)
    mel = oa.aux_adaptor_forward(params, cond, m, 1, g[f"{tag}_smin"], g[f"{tag}_smax"], infer=True)
    if tag == "default":
        raw, mel = raw[:, ::3], mel[:, ::3]
    assert rel_err(raw, g[f"{tag}_raw"]) < 2e-5
    assert rel_err(mel, g[f"{tag}_mel"]) < 2e-5


@pytest.mark.parametrize("tag", ("ddpm_dpm", "reflow_euler"))
def test_g7_acoustic_glue(tag):
    """aux decoder -> padding mask -> shallow diffusion -> padding mask (toplevel.py:84-105)."""
    g = load("g7_aux_decoder")
    params, (hsz, m, _, _, _) = aux_params(g, "small")
    bsz, t_len, nseed = (int(v) for v in g["glue_meta"])
    smin, smax = g["glue_smin"], g["glue_smax"]
    net_params = synth_params("wavenet", m, 1, SN_ARGS, 45)
    net = lambda x, t, c: ob.wavenet_forward(net_params, x, t, c, dilation_cycle_length=2)  # noqa: E731
    if tag == "ddpm_dpm":
        d = od.GaussianDiffusion(net, m, 1, timesteps=1000, k_step=400, spec_min=smin.tolist(),
                                 spec_max=smax.tolist(), use_shallow_diffusion=True)
        kw = dict(diff_accelerator="dpm-solver", diff_speedup=20, K_step_infer=400)
    else:
        d = od.RectifiedFlow(net, m, 1, t_start=0.4, time_scale_factor=1000, spec_min=smin.tolist(),
                             spec_max=smax.tolist(), use_shallow_diffusion=True)
        kw = dict(T_start_infer=0.4, sampling_algorithm="euler", sampling_steps=20)
    cond = synth.synth_normal((bsz, t_len, hsz), 7500)
    noise = synth.synth_normal((bsz, 1, m, t_len), nseed)
    aux, mel = oa.acoustic_infer(params, d, cond, g["glue_mel2ph"], noise, smin, smax, m, **kw)
    assert rel_err(aux, g[f"glue_{tag}_aux"]) < 2e-5
    assert rel_err(mel, g[f"glue_{tag}_mel"]) < 2e-4
    assert np.all(mel[0, 41:] == 0) and np.all(mel[1, 48:] == 0)


# --------------------------------------------------------------------------- G8 (section 8(f) rank 2)
from oracle import encoder as oe  # noqa: E402

ENC_SYNTH = {
    "default": dict(), "padded": dict(),
    "full": dict(num_spk=3, num_lang=2, variances=("energy", "breathiness"), key_shift=True, speed=True),
    "k9": dict(),
    "relpos": dict(rope=False), "nopos": dict(rope=False), "sinpos": dict(rope=False, sinpos=True),
    "relu": dict(), "swish": dict(), "swiglu": dict(ffn_act="swiglu"),
}
ENC_POS = {"relpos": "rel", "nopos": "none", "sinpos": "sin"}
ENC_ACT = {"relu": "relu", "swish": "swish", "swiglu": "swiglu"}        # ffn_act: gelu unless listed


def enc_case(g, tag):
    vocab, hidden, layers, heads, ks, bsz, t_txt, t_mel, wseed = (int(v) for v in g[f"{tag}_meta"])
    kw = dict(hidden_size=hidden, enc_layers=layers, num_heads=heads, ffn_kernel_size=ks)
    kw.update(ENC_SYNTH[tag])
    params = synth.synth_state_dict(synth.fs2_acoustic_param_shapes(vocab, **kw), seed=wseed)
    extra = {k: g[f"{tag}_{k}"] for k in ("key_shift", "speed", "energy", "breathiness", "languages", "spk_embed_id")
             if f"{tag}_{k}" in g}
    return params, heads, g[f"{tag}_tokens"], g[f"{tag}_mel2ph"], g[f"{tag}_f0"], extra


@pytest.mark.parametrize("tag", sorted(ENC_SYNTH))
def test_g8_fs2_acoustic_encoder(tag):
    """FastSpeech2Acoustic (rotary configuration) restatement vs the reference; tolerance 2e-5 of the output range
    (4 transformer layers: LayerNorm, RoPE attention, k-tap FFN; padded batches; every optional embedding)."""
    g = load("g8_encoder")
    params, heads, tokens, mel2ph, f0, extra = enc_case(g, tag)
    cond = oe.fs2_acoustic_forward(params, tokens, mel2ph, f0, num_heads=heads, pos=ENC_POS.get(tag, "rope"),
                                   ffn_act=ENC_ACT.get(tag, "gelu"), **extra)
    want = g[f"{tag}_cond"]
    if tag in ("default", "padded", "relpos", "nopos", "sinpos"):
        cond = cond[:, ::2]
    assert cond.shape == want.shape
    assert rel_err(cond, want) < 2e-5
    if tag == "padded":       # the case really has padded tokens and padded frames
        assert (tokens == 0).any() and (mel2ph == 0).any()


# --------------------------------------------------------------------------- G9: tokens -> mel
def g9_parts(g):
    vocab, m, bsz, t_txt, t_mel, nseed = (int(v) for v in g["meta"])
    fs2 = synth.synth_state_dict(synth.fs2_acoustic_param_shapes(vocab, enc_layers=2), seed=9200)
    aux = synth.synth_state_dict(synth.convnext_param_shapes(256, m, num_channels=64, num_layers=2, prefix="decoder."),
                                 seed=9201)
    net = synth_params("wavenet", m, 1, SN_ARGS, 9202)
    return (vocab, m, bsz, t_txt, t_mel, nseed), fs2, aux, net


@pytest.mark.parametrize("tag", ("ddpm_dpm", "reflow_euler"))
def test_g9_acoustic_model_tokens_to_mel(tag):
    """The reference's top-level DiffSingerAcoustic (infer branch) restated as encoder -> aux decoder -> shallow loop."""
    g = load("g9_acoustic_model")
    (vocab, m, bsz, t_txt, t_mel, nseed), fs2, aux, netp = g9_parts(g)
    cond = oe.fs2_acoustic_forward(fs2, g["tokens"], g["mel2ph"], g["f0"], num_heads=2)
    net = lambda x, t, c: ob.wavenet_forward(netp, x, t, c, dilation_cycle_length=2)  # noqa: E731
    smin, smax = g["smin"], g["smax"]
    if tag == "ddpm_dpm":
        d = od.GaussianDiffusion(net, m, 1, timesteps=1000, k_step=400, spec_min=smin.tolist(), spec_max=smax.tolist(),
                                 use_shallow_diffusion=True)
        kw = dict(diff_accelerator="dpm-solver", diff_speedup=20, K_step_infer=400)
    else:
        d = od.RectifiedFlow(net, m, 1, t_start=0.4, time_scale_factor=1000, spec_min=smin.tolist(),
                             spec_max=smax.tolist(), use_shallow_diffusion=True)
        kw = dict(T_start_infer=0.4, sampling_algorithm="euler", sampling_steps=20)
    noise = synth.synth_normal((bsz, 1, m, t_mel), nseed)
    aux_mel, mel = oa.acoustic_infer(aux, d, cond, g["mel2ph"], noise, smin, smax, m, **kw)
    assert rel_err(aux_mel, g[f"{tag}_aux"]) < 2e-5
    assert rel_err(mel, g[f"{tag}_mel"]) < 2e-4


# --------------------------------------------------------------------------- G10 (section 8(f) rank 3)
from oracle import vocoder as ov  # noqa: E402

VOC_OVER = {
    "default": dict(),
    "small_rb2": dict(num_mels=32, upsample_rates=[4, 2, 2], upsample_kernel_sizes=[8, 4, 4], upsample_initial_channel=64,
                      resblock="2", resblock_kernel_sizes=[3, 5], resblock_dilation_sizes=[[1, 2], [2, 6]], hop_size=16),
    "mini_nsf": dict(mini_nsf=True),
    "mini_small": dict(mini_nsf=True, num_mels=32, upsample_rates=[4, 4, 2], upsample_kernel_sizes=[8, 8, 4],
                       upsample_initial_channel=128, resblock_kernel_sizes=[3, 7],
                       resblock_dilation_sizes=[[1, 3, 5], [1, 2, 3]], hop_size=32),
    "small_sigma": dict(num_mels=32, upsample_rates=[4, 2, 2], upsample_kernel_sizes=[8, 4, 4], upsample_initial_channel=64,
                        resblock="2", resblock_kernel_sizes=[3, 5], resblock_dilation_sizes=[[1, 2], [2, 6]], hop_size=16,
                        noise_sigma=0.3),
    "mini_sigma": dict(mini_nsf=True, num_mels=32, upsample_rates=[4, 4, 2], upsample_kernel_sizes=[8, 8, 4],
                       upsample_initial_channel=128, resblock_kernel_sizes=[3, 7],
                       resblock_dilation_sizes=[[1, 3, 5], [1, 2, 3]], hop_size=32, noise_sigma=0.2),
}
VOC_GAIN = 0.7


def voc_case(g, tag):
    bsz, t_len, wseed, upp = (int(v) for v in g[f"{tag}_meta"])
    h = dict(synth.NSF_HIFIGAN_DEFAULT)
    h.update(VOC_OVER[tag])
    params = synth.synth_state_dict(synth.nsf_hifigan_param_shapes(h), seed=wseed, gain=VOC_GAIN)
    mel = (synth.synth_normal((bsz, t_len, h["num_mels"]), wseed + 1) * 1.5 - 5.0).astype(np.float32)
    noise = synth.synth_normal((bsz, t_len * upp, 9), wseed + 3)
    pre = synth.synth_normal((bsz, h["upsample_initial_channel"], t_len), wseed + 4)       # used when noise_sigma > 0
    return h, params, mel, g[f"{tag}_f0"], g[f"{tag}_rand_ini"], noise, pre


@pytest.mark.parametrize("tag", sorted(VOC_OVER))
def test_g10_nsf_hifigan_generator(tag):
    """NSF-HiFiGAN generator restatement (sine source with injected phases/noise, transposed-conv upsampling, noise
    convs, ResBlock1/2, tanh) vs the reference; tolerance 5e-5 of the waveform range."""
    g = load("g10_vocoder")
    h, params, mel, f0, rand_ini, noise, pre = voc_case(g, tag)
    wav = ov.spec2wav(params, h, mel, f0, rand_ini, noise, pre_noise=pre)
    want = g[f"{tag}_wav"].reshape(-1)
    assert wav.shape == want.shape
    assert rel_err(wav, want) < 5e-5


# --------------------------------------------------------------------------- G12 (BASELINE config 5's callers)
import variance_cases as vc  # noqa: E402
from oracle import variance as ovar  # noqa: E402


def variance_case(g, tag, named_shapes=None):
    """hparams, weights (numpy, flat), inputs and the injected x_T of a G12 case."""
    hp = vc.case_hparams(tag)
    shapes = OrderedDict()
    for item in g[f"{tag}_params"]:
        name, shp = str(item).split(":")
        shapes[name] = tuple(int(s) for s in shp.split("x")) if shp else ()
    if named_shapes is not None:
        assert list(named_shapes.items()) == list(shapes.items()), "parameter names / shapes differ from the reference's"
    c = vc.CASES[tag]
    params = vc.synth_weights(shapes, c["seed"] + 1)
    inp = vc.case_inputs(tag)
    names = [n for n in ovar.VARIANCE_CHECKLIST if hp.get("predict_" + n)]
    t_len, bsz, seed = c["t_len"], c["bsz"], c["seed"] + 2
    noise = {}
    if hp["predict_pitch"]:
        noise["noise_pitch"] = synth.synth_normal((bsz, 1, hp["pitch_prediction_args"]["repeat_bins"], t_len), seed)
        seed += 1
    if names:
        rb = hp["variances_prediction_args"]["total_repeat_bins"] // len(names)
        noise["noise_var"] = synth.synth_normal((bsz, len(names), rb, t_len), seed)
    assert len(g[f"{tag}_randn"]) == len(noise)
    return hp, params, inp, noise, names


@pytest.mark.parametrize("tag", list(vc.CASES))
def test_g12_variance_model(tag):
    """DiffSingerVariance.forward(infer=True) restatement (word / phoneme encoder, duration predictor, rhythm + length
    regulators, melody encoder, retake embeddings, pitch and multi-variance denoisers) vs the reference."""
    g = load("g12_variance_model")
    hp, params, inp, noise, names = variance_case(g, tag)

    def make_fn(prefix, args):
        sub = ovar.sub(params, prefix)
        return lambda x, t, c: ob.wavenet_forward(sub, x, t, c, dilation_cycle_length=args["dilation_cycle_length"])

    variances = {n: inp.pop(n) for n in list(inp) if n in ovar.VARIANCE_CHECKLIST}
    dur, pitch, var = ovar.variance_model_forward(params, hp, make_fn, variances=variances, **inp, **noise)
    if hp["predict_dur"]:
        want = g[f"{tag}_dur"]
        assert np.abs(dur - want).max() < 2e-5 * max(1.0, np.abs(want).max())
        if f"{tag}_mel2ph" in g.files:
            aligned = ovar.rhythm_regulator(dur, inp["ph2word"], inp["word_dur"])
            assert np.array_equal(aligned, g[f"{tag}_dur_aligned"])
            assert np.array_equal(ovar.length_regulator(aligned), g[f"{tag}_mel2ph"])
    else:
        assert dur is None
    if hp["predict_pitch"]:
        want = g[f"{tag}_pitch"]
        assert (np.abs(want) < 11.99).mean() > 0.5          # mostly inside the clip range: the comparison means something
        assert np.abs(pitch - want).max() < 1e-4 * max(1.0, np.abs(want).max())
    else:
        assert pitch is None
    assert list(var) == names
    for n in names:
        want = g[f"{tag}_{n}"]
        assert np.abs(var[n] - want).max() < 1e-4 * np.abs(want).max()


def test_g5_config1_full_size_wavenet_pndm50():
    """BASELINE configs[0]: 20x256 WaveNet, one utterance, PNDM 1000 -> 50 (the reference's CPU-runnable case)."""
    g = load("g5_config1_pndm50")
    bsz, t_len, nseed, _, cseed = (int(v) for v in g["meta"])
    args = dict(num_layers=20, num_channels=256, dilation_cycle_length=4)
    params = synth_params("wavenet", 128, 1, args, 42)
    fn = lambda x, t, c: ob.wavenet_forward(params, x, t, c, dilation_cycle_length=4)
    d = od.GaussianDiffusion(fn, 128, 1, spec_min=[-12.0], spec_max=[0.0])
    out = d.forward(synth.synth_normal((bsz, t_len, 256), cseed), synth.synth_normal((bsz, 1, 128, t_len), nseed),
                    diff_accelerator="pndm", diff_speedup=20, K_step_infer=1000)
    assert rel_err(out, g["out"]) < 2e-4


# --------------------------------------------------------------------------- G16: the ONNX twins' runtime inputs
GD_ONNX = ["gd_steps30", "gd_steps7", "gd_depth037_steps11", "gd_depth06_steps50", "gd_depth1_steps20",
           "gd_depth0012_steps20"]
RF_ONNX = ["rf_steps20", "rf_depth05_steps13", "rf_depth09_steps9", "rf_depth1_steps10", "rf_depth0_steps5"]


@pytest.mark.parametrize("tag", GD_ONNX)
def test_g16_gaussian_diffusion_onnx_twin(tag):
    """GaussianDiffusionONNX.forward(condition, x_start, depth, steps) - factor-snapped speed-up, depth rounded down to
    a multiple (deployment/modules/diffusion.py:105-161): oracle against the reference class's own output."""
    g = load("g16_onnx_twins")
    t_len, nseed, n_randn, k_step, shallow, steps = (int(v) for v in g[f"{tag}_meta"])
    depth = float(g[f"{tag}_depth"])
    d = od.GaussianDiffusion(_sampler_net(), 32, 1, timesteps=1000, k_step=k_step, spec_min=[-12.0], spec_max=[0.0],
                             use_shallow_diffusion=bool(shallow))
    cond = synth.synth_normal((1, t_len, 256), nseed + 500)
    src = None if depth < 0 else (synth.synth_normal((1, t_len, 32), nseed + 501) * 1.5 - 6.0).astype(np.float32)
    noise = synth.synth_normal((1, 1, 32, t_len), nseed)
    step_noise = [synth.synth_normal((1, 1, 32, t_len), nseed + 1 + i) for i in range(n_randn - 1)]
    out = d.forward_onnx(cond, noise, x_start=src, depth=None if depth < 0 else depth, steps=steps, step_noise=step_noise)
    assert out.shape == g[f"{tag}_out"].shape
    assert rel_err(out, g[f"{tag}_out"]) < 1e-4, tag


def test_g16_onnx_plan_known_answers():
    d = od.GaussianDiffusion(None, 32, 1, timesteps=1000, k_step=400, spec_min=[-12.0], spec_max=[0.0],
                             use_shallow_diffusion=True)
    assert d.onnx_plan(30) == (400, 25) and d.onnx_plan(7) == (400, 125) and d.onnx_plan(5000) == (400, 1)
    assert d.onnx_plan(11, 0.37) == (363, 33) and d.onnx_plan(50, 0.6) == (400, 8) and d.onnx_plan(20, 0.012) == (12, 1)
    from diffsinger_amd import schedule
    factors = [i for i in range(1, 1001) if 1000 % i == 0]
    for steps, depth in [(30, None), (7, None), (5000, None), (11, 0.37), (50, 0.6), (20, 0.012), (20, 1.0), (3, 0.0005)]:
        assert schedule.onnx_ddpm_plan(1000, 400, factors, steps, depth) == d.onnx_plan(steps, depth)


@pytest.mark.parametrize("tag", RF_ONNX)
def test_g16_rectified_flow_onnx_twin(tag):
    g = load("g16_onnx_twins")
    t_len, nseed, _, shallow, steps = (int(v) for v in g[f"{tag}_meta"])
    depth, t_start = float(g[f"{tag}_depth"]), float(g[f"{tag}_tstart"])
    r = od.RectifiedFlow(_sampler_net(), 32, 1, t_start=t_start, time_scale_factor=1000, spec_min=[-12.0],
                         spec_max=[0.0], use_shallow_diffusion=bool(shallow))
    cond = synth.synth_normal((1, t_len, 256), nseed + 500)
    src = None if depth < 0 else (synth.synth_normal((1, t_len, 32), nseed + 501) * 1.5 - 6.0).astype(np.float32)
    noise = synth.synth_normal((1, 1, 32, t_len), nseed)
    out = r.forward_onnx(cond, noise, x_end=src, depth=None if depth < 0 else depth, steps=steps)
    assert rel_err(out, g[f"{tag}_out"]) < 5e-5, tag
