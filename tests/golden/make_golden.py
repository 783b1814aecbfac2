#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the imported reference.

Runs ONLY in the build container (it needs /root/reference, which never travels
to the GPU box).  It stubs the one missing third-party import (`lightning`,
pulled in by utils/training_utils.py but never executed on this path), imports
the reference's own `build_backbone`, `GaussianDiffusion`, `RectifiedFlow`,
`PitchDiffusion`, `MultiVarianceDiffusion`, `NoiseScheduleVP`, loads
deterministic synthetic weights (diffsinger_amd/synth.py) with strict=True and
records inputs' seeds and the reference outputs as small .npz files.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Fixtures are data (seeds, shapes, reference outputs); no reference source is
stored.  `torch.randn` is patched while the reference samplers run so that x_T
and the ancestral-DDPM step noise are the injected, seed-derived tensors that
the oracle and the HIP path are later fed.
"""
from __future__ import annotations

import copy
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)


def _install_lightning_stub():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    stub("lightning")
    stub("lightning.pytorch", LightningModule=_Dummy, Trainer=_Dummy, Callback=_Dummy)
    stub("lightning.fabric")
    stub("lightning.fabric.loggers")
    stub("lightning.fabric.loggers.tensorboard", _TENSORBOARD_AVAILABLE=False)
    stub("lightning.pytorch.callbacks", ModelCheckpoint=_Dummy, TQDMProgressBar=_Dummy)
    stub("lightning.pytorch.loggers", TensorBoardLogger=_Dummy)
    stub("lightning.pytorch.utilities")
    stub("lightning.pytorch.utilities.rank_zero", rank_zero_info=print,
         rank_zero_only=lambda f: f, rank_zero_debug=print)


_install_lightning_stub()
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from utils.hparams import hparams  # noqa: E402  (reference)
from modules.backbones import build_backbone  # noqa: E402  (reference)
from modules.commons.common_layers import SinusoidalPosEmb  # noqa: E402  (reference)
from modules.core import ddpm as ref_ddpm  # noqa: E402  (reference)
from modules.core import reflow as ref_reflow  # noqa: E402  (reference)
from inference import dpm_solver_pytorch as ref_dpm  # noqa: E402  (reference)
from inference import uni_pc as ref_unipc  # noqa: E402  (reference)

from diffsinger_amd import synth  # noqa: E402

torch.set_num_threads(8)
torch.manual_seed(0)

BASE_HP = dict(hidden_size=256, schedule_type="linear", use_shallow_diffusion=False,
               diff_speedup=10, diff_accelerator="ddim", infer=False,
               sampling_algorithm="euler", sampling_steps=20)


def set_hp(**kw):
    hparams.clear()
    hparams.update(BASE_HP)
    hparams.update(kw)


def to_t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def load_synth(module, kind, in_dims, n_feats, args, seed):
    shapes = synth.backbone_param_shapes(kind, in_dims, n_feats, hidden_size=hparams["hidden_size"], **args)
    sd = synth.synth_state_dict(shapes, seed=seed)
    module.load_state_dict({k: to_t(v) for k, v in sd.items()}, strict=True)
    module.eval()
    return synth.state_dict_digest(sd)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} KiB)")


class InjectRandn:
    """Replace torch.randn by a queue of seed-derived tensors for the duration of a block."""

    def __init__(self, seed0, patch_like=False):
        self.seed = seed0
        self.seeds = []
        self.patch_like = patch_like        # also torch.randn_like (the ONNX twin's p_sample draws with it)

    def __enter__(self):
        self._orig = torch.randn
        self._orig_like = torch.randn_like

        def fake(*size, **kw):
            if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
                size = tuple(size[0])
            arr = synth.synth_normal(tuple(int(s) for s in size), self.seed)
            self.seeds.append(self.seed)
            self.seed += 1
            return to_t(arr)

        torch.randn = fake
        if self.patch_like:
            torch.randn_like = lambda t, **kw: fake(tuple(t.shape))
        return self

    def __exit__(self, *exc):
        torch.randn = self._orig
        torch.randn_like = self._orig_like


# ----------------------------------------------------------------------------
def g1_posemb():
    out = {}
    for dim in (256, 192, 512):
        emb = SinusoidalPosEmb(dim)
        out[f"long_{dim}"] = emb(torch.tensor([0, 1, 499, 999])).numpy()
        out[f"float_{dim}"] = emb(torch.tensor([0.0, 499.5, 980.0200195, 0.001], dtype=torch.float32)).numpy()
    out["t_long"] = np.array([0, 1, 499, 999], dtype=np.int64)
    out["t_float"] = np.array([0.0, 499.5, 980.0200195, 0.001], dtype=np.float32)
    save("g1_posemb", **out)


WAVENET_CASES = [
    # name, in_dims, n_feats, args, weight seed, [(B, T, t values or None, t kind)], intermediates layers
    ("wn_acoustic", 128, 1, dict(num_layers=20, num_channels=256, dilation_cycle_length=4), 42,
     [(1, 1, "long"), (1, 7, "float"), (1, 96, "long"), (2, 130, "float")], (0, 3, 19)),
    ("wn_pitch", 64, 1, dict(num_layers=20, num_channels=256, dilation_cycle_length=5), 43,
     [(1, 7, "long"), (1, 40, "float")], ()),
    ("wn_multivar", 24, 2, dict(num_layers=10, num_channels=192, dilation_cycle_length=4), 44,
     [(2, 50, "long"), (3, 33, "one")], ()),
    ("wn_small", 32, 1, dict(num_layers=4, num_channels=64, dilation_cycle_length=2), 45,
     [(2, 50, "float")], ()),
    # a channel count that is not a multiple of 32 (the library runs it zero-padded to 256)
    ("wn_c250", 32, 1, dict(num_layers=5, num_channels=250, dilation_cycle_length=3), 46,
     [(2, 70, "float"), (1, 33, "long")], ()),
]

LYNX_CASES = [
    ("lx_default", 128, 1, dict(num_layers=6, num_channels=512, expansion_factor=2, kernel_size=31,
                                activation="PReLU", strong_cond=False), 52,
     [(1, 96, "long"), (2, 20, "one")]),
    ("lx_acoustic1024", 128, 1, dict(num_layers=6, num_channels=1024, expansion_factor=2, kernel_size=31,
                                     activation="PReLU", strong_cond=True), 53,
     [(1, 40, "float")]),
    ("lx_silu", 64, 1, dict(num_layers=2, num_channels=128, expansion_factor=2, kernel_size=31,
                            activation="SiLU", strong_cond=False), 54, [(2, 45, "float")]),
    ("lx_relu", 24, 2, dict(num_layers=2, num_channels=128, expansion_factor=1, kernel_size=7,
                            activation="ReLU", strong_cond=True), 55, [(2, 45, "long")]),
]


def make_t(kind, bsz, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    if kind == "long":
        return rng.integers(0, 1000, size=(bsz,)).astype(np.int64)
    if kind == "float":
        return (rng.random(size=(bsz,), dtype=np.float32) * np.float32(999.0)).astype(np.float32)
    if kind == "one":       # reflow-style [1] tensor broadcast over the batch
        return (rng.random(size=(1,), dtype=np.float32) * np.float32(1000.0)).astype(np.float32)
    raise KeyError(kind)


def g2_g3_backbones():
    set_hp()
    for name, in_dims, n_feats, args, wseed, cases, inter_layers in WAVENET_CASES:
        net = build_backbone(in_dims, n_feats, "wavenet", args)
        digest = load_synth(net, "wavenet", in_dims, n_feats, args, wseed)
        out = {"digest": np.array(digest), "weight_seed": np.array(wseed)}
        for ci, (bsz, t_len, tkind) in enumerate(cases):
            xs, cs, ts = 1000 + 10 * ci, 1001 + 10 * ci, 1002 + 10 * ci
            x = synth.synth_normal((bsz, n_feats, in_dims, t_len), xs)
            cond = synth.synth_normal((bsz, hparams["hidden_size"], t_len), cs)
            t = make_t(tkind, bsz, ts)
            hooks, inter = [], {}
            if inter_layers and t_len == 96:
                for l in inter_layers:
                    def hook(mod, inp, outp, l=l):
                        inter[f"c{ci}_x_after_{l}"] = outp[0].detach().numpy().copy()
                        inter[f"c{ci}_skip_{l}"] = outp[1].detach().numpy().copy()
                    hooks.append(net.residual_layers[l].register_forward_hook(hook))
            with torch.no_grad():
                y = net(to_t(x), to_t(t), to_t(cond)).numpy()
            for h in hooks:
                h.remove()
            out[f"c{ci}_meta"] = np.array([bsz, t_len, xs, cs, ts], dtype=np.int64)
            out[f"c{ci}_tkind"] = np.array(tkind)
            out[f"c{ci}_t"] = t
            out[f"c{ci}_out"] = y
            out.update(inter)
        save("g2_" + name, **out)

    for name, in_dims, n_feats, args, wseed, cases in LYNX_CASES:
        net = build_backbone(in_dims, n_feats, "lynxnet", args)
        digest = load_synth(net, "lynxnet", in_dims, n_feats, args, wseed)
        out = {"digest": np.array(digest), "weight_seed": np.array(wseed)}
        for ci, (bsz, t_len, tkind) in enumerate(cases):
            xs, cs, ts = 2000 + 10 * ci, 2001 + 10 * ci, 2002 + 10 * ci
            x = synth.synth_normal((bsz, n_feats, in_dims, t_len), xs)
            cond = synth.synth_normal((bsz, hparams["hidden_size"], t_len), cs)
            t = make_t(tkind, bsz, ts)
            with torch.no_grad():
                y = net(to_t(x), to_t(t), to_t(cond)).numpy()
            out[f"c{ci}_meta"] = np.array([bsz, t_len, xs, cs, ts], dtype=np.int64)
            out[f"c{ci}_tkind"] = np.array(tkind)
            out[f"c{ci}_t"] = t
            out[f"c{ci}_out"] = y
        save("g3_" + name, **out)


def g4_schedules():
    set_hp()
    d = ref_ddpm.GaussianDiffusion(
        32, 1, timesteps=1000, k_step=1000, backbone_type="wavenet",
        backbone_args=dict(num_layers=1, num_channels=16, dilation_cycle_length=1),
        spec_min=[-12.0], spec_max=[0.0])
    out = {}
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
              "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
              "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1",
              "posterior_mean_coef2"):
        out[k] = getattr(d, k).numpy()
    for tag, n_keep, steps in (("full", 1000, 50), ("full20", 1000, 20), ("shallow", 400, 20)):
        ns = ref_dpm.NoiseScheduleVP(schedule="discrete", betas=d.betas[:n_keep])
        nu = ref_unipc.NoiseScheduleVP(schedule="discrete", betas=d.betas[:n_keep])
        solver = ref_dpm.DPM_Solver(lambda x, t: x, ns, algorithm_type="dpmsolver++")
        ts = solver.get_time_steps("time_uniform", ns.T, 1.0 / ns.total_N, steps, "cpu")
        out[f"{tag}_log_alpha_array"] = ns.log_alpha_array.numpy()[0]
        out[f"{tag}_t_array"] = ns.t_array.numpy()[0]
        out[f"{tag}_unipc_log_alpha_array"] = nu.log_alpha_array.numpy()[0]
        out[f"{tag}_timesteps"] = ts.numpy()
        out[f"{tag}_lambda"] = torch.cat([ns.marginal_lambda(t) for t in ts]).numpy()
        out[f"{tag}_alpha"] = torch.cat([ns.marginal_alpha(t) for t in ts]).numpy()
        out[f"{tag}_sigma"] = torch.cat([ns.marginal_std(t) for t in ts]).numpy()
        out[f"{tag}_model_t"] = torch.stack([(t - 1.0 / ns.total_N) * ns.total_N for t in ts]).numpy()
    # cosine beta schedule buffers too
    set_hp(schedule_type="cosine")
    dc = ref_ddpm.GaussianDiffusion(
        32, 1, timesteps=1000, k_step=1000, backbone_type="wavenet",
        backbone_args=dict(num_layers=1, num_channels=16, dilation_cycle_length=1),
        spec_min=[-12.0], spec_max=[0.0])
    out["cosine_betas"] = dc.betas.numpy()
    out["cosine_alphas_cumprod"] = dc.alphas_cumprod.numpy()
    save("g4_schedules", **out)


SAMPLER_NET = dict(in_dims=32, n_feats=1, args=dict(num_layers=4, num_channels=64, dilation_cycle_length=2),
                   wseed=45)


def _build_gd(in_dims, n_feats, args, wseed, k_step=1000, btype="wavenet", spec_min=(-12.0,), spec_max=(0.0,)):
    d = ref_ddpm.GaussianDiffusion(in_dims, n_feats, timesteps=1000, k_step=k_step, backbone_type=btype,
                                   backbone_args=args, spec_min=list(spec_min), spec_max=list(spec_max))
    load_synth(d.denoise_fn, btype, in_dims, n_feats, args, wseed)
    return d


def g5_samplers():
    out = {}
    sn = SAMPLER_NET
    t_len, hsz = 50, 256

    def run_gd(tag, bsz, hp, k_step=1000, shallow=False, noise_seed=3000):
        set_hp(use_shallow_diffusion=shallow, **hp)
        d = _build_gd(sn["in_dims"], sn["n_feats"], sn["args"], sn["wseed"], k_step=k_step)
        cond = synth.synth_normal((bsz, t_len, hsz), noise_seed + 500)
        src = None
        if shallow:
            # a plausible mel in [spec_min, spec_max]
            src = (synth.synth_normal((bsz, t_len, sn["in_dims"]), noise_seed + 501) * 1.5 - 6.0).astype(np.float32)
        with InjectRandn(noise_seed) as inj, torch.no_grad():
            y = d(to_t(cond), src_spec=None if src is None else to_t(src), infer=True).numpy()
        out[f"{tag}_out"] = y
        out[f"{tag}_meta"] = np.array([bsz, t_len, noise_seed, len(inj.seeds), k_step, int(shallow)], dtype=np.int64)
        print(f"  {tag}: randn calls={len(inj.seeds)} out={y.shape} absmax={np.abs(y).max():.3f}")

    run_gd("ddim10", 2, dict(diff_accelerator="ddim", diff_speedup=10, K_step_infer=1000))
    run_gd("ddim100", 2, dict(diff_accelerator="ddim", diff_speedup=100, K_step_infer=1000))
    run_gd("pndm20", 1, dict(diff_accelerator="pndm", diff_speedup=20, K_step_infer=1000))
    run_gd("dpm20", 2, dict(diff_accelerator="dpm-solver", diff_speedup=50, K_step_infer=1000))
    run_gd("dpm50", 1, dict(diff_accelerator="dpm-solver", diff_speedup=20, K_step_infer=1000))
    run_gd("dpm5", 1, dict(diff_accelerator="dpm-solver", diff_speedup=200, K_step_infer=1000))
    run_gd("unipc20", 2, dict(diff_accelerator="unipc", diff_speedup=50, K_step_infer=1000))
    run_gd("unipc50", 1, dict(diff_accelerator="unipc", diff_speedup=20, K_step_infer=1000))
    run_gd("ddpm_shallow20", 2, dict(diff_accelerator="ddim", diff_speedup=1, K_step_infer=20),
           k_step=400, shallow=True)
    run_gd("dpm_shallow", 2, dict(diff_accelerator="dpm-solver", diff_speedup=20, K_step_infer=400),
           k_step=400, shallow=True)
    run_gd("ddim_shallow", 2, dict(diff_accelerator="ddim", diff_speedup=10, K_step_infer=200),
           k_step=400, shallow=True)

    def run_rf(tag, bsz, algo, steps, t_start=0.0, shallow=False, noise_seed=4000):
        set_hp(use_shallow_diffusion=shallow, sampling_algorithm=algo, sampling_steps=steps,
               T_start_infer=t_start)
        r = ref_reflow.RectifiedFlow(sn["in_dims"], sn["n_feats"], t_start=t_start, time_scale_factor=1000,
                                     backbone_type="wavenet", backbone_args=sn["args"],
                                     spec_min=[-12.0], spec_max=[0.0])
        load_synth(r.velocity_fn, "wavenet", sn["in_dims"], sn["n_feats"], sn["args"], sn["wseed"])
        cond = synth.synth_normal((bsz, t_len, hsz), noise_seed + 500)
        src = None
        if shallow:
            src = (synth.synth_normal((bsz, t_len, sn["in_dims"]), noise_seed + 501) * 1.5 - 6.0).astype(np.float32)
        with InjectRandn(noise_seed) as inj, torch.no_grad():
            y = r(to_t(cond), src_spec=None if src is None else to_t(src), infer=True).numpy()
        out[f"{tag}_out"] = y
        out[f"{tag}_meta"] = np.array([bsz, t_len, noise_seed, len(inj.seeds), steps, int(shallow)], dtype=np.int64)
        out[f"{tag}_tstart"] = np.array(t_start, dtype=np.float64)
        print(f"  {tag}: randn calls={len(inj.seeds)} out={y.shape} absmax={np.abs(y).max():.3f}")

    run_rf("rf_euler20", 2, "euler", 20)
    run_rf("rf_rk2_20", 2, "rk2", 20)
    run_rf("rf_rk4_20", 2, "rk4", 20)
    run_rf("rf_rk5_20", 1, "rk5", 20)
    run_rf("rf_euler_shallow", 2, "euler", 20, t_start=0.4, shallow=True)
    save("g5_samplers", **out)

    # one full-size acoustic WaveNet run: DPM-Solver++ 1000 -> 20, T=64, B=1
    set_hp(diff_accelerator="dpm-solver", diff_speedup=50, K_step_infer=1000)
    args = dict(num_layers=20, num_channels=256, dilation_cycle_length=4)
    d = _build_gd(128, 1, args, 42)
    cond = synth.synth_normal((1, 64, 256), 3600)
    with InjectRandn(3100) as inj, torch.no_grad():
        y = d(to_t(cond), infer=True).numpy()
    save("g5_full_dpm20", out=y, meta=np.array([1, 64, 3100, len(inj.seeds), 3600], dtype=np.int64))

    # LYNXNet (acoustic.yaml-like, smaller) under DDIM and reflow euler
    largs = dict(num_layers=3, num_channels=256, expansion_factor=2, kernel_size=31,
                 activation="PReLU", strong_cond=True)
    set_hp(diff_accelerator="ddim", diff_speedup=50, K_step_infer=1000)
    d = _build_gd(128, 1, largs, 56, btype="lynxnet")
    cond = synth.synth_normal((2, 40, 256), 3700)
    with InjectRandn(3200) as inj, torch.no_grad():
        y = d(to_t(cond), infer=True).numpy()
    set_hp(sampling_algorithm="euler", sampling_steps=10)
    r = ref_reflow.RectifiedFlow(128, 1, t_start=0.0, time_scale_factor=1000, backbone_type="lynxnet",
                                 backbone_args=largs, spec_min=[-12.0], spec_max=[0.0])
    load_synth(r.velocity_fn, "lynxnet", 128, 1, largs, 56)
    with InjectRandn(3300) as inj2, torch.no_grad():
        y2 = r(to_t(cond), infer=True).numpy()
    save("g5_lynx", ddim20_out=y, rf_euler10_out=y2,
         meta=np.array([2, 40, 3200, 3300, 3700, 56], dtype=np.int64))


def g6_wrappers():
    out = {}
    set_hp()
    args = dict(num_layers=1, num_channels=16, dilation_cycle_length=1)
    rng = np.random.Generator(np.random.PCG64(7))
    # plain GaussianDiffusion norm/denorm with per-bin spec_min/max
    smin = (-12.0 + rng.random(32)).tolist()
    smax = (0.0 + rng.random(32)).tolist()
    d = ref_ddpm.GaussianDiffusion(32, 1, backbone_type="wavenet", backbone_args=args,
                                   spec_min=smin, spec_max=smax)
    mel = (rng.standard_normal((2, 9, 32)) * 3 - 6).astype(np.float32)
    out["gd_smin"], out["gd_smax"], out["gd_mel"] = np.array(smin, np.float32), np.array(smax, np.float32), mel
    out["gd_norm"] = d.norm_spec(to_t(mel)).numpy()
    out["gd_denorm"] = d.denorm_spec(to_t(mel)).numpy()
    # PitchDiffusion
    p = ref_ddpm.PitchDiffusion(vmin=-8.0, vmax=8.0, cmin=-12.0, cmax=12.0, repeat_bins=64,
                                backbone_type="wavenet", backbone_args=args)
    pitch = (rng.standard_normal((2, 11)) * 8).astype(np.float32)
    out["pitch_in"] = pitch
    out["pitch_norm"] = p.norm_spec(to_t(pitch)).numpy()
    xr = rng.standard_normal((2, 11, 64)).astype(np.float32) * 2
    out["pitch_x"] = xr
    out["pitch_denorm"] = p.denorm_spec(to_t(xr)).numpy()
    # MultiVarianceDiffusion, F = 2 and F = 1
    ranges, clamps = [(-96.0, -12.0), (-96.0, -20.0)], [(-96.0, 0.0), None]
    m = ref_ddpm.MultiVarianceDiffusion(ranges=ranges, clamps=clamps, repeat_bins=24,
                                        backbone_type="wavenet", backbone_args=args)
    v0 = (rng.standard_normal((2, 11)) * 40 - 50).astype(np.float32)
    v1 = (rng.standard_normal((2, 11)) * 40 - 50).astype(np.float32)
    out["mv_in0"], out["mv_in1"] = v0, v1
    out["mv_norm"] = m.norm_spec([to_t(v0), to_t(v1)]).numpy()
    xm = rng.standard_normal((2, 2, 11, 24)).astype(np.float32) * 2
    out["mv_x"] = xm
    den = m.denorm_spec(to_t(xm))
    out["mv_denorm0"], out["mv_denorm1"] = den[0].numpy(), den[1].numpy()
    m1 = ref_ddpm.MultiVarianceDiffusion(ranges=[(-96.0, -12.0)], clamps=[(-96.0, 0.0)], repeat_bins=48,
                                         backbone_type="wavenet", backbone_args=args)
    out["mv1_norm"] = m1.norm_spec([to_t(v0)]).numpy()
    xm1 = rng.standard_normal((2, 11, 48)).astype(np.float32) * 2
    out["mv1_x"] = xm1
    out["mv1_denorm0"] = m1.denorm_spec(to_t(xm1))[0].numpy()
    # reflow twins share the maps; pin one
    set_hp()
    pr = ref_reflow.PitchRectifiedFlow(vmin=-8.0, vmax=8.0, cmin=-12.0, cmax=12.0, repeat_bins=64,
                                       backbone_type="wavenet", backbone_args=args)
    out["rf_pitch_norm"] = pr.norm_spec(to_t(pitch)).numpy()
    out["rf_pitch_denorm"] = pr.denorm_spec(to_t(xr)).numpy()
    save("g6_wrappers", **out)


# --------------------------------------------------------------------------- G7: aux decoder + acoustic glue
AUX_CASES = {
    # tag: (in_dims(hidden), out_dims, decoder args, B, T, weight seed)
    "default": (256, 128, dict(num_channels=512, num_layers=6, kernel_size=7, dropout_rate=0.1), 2, 150, 70),
    "small": (256, 32, dict(num_channels=64, num_layers=2, kernel_size=7, dropout_rate=0.1), 3, 37, 71),
    "k5": (192, 80, dict(num_channels=128, num_layers=3, kernel_size=5, dropout_rate=0.0), 1, 64, 72),
}


def g7_aux_decoder():
    from modules.aux_decoder import AuxDecoderAdaptor  # (reference)
    out = {}
    set_hp()
    for tag, (hsz, m, args, bsz, t_len, wseed) in AUX_CASES.items():
        rng = np.random.Generator(np.random.PCG64(wseed))
        smin = (-12.0 + rng.random(m)).astype(np.float32)
        smax = (0.0 + rng.random(m)).astype(np.float32)
        a = AuxDecoderAdaptor(hsz, m, 1, smin.tolist(), smax.tolist(), "convnext", dict(args))
        shapes = synth.convnext_param_shapes(hsz, m, num_channels=args["num_channels"], num_layers=args["num_layers"],
                                             kernel_size=args["kernel_size"], prefix="decoder.")
        sd = synth.synth_state_dict(shapes, seed=wseed)
        a.load_state_dict({k: to_t(v) for k, v in sd.items()}, strict=True)
        a.eval()
        cond = synth.synth_normal((bsz, t_len, hsz), wseed + 100)
        with torch.no_grad():
            raw = a(to_t(cond), infer=False).numpy()
            mel = a(to_t(cond), infer=True).numpy()
        # keep the default case small: a strided subset of the raw output plus the full denormed one for small nets
        out[f"{tag}_meta"] = np.array([hsz, m, args["num_channels"], args["num_layers"], args["kernel_size"],
                                       bsz, t_len, wseed], dtype=np.int64)
        out[f"{tag}_smin"], out[f"{tag}_smax"] = smin, smax
        out[f"{tag}_raw"] = raw[:, ::3] if tag == "default" else raw
        out[f"{tag}_mel"] = mel[:, ::3] if tag == "default" else mel
        print(f"  aux {tag}: raw absmax={np.abs(raw).max():.3f} mel range=({mel.min():.2f},{mel.max():.2f})")

    # the glue of DiffSingerAcoustic.forward(infer=True) after the encoder (toplevel.py:84-105), with the small nets
    sn = SAMPLER_NET
    hsz, m, args, _, _, wseed = AUX_CASES["small"]
    bsz, t_len = 2, 50
    rng = np.random.Generator(np.random.PCG64(99))
    smin = (-12.0 + rng.random(m)).astype(np.float32)
    smax = (0.0 + rng.random(m)).astype(np.float32)
    mel2ph = np.ones((bsz, t_len), dtype=np.int64)
    mel2ph[0, 41:] = 0
    mel2ph[1, 48:] = 0
    out["glue_smin"], out["glue_smax"], out["glue_mel2ph"] = smin, smax, mel2ph
    out["glue_meta"] = np.array([bsz, t_len, 7000], dtype=np.int64)

    def glue(tag, build_diff):
        a = AuxDecoderAdaptor(hsz, m, 1, smin.tolist(), smax.tolist(), "convnext", dict(args))
        shapes = synth.convnext_param_shapes(hsz, m, num_channels=args["num_channels"],
                                             num_layers=args["num_layers"], kernel_size=args["kernel_size"],
                                             prefix="decoder.")
        a.load_state_dict({k: to_t(v) for k, v in synth.synth_state_dict(shapes, seed=wseed).items()}, strict=True)
        a.eval()
        d = build_diff()
        cond = to_t(synth.synth_normal((bsz, t_len, hsz), 7500))
        mask = (to_t(mel2ph) > 0).float()[:, :, None]
        with InjectRandn(7000) as inj, torch.no_grad():
            aux = a(cond, infer=True)
            aux *= mask
            mel = d(cond, src_spec=aux, infer=True)
            mel *= mask
        out[f"glue_{tag}_aux"], out[f"glue_{tag}_mel"] = aux.numpy(), mel.numpy()
        out[f"glue_{tag}_nrandn"] = np.array(len(inj.seeds), dtype=np.int64)
        print(f"  glue {tag}: randn calls={len(inj.seeds)} mel absmax={mel.abs().max():.3f}")

    def build_gd():
        set_hp(use_shallow_diffusion=True, diff_accelerator="dpm-solver", diff_speedup=20, K_step_infer=400)
        d = ref_ddpm.GaussianDiffusion(m, 1, timesteps=1000, k_step=400, backbone_type="wavenet",
                                       backbone_args=sn["args"], spec_min=smin.tolist(), spec_max=smax.tolist())
        load_synth(d.denoise_fn, "wavenet", m, 1, sn["args"], sn["wseed"])
        return d

    def build_rf():
        set_hp(use_shallow_diffusion=True, sampling_algorithm="euler", sampling_steps=20, T_start_infer=0.4)
        r = ref_reflow.RectifiedFlow(m, 1, t_start=0.4, time_scale_factor=1000, backbone_type="wavenet",
                                     backbone_args=sn["args"], spec_min=smin.tolist(), spec_max=smax.tolist())
        load_synth(r.velocity_fn, "wavenet", m, 1, sn["args"], sn["wseed"])
        return r

    glue("ddpm_dpm", build_gd)
    glue("reflow_euler", build_rf)
    save("g7_aux_decoder", **out)


# --------------------------------------------------------------------------- G8: FastSpeech2 acoustic encoder -> condition
ENC_CASES = {
    # tag: (vocab, extra hparams, synth kwargs, B, T_txt, T_mel, weight seed)
    "default": (60, dict(), dict(), 1, 23, 180, 80),
    "padded": (60, dict(), dict(), 3, 17, 96, 81),
    "full": (45, dict(use_spk_id=True, num_spk=3, use_lang_id=True, num_lang=2, use_energy_embed=True,
                      use_breathiness_embed=True, use_key_shift_embed=True, use_speed_embed=True),
             dict(num_spk=3, num_lang=2, variances=("energy", "breathiness"), key_shift=True, speed=True), 2, 12, 70, 82),
    "k9": (30, dict(enc_ffn_kernel_size=9, enc_layers=2, hidden_size=128), dict(ffn_kernel_size=9, enc_layers=2,
                                                                               hidden_size=128), 2, 9, 40, 83),
    # pre-rotary checkpoints: RelPositionalEncoding + torch.nn.MultiheadAttention (use_rope false), or no positions at all
    "relpos": (40, dict(use_rope=False, rel_pos=True, enc_layers=2), dict(enc_layers=2, rope=False), 3, 15, 64, 84),
    "nopos": (40, dict(use_rope=False, use_pos_embed=False, enc_layers=2), dict(enc_layers=2, rope=False), 2, 11, 50, 85),
    "sinpos": (40, dict(use_rope=False, rel_pos=False, enc_layers=2), dict(enc_layers=2, rope=False, sinpos=True), 3, 13, 56, 86),
    # the other feed-forward activations of TransformerFFNLayer (common_layers.py:126-136); SwiGLU doubles ffn_1
    "relu": (40, dict(ffn_act="relu", enc_layers=2), dict(enc_layers=2), 2, 12, 48, 87),
    "swish": (40, dict(ffn_act="swish", enc_layers=2), dict(enc_layers=2), 2, 12, 48, 88),
    "swiglu": (40, dict(ffn_act="swiglu", enc_layers=2, enc_ffn_kernel_size=5), dict(enc_layers=2, ffn_kernel_size=5,
                                                                                 ffn_act="swiglu"), 2, 14, 52, 89),
}


def enc_inputs(tag, vocab, bsz, t_txt, t_mel, seed, n_lang=0, n_spk=0):
    """Deterministic phoneme tokens / durations / f0 (shared with the tests through the fixture)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    tokens = np.zeros((bsz, t_txt), dtype=np.int64)
    mel2ph = np.zeros((bsz, t_mel), dtype=np.int64)
    for b in range(bsz):
        n_tok = t_txt if b == 0 else int(rng.integers(max(2, t_txt // 2), t_txt + 1))
        tokens[b, :n_tok] = rng.integers(1, vocab, n_tok)
        n_frames = t_mel if b == 0 else int(rng.integers(max(n_tok, t_mel // 2), t_mel + 1))
        cuts = np.sort(rng.choice(np.arange(1, n_frames), n_tok - 1, replace=False))
        durs = np.diff(np.concatenate([[0], cuts, [n_frames]]))
        mel2ph[b, :n_frames] = np.repeat(np.arange(1, n_tok + 1), durs)
    f0 = (220.0 * 2.0 ** rng.uniform(-1, 1, (bsz, t_mel))).astype(np.float32)
    f0[mel2ph == 0] = 0.0
    extra = dict(
        key_shift=rng.uniform(-3, 3, (bsz, t_mel)).astype(np.float32),
        speed=rng.uniform(0.5, 2, (bsz, t_mel)).astype(np.float32),
        energy=rng.uniform(-60, -10, (bsz, t_mel)).astype(np.float32),
        breathiness=rng.uniform(-80, -20, (bsz, t_mel)).astype(np.float32),
        languages=(rng.integers(1, n_lang + 1, (bsz, t_txt)) * (tokens > 0)).astype(np.int64) if n_lang else None,
        spk_embed_id=rng.integers(0, n_spk, (bsz,)).astype(np.int64) if n_spk else None,
    )
    return tokens, mel2ph, f0, extra


def g8_encoder():
    from modules.fastspeech.acoustic_encoder import FastSpeech2Acoustic  # (reference)
    out = {}
    for tag, (vocab, hp, skw, bsz, t_txt, t_mel, wseed) in ENC_CASES.items():
        base = dict(hidden_size=256, enc_layers=4, enc_ffn_kernel_size=3, ffn_act="gelu", dropout=0.1, num_heads=2,
                    use_pos_embed=True, rel_pos=True, use_rope=True, use_spk_id=False, num_spk=1, use_lang_id=False,
                    num_lang=1)
        base.update(hp)
        set_hp(**base)
        m = FastSpeech2Acoustic(vocab)
        kw = dict(hidden_size=base["hidden_size"], enc_layers=base["enc_layers"], num_heads=base["num_heads"],
                  ffn_kernel_size=base["enc_ffn_kernel_size"])
        kw.update(skw)
        sd = synth.synth_state_dict(synth.fs2_acoustic_param_shapes(vocab, **kw), seed=wseed)
        m.load_state_dict({k: to_t(v) for k, v in sd.items()}, strict=True)
        m.eval()
        tokens, mel2ph, f0, ex = enc_inputs(tag, vocab, bsz, t_txt, t_mel, wseed + 1000,
                                            n_lang=skw.get("num_lang", 0), n_spk=skw.get("num_spk", 0))
        kwargs = {}
        if skw.get("key_shift"):
            kwargs["key_shift"] = to_t(ex["key_shift"])
        if skw.get("speed"):
            kwargs["speed"] = to_t(ex["speed"])
        for v in skw.get("variances", ()):
            kwargs[v] = to_t(ex[v])
        if skw.get("num_lang"):
            kwargs["languages"] = to_t(ex["languages"])
        if skw.get("num_spk"):
            kwargs["spk_embed_id"] = to_t(ex["spk_embed_id"])
        with torch.no_grad():
            cond = m(to_t(tokens), to_t(mel2ph), to_t(f0), **kwargs).numpy()
        out[f"{tag}_meta"] = np.array([vocab, base["hidden_size"], base["enc_layers"], base["num_heads"],
                                       base["enc_ffn_kernel_size"], bsz, t_txt, t_mel, wseed], dtype=np.int64)
        out[f"{tag}_tokens"], out[f"{tag}_mel2ph"], out[f"{tag}_f0"] = tokens, mel2ph, f0
        for k in ("key_shift", "speed", "energy", "breathiness", "languages", "spk_embed_id"):
            if k in kwargs or (k in ("energy", "breathiness") and k in skw.get("variances", ())):
                out[f"{tag}_{k}"] = ex[k]
        out[f"{tag}_cond"] = cond[:, ::2] if tag in ("default", "padded", "relpos", "nopos", "sinpos") else cond      # every other frame: half the bytes
        print(f"  enc {tag}: cond {cond.shape} absmax={np.abs(cond).max():.3f}")
    save("g8_encoder", **out)


# --------------------------------------------------------------------------- G9: DiffSingerAcoustic, tokens -> mel
def g9_acoustic_model():
    """The reference's own top-level acoustic model (modules/toplevel.py:32-105), infer branch, small nets:
    fs2 encoder -> ConvNeXt aux decoder -> mask -> shallow loop -> mask."""
    from modules.toplevel import DiffSingerAcoustic  # (reference)
    out = {}
    vocab, m_bins, bsz, t_txt, t_mel = 40, 32, 2, 14, 60
    enc_hp = dict(hidden_size=256, enc_layers=2, enc_ffn_kernel_size=3, ffn_act="gelu", dropout=0.1, num_heads=2,
                  use_pos_embed=True, rel_pos=True, use_rope=True, use_spk_id=False, num_spk=1, use_lang_id=False,
                  num_lang=1)
    rng = np.random.Generator(np.random.PCG64(123))
    smin = (-12.0 + rng.random(m_bins)).astype(np.float32)
    smax = (0.0 + rng.random(m_bins)).astype(np.float32)
    aux_args = dict(num_channels=64, num_layers=2, kernel_size=7, dropout_rate=0.1)
    common = dict(use_shallow_diffusion=True, spec_min=smin.tolist(), spec_max=smax.tolist(),
                  shallow_diffusion_args=dict(aux_decoder_arch="convnext", aux_decoder_args=aux_args,
                                              val_gt_start=False, train_aux_decoder=True, train_diffusion=True,
                                              aux_decoder_grad=0.1),
                  backbone_type="wavenet", backbone_args=SAMPLER_NET["args"], timesteps=1000, K_step=400,
                  T_start=0.4, time_scale_factor=1000)
    tokens, mel2ph, f0, _ = enc_inputs("g9", vocab, bsz, t_txt, t_mel, 9001)
    out["tokens"], out["mel2ph"], out["f0"] = tokens, mel2ph, f0
    out["smin"], out["smax"] = smin, smax
    out["meta"] = np.array([vocab, m_bins, bsz, t_txt, t_mel, 9100], dtype=np.int64)
    for tag, hp in (("ddpm_dpm", dict(diffusion_type="ddpm", diff_accelerator="dpm-solver", diff_speedup=20,
                                      K_step_infer=400)),
                    ("reflow_euler", dict(diffusion_type="reflow", sampling_algorithm="euler", sampling_steps=20,
                                          T_start_infer=0.4))):
        set_hp(**enc_hp, **common, **hp)
        model = DiffSingerAcoustic(vocab, m_bins)
        sd = dict(model.state_dict())
        fs2 = synth.synth_state_dict(synth.fs2_acoustic_param_shapes(vocab, enc_layers=2), seed=9200)
        sd.update({"fs2." + k: to_t(v) for k, v in fs2.items()})
        aux = synth.synth_state_dict(synth.convnext_param_shapes(256, m_bins, num_channels=64, num_layers=2,
                                                                 prefix="aux_decoder.decoder."), seed=9201)
        sd.update({k: to_t(v) for k, v in aux.items()})
        fn = "denoise_fn" if hp["diffusion_type"] == "ddpm" else "velocity_fn"
        net = synth.synth_state_dict(synth.backbone_param_shapes("wavenet", m_bins, 1, hidden_size=256,
                                                                 **SAMPLER_NET["args"]), seed=9202)
        sd.update({f"diffusion.{fn}.{k}": to_t(v) for k, v in net.items()})
        model.load_state_dict(sd, strict=True)
        model.eval()
        with InjectRandn(9100) as inj, torch.no_grad():
            res = model(to_t(tokens), to_t(mel2ph), to_t(f0), infer=True)
        out[f"{tag}_aux"], out[f"{tag}_mel"] = res.aux_out.numpy(), res.diff_out.numpy()
        print(f"  acoustic {tag}: randn calls={len(inj.seeds)} mel {res.diff_out.shape} absmax={res.diff_out.abs().max():.3f}")
    save("g9_acoustic_model", **out)


# --------------------------------------------------------------------------- G10: NSF-HiFiGAN generator, mel + f0 -> wav
VOC_CASES = {
    # tag: (config overrides, B, T, weight seed)
    "default": (dict(), 2, 10, 300),
    "small_rb2": (dict(num_mels=32, upsample_rates=[4, 2, 2], upsample_kernel_sizes=[8, 4, 4],
                       upsample_initial_channel=64, resblock="2", resblock_kernel_sizes=[3, 5],
                       resblock_dilation_sizes=[[1, 2], [2, 6]], hop_size=16), 2, 37, 301),
    "mini_nsf": (dict(mini_nsf=True), 2, 9, 302),
    "mini_small": (dict(mini_nsf=True, num_mels=32, upsample_rates=[4, 4, 2], upsample_kernel_sizes=[8, 8, 4],
                        upsample_initial_channel=128, resblock_kernel_sizes=[3, 7], resblock_dilation_sizes=[[1, 3, 5], [1, 2, 3]],
                        hop_size=32), 3, 41, 303),
    # noise_sigma > 0: sigma * randn_like(x) after conv_pre (models.py:272-273); the draw is synth_normal(seed + 4)
    "small_sigma": (dict(num_mels=32, upsample_rates=[4, 2, 2], upsample_kernel_sizes=[8, 4, 4],
                         upsample_initial_channel=64, resblock="2", resblock_kernel_sizes=[3, 5],
                         resblock_dilation_sizes=[[1, 2], [2, 6]], hop_size=16, noise_sigma=0.3), 2, 21, 304),
    "mini_sigma": (dict(mini_nsf=True, num_mels=32, upsample_rates=[4, 4, 2], upsample_kernel_sizes=[8, 8, 4],
                        upsample_initial_channel=128, resblock_kernel_sizes=[3, 7], resblock_dilation_sizes=[[1, 3, 5], [1, 2, 3]],
                        hop_size=32, noise_sigma=0.2), 2, 19, 305),
}
VOC_GAIN = 0.7


def g10_vocoder():
    from modules.nsf_hifigan.models import Generator  # (reference)
    from modules.nsf_hifigan.env import AttrDict  # (reference)
    out = {}
    for tag, (over, bsz, t_len, wseed) in VOC_CASES.items():
        h = dict(synth.NSF_HIFIGAN_DEFAULT)
        h.update(over)
        gen = Generator(AttrDict(h))
        gen.remove_weight_norm()
        sd = synth.synth_state_dict(synth.nsf_hifigan_param_shapes(h), seed=wseed, gain=VOC_GAIN)
        gen.load_state_dict({k: to_t(v) for k, v in sd.items()}, strict=True)
        gen.eval()
        upp = int(np.prod(h["upsample_rates"]))
        mel = (synth.synth_normal((bsz, t_len, h["num_mels"]), wseed + 1) * 1.5 - 5.0).astype(np.float32)   # log10 mel
        rng = np.random.Generator(np.random.PCG64(wseed + 2))
        f0 = (220.0 * 2.0 ** rng.uniform(-1, 1, (bsz, t_len))).astype(np.float32)
        f0[:, : t_len // 4] = 0.0                                         # an unvoiced stretch
        f0[:, t_len // 4] = 55.0                                          # a jump right after it (mini_nsf interpolates)
        rand_ini = rng.random(9).astype(np.float32)
        noise = synth.synth_normal((bsz, t_len * upp, 9), wseed + 3)
        orig_rand, orig_randn_like = torch.rand, torch.randn_like
        torch.rand = lambda *a, **k: to_t(rand_ini).reshape(1, 1, 9).clone()
        pre_noise = synth.synth_normal((bsz, h["upsample_initial_channel"], t_len), wseed + 4)
        torch.randn_like = lambda x, **k: to_t(noise if x.shape[-1] == 9 else pre_noise).clone()
        try:
            with torch.no_grad():
                c = 2.30259 * to_t(mel).transpose(2, 1)                   # vocoders/nsf_hifigan.py:59-64
                wav = gen(c, to_t(f0)).numpy()
        finally:
            torch.rand, torch.randn_like = orig_rand, orig_randn_like
        out[f"{tag}_meta"] = np.array([bsz, t_len, wseed, upp], dtype=np.int64)
        out[f"{tag}_f0"], out[f"{tag}_rand_ini"] = f0, rand_ini
        out[f"{tag}_wav"] = wav
        print(f"  vocoder {tag}: wav {wav.shape} absmax={np.abs(wav).max():.3f} std={wav.std():.4f}")
    save("g10_vocoder", **out)


def g5_config1_pndm50():
    """BASELINE configs[0] (the reference's own CPU-runnable case): the 20-layer WaveNet, one utterance, PNDM 1000 -> 50
    steps (ddpm.py:149-204,323-347), at full width; T = 48 keeps the fixture small."""
    set_hp(diff_accelerator="pndm", diff_speedup=20, K_step_infer=1000)
    args = dict(num_layers=20, num_channels=256, dilation_cycle_length=4)
    d = _build_gd(128, 1, args, 42)
    cond = synth.synth_normal((1, 48, 256), 3700)
    with InjectRandn(3200) as inj, torch.no_grad():
        y = d(to_t(cond), infer=True).numpy()
    print(f"  config 1 (PNDM 50): randn calls={len(inj.seeds)} out={y.shape} absmax={np.abs(y).max():.3f}")
    save("g5_config1_pndm50", out=y, meta=np.array([1, 48, 3200, len(inj.seeds), 3700], dtype=np.int64))



def time_reference_cpu():
    """Not a fixture: the reference's own PyTorch modules timed on THIS container's CPU cores at the headline
    configuration (WaveNet 20x256, DPM-Solver++ 1000 -> 50, B = 1, T = 1000) - the number DESIGN.md quotes beside the numpy
    port that bench.py times on the GPU box (the reference itself cannot travel there)."""
    import time
    set_hp(diff_accelerator="dpm-solver", diff_speedup=20, K_step_infer=1000)
    args = dict(num_layers=20, num_channels=256, dilation_cycle_length=4)
    d = _build_gd(128, 1, args, 42)
    cond = to_t(synth.synth_normal((1, 1000, 256), 0))
    with torch.no_grad():
        d(cond, infer=True)
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            d(cond, infer=True)
        dt = (time.perf_counter() - t0) / n
    print(f"reference (PyTorch {torch.__version__}, {torch.get_num_threads()} threads): {dt:.2f} s per utterance, "
          f"{1000 * 50 / dt / 1e3:.1f} k mel-frames/s per denoise step")



# --------------------------------------------------------------------------- G14: phoneme dictionary (token ids of a model)
G14_DICTS = {
    "zh": "a\ta\nai\tai\nba\tb a\nshi\tsh ir\nn\tn\nyu\ty v\n",
    "ja": "a\ta\nka\tk a\nshi\tsh i\nn\tN\ntsu\tts u\ncl\tcl\n",
}
G14_CASES = {
    "multi": dict(langs=["zh", "ja"], extra=["EP", "ja/vf", "GlottalStop"],
                  merged=[["zh/a", "ja/a"], ["zh/sh", "ja/sh"], ["ja/N", "zh/n"], ["zh/n", "AP"], ["zh/b", "zh/b"], ["zh/y", "zh/v"]]),
    "single": dict(langs=["zh"], extra=["EP"], merged=[["zh/y", "v"], ["a", "ai"]]),
}


def g14_phoneme_dictionary():
    """The reference's PhonemeDictionary (utils/phoneme_utils.py:10-176) on two small synthetic pronunciation dictionaries:
    multilingual naming, extra phonemes, merged (also overlapping and cross-lingual) groups; ids, aliases, encode / decode."""
    import json
    import pathlib
    import shutil
    from utils.phoneme_utils import PhonemeDictionary  # (reference)
    work = pathlib.Path(HERE) / "_g14_work"
    work.mkdir(exist_ok=True)
    out = {"dicts": G14_DICTS, "cases": {}}
    try:
        for lang, text in G14_DICTS.items():
            (work / f"{lang}.txt").write_text(text, encoding="utf8")
        for tag, c in G14_CASES.items():
            d = PhonemeDictionary({l: work / f"{l}.txt" for l in c["langs"]}, extra_phonemes=c["extra"], merged_groups=c["merged"])
            sent = "a sh ir N" if tag == "multi" else "a sh ir n y"
            out["cases"][tag] = dict(
                config=c, vocab_size=d.vocab_size, phone_to_id=d._phone_to_id,
                id_to_phone=[list(p) if isinstance(p, tuple) else p for p in d._id_to_phone],
                cross_lingual=sorted(d.cross_lingual_phonemes),
                encode_zh=d.encode("a sh ir n ja/k EP AP", lang="zh") if tag == "multi" else d.encode(sent),
                encode_ja=d.encode("a sh i N zh/b", lang="ja") if tag == "multi" else None,
                decode=[d.decode(range(1, d.vocab_size), lang=l) for l in ([None, "zh", "ja"] if tag == "multi" else [None])],
                decode_groups=[list(p) if isinstance(p, tuple) else p for p in
                               (d.decode_one(i, scalar=False) for i in range(1, d.vocab_size))])
            print(f"  phoneme dictionary {tag}: vocab {d.vocab_size}, cross-lingual {sorted(d.cross_lingual_phonemes)}")
        with open(os.path.join(HERE, "g14_phoneme_dictionary.json"), "w", encoding="utf8") as f:
            json.dump(out, f, ensure_ascii=False, indent=1, sort_keys=True)
    finally:
        shutil.rmtree(work, ignore_errors=True)



def g15_infer_utils():
    """The reference's `parse_commandline_spk_mix` on a set of --spk strings, and `trans_key` on a small project.  The
    latter calls librosa (absent): note_to_midi / midi_to_note come from diffsinger_amd here, so the NOTE NAMES of this
    fixture are pinned by known answers only (tests/test_harness.py); the f0 arithmetic and the handling of rests are the
    reference's."""
    import json
    import types
    from diffsinger_amd import harness as hz
    from diffsinger_amd import variance_harness as vh
    lib = sys.modules.get("librosa") or types.ModuleType("librosa")
    lib.__path__ = []
    lib.note_to_midi = lambda n, round_midi=True: (int(round(vh.note_to_midi(n))) if round_midi else vh.note_to_midi(n))
    lib.midi_to_note = lambda m, unicode=True: hz.midi_to_note(m)
    sys.modules["librosa"] = lib
    from utils.infer_utils import parse_commandline_spk_mix, trans_key  # (reference)
    mixes = ["opencpop", "a|b", "a:0.5|b:0.5", "a:0.3|b", "a:0.2|b|c", "a:2|b:6", "x_1:0.25|y-2:0.25|z"]
    bad = ["a|a", "a:0.7|b:0.6|c", "a:|b", "a b", "a:0|b:0"]
    proj = [dict(note_seq="C4 rest A#3 Db4+20 B3", f0_seq="220.0 0.0 261.6 440.05"), dict(note_seq="rest G9 C-1")]
    out = dict(mixes={m: parse_commandline_spk_mix(m) for m in mixes}, bad=bad, project=proj, shifted={})
    for key in (-13, -1, 0, 2, 12):
        out["shifted"][str(key)] = trans_key(copy.deepcopy(proj), key)
    for m in bad:
        try:
            parse_commandline_spk_mix(m)
            raise SystemExit(f"expected an assertion for {m!r}")
        except AssertionError:
            pass
    with open(os.path.join(HERE, "g15_infer_utils.json"), "w", encoding="utf8") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(f"  infer utils: {len(mixes)} mixes, {len(out['shifted'])} key shifts")



# --------------------------------------------------------------------------- G12: DiffSingerVariance, tokens -> dur / pitch / variances
def g12_variance_model():
    """The reference's own top-level variance model (modules/toplevel.py:125-309), infer branch, small nets; configurations
    and seeded inputs in tests/variance_cases.py; weights = synth_state_dict over the NAME-sorted parameters."""
    sys.path.insert(0, os.path.dirname(HERE))
    import variance_cases as vc
    from modules.toplevel import DiffSingerVariance  # (reference)
    out = {}
    for tag, c in vc.CASES.items():
        hp = vc.case_hparams(tag)
        set_hp(**hp)
        model = DiffSingerVariance(c["vocab"])
        shapes = vc.sorted_param_shapes(model.named_parameters())
        sd = vc.synth_weights(shapes, c["seed"] + 1)
        model.load_state_dict({k: to_t(v) for k, v in sd.items()}, strict=False)
        model.eval()
        inp = vc.case_inputs(tag)
        kw = {k: (None if v is None else ({n: to_t(a) for n, a in v.items()} if isinstance(v, dict) else to_t(v)))
              for k, v in inp.items()}
        with InjectRandn(c["seed"] + 2) as inj, torch.no_grad():
            dur, pitch, var = model(infer=True, **kw)
        out[f"{tag}_params"] = np.array([f"{n}:{'x'.join(map(str, sh))}" for n, sh in shapes.items()])
        out[f"{tag}_randn"] = np.array(inj.seeds, dtype=np.int64)
        msg = f"  variance {tag}: {len(shapes)} tensors, randn calls={len(inj.seeds)}"
        if dur is not None:
            out[f"{tag}_dur"] = dur.numpy()
            msg += f" dur[0]={np.round(dur[0].numpy(), 2).tolist()}"
            if "word_dur" in inp and c["t_len"]:
                from modules.fastspeech.tts_modules import LengthRegulator, RhythmRegulator  # (reference)
                aligned = RhythmRegulator()(dur, kw["ph2word"], kw["word_dur"])
                out[f"{tag}_dur_aligned"] = aligned.numpy()
                out[f"{tag}_mel2ph"] = LengthRegulator()(aligned).numpy()
        if pitch is not None:
            out[f"{tag}_pitch"] = pitch.numpy()
            msg += f" pitch absmax={pitch.abs().max():.3f}"
        for n, v in (var or {}).items():
            out[f"{tag}_{n}"] = v.numpy()
            msg += f" {n}[{v.min():.1f},{v.max():.1f}]"
        print(msg)
    save("g12_variance_model", **out)



# --------------------------------------------------------------------------- G13: variance .ds harness (host-side wire format)
def g13_variance_harness():
    """The reference's own DiffSingerVarianceInfer (inference/ds_variance.py) - its real __init__, preprocess_input and
    run_inference - on the synthetic project of tests/variance_cases.py, around a stand-in model (FakeVarianceModel) so
    that no checkpoint is needed.  librosa is absent: its three functions the harness calls (note_to_midi, hz_to_midi,
    midi_to_hz) are supplied from diffsinger_amd.variance_harness, so THOSE are not pinned by this fixture (known-answer
    checks in tests/test_variance_harness.py); everything else is the reference's arithmetic."""
    import contextlib
    import io
    import json
    import pathlib
    import shutil
    import types
    sys.path.insert(0, os.path.dirname(HERE))
    import variance_cases as vc
    from diffsinger_amd import variance_harness as vh
    from diffsinger_amd.harness import SimplePhonemeTable
    lib = sys.modules.get("librosa") or types.ModuleType("librosa")
    lib.__path__ = []
    lib.note_to_midi = lambda n, round_midi=True: vh.note_to_midi(n)
    lib.hz_to_midi, lib.midi_to_hz = vh.hz_to_midi, vh.midi_to_hz
    filt = types.ModuleType("librosa.filters")
    filt.mel = lambda *a, **k: None
    sys.modules["librosa"] = lib
    sys.modules.setdefault("librosa.filters", filt)
    import inference.ds_variance as dv  # (reference)
    work = pathlib.Path(HERE) / "_g13_work"
    work.mkdir(exist_ok=True)
    try:
        with open(work / "spk_map.json", "w") as f:
            json.dump(vc.HARNESS_SPK, f)
        set_hp(work_dir=str(work), **vc.HARNESS_HP)
        dv.load_phoneme_dictionary = lambda: SimplePhonemeTable(vc.HARNESS_PHONES)
        segs = vc.make_variance_segments()
        with open(os.path.join(HERE, "g13_variance_segments.ds"), "w", encoding="utf8") as f:
            json.dump(segs, f, indent=1)
        out = {}
        for mode, predictions in (("auto", set()), ("pitch_only", {"pitch"}), ("dur_energy", {"dur", "energy"})):
            model = vc.FakeVarianceModel()

            class Infer(dv.DiffSingerVarianceInfer):
                def build_model(self, ckpt_steps=None):
                    return model

            infer = Infer(device="cpu", predictions=predictions)
            recorded = []
            orig = infer.preprocess_input

            def spy(param, idx=0, load_dur=False, load_pitch=False, _orig=orig, _rec=recorded):
                batch = _orig(param, idx=idx, load_dur=load_dur, load_pitch=load_pitch)
                _rec.append((load_dur, load_pitch, batch))
                return batch

            infer.preprocess_input = spy
            with contextlib.redirect_stdout(io.StringIO()):
                infer.run_inference(copy.deepcopy(segs), out_dir=work, title=mode, num_runs=1, seed=11)
            with open(work / f"{mode}.ds", encoding="utf8") as f:
                done = json.load(f)
            with open(os.path.join(HERE, f"g13_out_{mode}.ds"), "w", encoding="utf8") as f:
                json.dump(done, f, indent=1)
            out[f"{mode}_calls"] = np.array(json.dumps(model.calls))
            for i, (ld, lp, batch) in enumerate(recorded):
                out[f"{mode}_seg{i}_load"] = np.array([ld, lp])
                for k, v in batch.items():
                    if v is not None:
                        out[f"{mode}_seg{i}_{k}"] = v.numpy()
            print(f"  variance harness {mode}: {len(recorded)} segments, loads {[(a, b) for a, b, _ in recorded]}")
        out["smooth_kernel"] = infer.smooth.weight.data.numpy().reshape(-1)
        save("g13_variance_harness", **out)
    finally:
        shutil.rmtree(work, ignore_errors=True)



# --------------------------------------------------------------------------- G11: .ds harness (host-side wire format)
def make_ds_segments():
    """A synthetic three-segment project in the .ds wire format (written next to the fixtures as g11_segments.ds)."""
    rng = np.random.Generator(np.random.PCG64(1100))
    phones = ["a", "b", "c", "d", "e", "SP", "AP"]
    segs = []
    offset = 0.0
    for i, n_ph in enumerate((6, 9, 5)):
        ph = [phones[int(k)] for k in rng.integers(0, len(phones), n_ph)]
        dur = rng.uniform(0.05, 0.4, n_ph).round(4)
        total = float(dur.sum())
        n_f0 = int(total / 0.005) + 1
        f0 = (220.0 * 2.0 ** rng.uniform(-0.5, 0.5, n_f0)).round(1)
        seg = {"offset": round(offset, 3), "ph_seq": " ".join(ph), "ph_dur": " ".join(str(v) for v in dur),
               "f0_seq": " ".join(str(v) for v in f0), "f0_timestep": "0.005",
               "energy": " ".join(str(v) for v in rng.uniform(-60, -10, n_f0 // 2).round(2)), "energy_timestep": "0.01",
               "seed": 100 + i}
        if i == 0:
            seg["gender"] = -0.5
            seg["spk_mix"] = {"alice": 0.25, "bob": 0.75}
        elif i == 1:
            seg["gender"] = " ".join(str(v) for v in rng.uniform(-1, 1, 40).round(3))
            seg["gender_timestep"] = str(round(total / 39, 5))
            seg["velocity"] = " ".join(str(v) for v in rng.uniform(0.3, 2.5, 30).round(3))
            seg["velocity_timestep"] = str(round(total / 29, 5))
            seg["spk_mix"] = {"alice": " ".join(str(v) for v in rng.uniform(0, 1, 25).round(3)), "bob": 0.5}
            seg["spk_mix_timestep"] = str(round(total / 24, 5))
        else:
            seg["spk_mix"] = {"bob": 1.0}
        segs.append(seg)
        offset += total * 0.9                       # the next segment overlaps the tail of this one: cross-fade
    return segs


def g11_harness():
    import json
    import types
    lib = types.ModuleType("librosa")
    lib.__path__ = []
    filt = types.ModuleType("librosa.filters")
    filt.mel = lambda *a, **k: None
    sys.modules.setdefault("librosa", lib)
    sys.modules.setdefault("librosa.filters", filt)
    from inference.ds_acoustic import DiffSingerAcousticInfer  # (reference)
    from modules.fastspeech.tts_modules import LengthRegulator  # (reference)
    from utils.infer_utils import cross_fade, resample_align_curve  # (reference)
    from diffsinger_amd.harness import SimplePhonemeTable
    out = {}
    segs = make_ds_segments()
    with open(os.path.join(HERE, "g11_segments.ds"), "w", encoding="utf8") as f:
        json.dump(segs, f, indent=1)
    set_hp(hop_size=512, audio_sample_rate=44100, use_spk_id=True, use_lang_id=False, use_energy_embed=True,
           use_key_shift_embed=True, use_speed_embed=True,
           augmentation_args=dict(random_pitch_shifting=dict(range=[-5.0, 5.0]), random_time_stretching=dict(range=[0.5, 2.0])))
    fake = types.SimpleNamespace(device="cpu", timestep=512 / 44100, lang_map={}, spk_map={"alice": 0, "bob": 1, "carol": 2},
                                 phoneme_dictionary=SimplePhonemeTable(["a", "b", "c", "d", "e"]),
                                 variances_to_embed={"energy"}, lr=LengthRegulator())
    fake.load_speaker_mix = types.MethodType(DiffSingerAcousticInfer.load_speaker_mix, fake)
    import contextlib, io
    for i, seg in enumerate(segs):
        with contextlib.redirect_stdout(io.StringIO()):
            batch = DiffSingerAcousticInfer.preprocess_input(fake, seg, idx=i)
        for k, v in batch.items():
            out[f"seg{i}_{k}"] = v.numpy()
    # the small pieces on their own
    rng = np.random.Generator(np.random.PCG64(1101))
    dur = torch.from_numpy(rng.integers(0, 7, (3, 11)))
    pad = torch.from_numpy(rng.integers(0, 2, (3, 11)) * (np.arange(11)[None] > 7))
    out["lr_dur"], out["lr_pad"] = dur.numpy(), pad.numpy()
    out["lr_mel2ph"] = LengthRegulator()(dur, pad.bool()).numpy()
    out["lr_mel2ph_alpha"] = LengthRegulator()(dur, None, 1.3).numpy()
    pts = rng.uniform(0, 1, 57).astype(np.float32)
    out["rs_points"] = pts
    out["rs_long"] = resample_align_curve(pts, 0.01, 512 / 44100, 80)
    out["rs_short"] = resample_align_curve(pts, 0.01, 512 / 44100, 30)
    a, b = rng.standard_normal(1000), rng.standard_normal(700)
    out["cf_a"], out["cf_b"] = a, b
    out["cf_out"] = cross_fade(a, b, 820)
    save("g11_harness", **out)



def g16_onnx_twins():
    """The runtime inputs (`depth`, `steps`) of the ONNX deployment twins, from the reference's own
    GaussianDiffusionONNX / RectifiedFlowONNX (deployment/modules/diffusion.py:18-161, rectified_flow.py:12-68)
    with the same seeded weights and injected normals as G5."""
    from deployment.modules.diffusion import GaussianDiffusionONNX  # noqa: E402  (reference)
    from deployment.modules.rectified_flow import RectifiedFlowONNX  # noqa: E402  (reference)
    out = {}
    sn = SAMPLER_NET
    t_len, hsz = 50, 256

    def run_gd(tag, steps, depth=None, k_step=1000, shallow=False, noise_seed=6000):
        set_hp(use_shallow_diffusion=shallow)
        d = GaussianDiffusionONNX(sn["in_dims"], sn["n_feats"], timesteps=1000, k_step=k_step, backbone_type="wavenet",
                                  backbone_args=sn["args"], spec_min=[-12.0], spec_max=[0.0])
        load_synth(d.denoise_fn, "wavenet", sn["in_dims"], sn["n_feats"], sn["args"], sn["wseed"])
        cond = synth.synth_normal((1, t_len, hsz), noise_seed + 500)
        src = None
        if depth is not None:
            src = (synth.synth_normal((1, t_len, sn["in_dims"]), noise_seed + 501) * 1.5 - 6.0).astype(np.float32)
        with InjectRandn(noise_seed, patch_like=True) as inj, torch.no_grad():
            y = d(to_t(cond), x_start=None if src is None else to_t(src),
                  depth=None if depth is None else torch.tensor(depth, dtype=torch.float32), steps=steps).numpy()
        out[f"{tag}_out"] = y
        out[f"{tag}_meta"] = np.array([t_len, noise_seed, len(inj.seeds), k_step, int(shallow), steps], dtype=np.int64)
        out[f"{tag}_depth"] = np.array(-1.0 if depth is None else depth, dtype=np.float64)
        print(f"  {tag}: randn calls={len(inj.seeds)} out={y.shape} absmax={np.abs(y).max():.3f}")

    run_gd("gd_steps30", 30)                                        # 1000 // 30 = 33 -> factor 25: 40 DDIM steps
    run_gd("gd_steps7", 7)                                          # 142 -> factor 125: 8 steps
    run_gd("gd_depth037_steps11", 11, depth=0.37, k_step=400, shallow=True)     # 370 // 11 = 33 -> depth 363, 11 steps
    run_gd("gd_depth06_steps50", 50, depth=0.6, k_step=400, shallow=True)       # capped at k_step 400, speed-up 8
    run_gd("gd_depth1_steps20", 20, depth=1.0)                      # depth == timesteps: starts from noise
    run_gd("gd_depth0012_steps20", 20, depth=0.012, k_step=400, shallow=True)   # 12 // 20 = 0 -> speed-up 1: ancestral

    def run_rf(tag, steps, depth=None, t_start=0.0, shallow=False, noise_seed=6500):
        set_hp(use_shallow_diffusion=shallow)
        r = RectifiedFlowONNX(sn["in_dims"], sn["n_feats"], t_start=t_start, time_scale_factor=1000,
                              backbone_type="wavenet", backbone_args=sn["args"], spec_min=[-12.0], spec_max=[0.0])
        load_synth(r.velocity_fn, "wavenet", sn["in_dims"], sn["n_feats"], sn["args"], sn["wseed"])
        cond = synth.synth_normal((1, t_len, hsz), noise_seed + 500)
        src = None
        if depth is not None:
            src = (synth.synth_normal((1, t_len, sn["in_dims"]), noise_seed + 501) * 1.5 - 6.0).astype(np.float32)
        with InjectRandn(noise_seed) as inj, torch.no_grad():
            y = r(to_t(cond), x_end=None if src is None else to_t(src),
                  depth=None if depth is None else torch.tensor(depth, dtype=torch.float32), steps=steps).numpy()
        out[f"{tag}_out"] = y
        out[f"{tag}_meta"] = np.array([t_len, noise_seed, len(inj.seeds), int(shallow), steps], dtype=np.int64)
        out[f"{tag}_depth"] = np.array(-1.0 if depth is None else depth, dtype=np.float64)
        out[f"{tag}_tstart"] = np.array(t_start, dtype=np.float64)
        print(f"  {tag}: randn calls={len(inj.seeds)} out={y.shape} absmax={np.abs(y).max():.3f}")

    run_rf("rf_steps20", 20)
    run_rf("rf_depth05_steps13", 13, depth=0.5, t_start=0.2, shallow=True)      # starts at max(0.5, 0.2)
    run_rf("rf_depth09_steps9", 9, depth=0.9, t_start=0.4, shallow=True)        # the model's own T_start wins: 0.4
    run_rf("rf_depth1_steps10", 10, depth=1.0)                                  # t_start 0: from noise
    run_rf("rf_depth0_steps5", 5, depth=0.0, t_start=0.3, shallow=True)         # t_start 1: the source itself, dt = 0
    save("g16_onnx_twins", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g23", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g5c1", "g14", "g15", "g16"]
    if "g1" in which:
        g1_posemb()
    if "g23" in which:
        g2_g3_backbones()
    if "g4" in which:
        g4_schedules()
    if "g5" in which:
        g5_samplers()
    if "g6" in which:
        g6_wrappers()
    if "g7" in which:
        g7_aux_decoder()
    if "g8" in which:
        g8_encoder()
    if "g9" in which:
        g9_acoustic_model()
    if "g10" in which:
        g10_vocoder()
    if "time" in which:
        time_reference_cpu()
    if "g14" in which:
        g14_phoneme_dictionary()
    if "g15" in which:
        g15_infer_utils()
    if "g5c1" in which:
        g5_config1_pndm50()
    if "g12" in which:
        g12_variance_model()
    if "g13" in which:
        g13_variance_harness()
    if "g16" in which:
        g16_onnx_twins()
    if "g11" in which:
        g11_harness()
