"""-m gpu parity of the shallow-diffusion aux decoder (dsd_aux_decode) and of the acoustic glue around the loop
 against the numpy oracle and the fixtures generated from the reference (G7).

Stated fp32 tolerance: 2e-5 of the output range for one decoder pass (oracle-vs-reference is <= 2e-6); the
glue cases run a 20-NFE shallow sampler afterwards and use the sampler tolerance 5e-4.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from diffsinger_amd import synth  # noqa: E402
from gpu_util import dev, rel_err, set_hp, synth_params  # noqa: E402
from oracle import aux_decoder as oa  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL_AUX = 2e-5
TOL_SAMPLER = 1.5e-5      # measured <= 1.3e-6 (profiles/r02_parity.json)


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    set_hp()


def make_adaptor(hsz, m, c, nl, ks, wseed, smin, smax):
    from diffsinger_amd.aux_decoder import AuxDecoderAdaptor
    a = AuxDecoderAdaptor(hsz, m, 1, list(map(float, smin)), list(map(float, smax)), "convnext",
                          dict(num_channels=c, num_layers=nl, kernel_size=ks, dropout_rate=0.1, unknown_key=1))
    shapes = synth.convnext_param_shapes(hsz, m, num_channels=c, num_layers=nl, kernel_size=ks, prefix="decoder.")
    params = synth.synth_state_dict(shapes, seed=wseed)
    a.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    return a.cuda().eval(), params


@pytest.mark.parametrize("tag", ("default", "small", "k5"))
def test_aux_decoder_vs_golden(tag):
    g = load("g7_aux_decoder")
    hsz, m, c, nl, ks, bsz, t_len, wseed = (int(v) for v in g[f"{tag}_meta"])
    a, _ = make_adaptor(hsz, m, c, nl, ks, wseed, g[f"{tag}_smin"], g[f"{tag}_smax"])
    cond = dev(synth.synth_normal((bsz, t_len, hsz), wseed + 100))
    with torch.no_grad():
        raw = a(cond, infer=False)
        mel = a(cond, infer=True)
    if tag == "default":
        raw, mel = raw[:, ::3], mel[:, ::3]
    assert rel_err(raw, g[f"{tag}_raw"]) < TOL_AUX
    assert rel_err(mel, g[f"{tag}_mel"]) < TOL_AUX
    a.decoder.release_native()


@pytest.mark.parametrize("bsz,t_len", [(1, 1), (1, 5), (3, 97), (2, 1000), (8, 640)])
def test_aux_decoder_vs_oracle_shapes(bsz, t_len):
    """Ragged / tiny / long inputs, both tile widths (64-frame tiles from B*T >= ~4k frames), [B,H,T]-strided cond."""
    rng = np.random.Generator(np.random.PCG64(5))
    smin, smax = (-12.0 + rng.random(128)).astype(np.float32), rng.random(128).astype(np.float32)
    a, params = make_adaptor(256, 128, 512, 6, 7, 81, smin, smax)
    cond = synth.synth_normal((bsz, t_len, 256), 900 + t_len)
    want = oa.aux_adaptor_forward(params, cond, 128, 1, smin, smax, infer=True)
    with torch.no_grad():
        got = a(dev(cond), infer=True)
        got_t = a(dev(np.ascontiguousarray(cond.transpose(0, 2, 1))).transpose(1, 2), infer=True)   # non-contiguous view
    assert rel_err(got, want) < TOL_AUX
    assert torch.equal(got, got_t)
    a.decoder.release_native()


def test_aux_decoder_errors():
    a, _ = make_adaptor(256, 32, 64, 2, 7, 71, [-12.0], [0.0])
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU path"):
        a.decoder(torch.zeros(1, 8, 256))
    with torch.no_grad(), pytest.raises(ValueError, match="must be"):
        a.decoder(torch.zeros(1, 8, 100).cuda())
    with pytest.raises(RuntimeError, match="inference-only"):
        a.decoder(torch.zeros(1, 8, 256).cuda())
    from diffsinger_amd.aux_decoder import build_aux_decoder
    with pytest.raises(KeyError):
        build_aux_decoder(256, 32, "nope", {})
    a.decoder.release_native()


@pytest.mark.parametrize("tag", ("ddpm_dpm", "reflow_euler"))
def test_acoustic_decoder_glue_vs_golden(tag):
    """AcousticDecoder = DiffSingerAcoustic after the encoder: aux decoder -> mask -> shallow loop -> mask."""
    from diffsinger_amd.toplevel import AcousticDecoder
    g = load("g7_aux_decoder")
    hsz, m, c, nl, ks, _, _, wseed = (int(v) for v in g["small_meta"])
    bsz, t_len, nseed = (int(v) for v in g["glue_meta"])
    sn_args = dict(num_layers=4, num_channels=64, dilation_cycle_length=2)
    hp = dict(use_shallow_diffusion=True, spec_min=g["glue_smin"].tolist(), spec_max=g["glue_smax"].tolist(),
              shallow_diffusion_args=dict(aux_decoder_arch="convnext", val_gt_start=False,
                                          aux_decoder_args=dict(num_channels=c, num_layers=nl, kernel_size=ks)),
              backbone_type="wavenet", backbone_args=sn_args, timesteps=1000, K_step=400, T_start=0.4,
              time_scale_factor=1000)
    if tag == "ddpm_dpm":
        set_hp(diffusion_type="ddpm", diff_accelerator="dpm-solver", diff_speedup=20, K_step_infer=400, **hp)
    else:
        set_hp(diffusion_type="reflow", sampling_algorithm="euler", sampling_steps=20, T_start_infer=0.4, **hp)
    model = AcousticDecoder(m)
    sd = {}
    aux_shapes = synth.convnext_param_shapes(hsz, m, num_channels=c, num_layers=nl, kernel_size=ks,
                                             prefix="aux_decoder.decoder.")
    sd.update(synth.synth_state_dict(aux_shapes, seed=wseed))
    fn = "denoise_fn" if tag == "ddpm_dpm" else "velocity_fn"
    for k, v in synth_params("wavenet", m, 1, sn_args, 45).items():
        sd[f"diffusion.{fn}.{k}"] = v
    full = dict(model.state_dict())          # schedule / spec buffers are persistent in the reference too
    assert set(sd) <= set(full) and all(".denoise_fn." not in k and ".velocity_fn." not in k and
                                        not k.startswith("aux_decoder.") for k in set(full) - set(sd))
    full.update({k: torch.from_numpy(v) for k, v in sd.items()})
    model.load_state_dict(full, strict=True)
    model = model.cuda().eval()
    cond = dev(synth.synth_normal((bsz, t_len, hsz), 7500))
    noise = dev(synth.synth_normal((bsz, 1, m, t_len), nseed))
    with torch.no_grad():
        out = model(cond, dev(g["glue_mel2ph"]), infer=True, noise=noise)
    assert rel_err(out.aux_out, g[f"glue_{tag}_aux"]) < TOL_AUX
    assert rel_err(out.diff_out, g[f"glue_{tag}_mel"]) < TOL_SAMPLER
    assert torch.all(out.diff_out[0, 41:] == 0) and torch.all(out.diff_out[1, 48:] == 0)
    with pytest.raises(NotImplementedError):
        model(cond, dev(g["glue_mel2ph"]), infer=False)
    set_hp()


def test_empty_inputs_return_empty_outputs():
    """B == 0 or T == 0: like the reference's torch modules, an empty input gives an empty output (no launch)."""
    from diffsinger_amd.diffusion import GaussianDiffusion, RectifiedFlow
    from gpu_util import make_backbone
    set_hp(diff_accelerator="ddim", diff_speedup=10, K_step_infer=1000)
    args = dict(num_layers=2, num_channels=64, dilation_cycle_length=2)
    d = GaussianDiffusion(32, 1, backbone_type="wavenet", backbone_args=args, spec_min=[-12.0], spec_max=[0.0]).cuda().eval()
    r = RectifiedFlow(32, 1, backbone_type="wavenet", backbone_args=args, spec_min=[-12.0], spec_max=[0.0]).cuda().eval()
    for shape in ((0, 17, 256), (2, 0, 256)):
        cond = torch.zeros(shape, device="cuda")
        with torch.no_grad():
            assert tuple(d(cond, infer=True).shape) == (shape[0], shape[1], 32)
            assert tuple(r(cond, infer=True).shape) == (shape[0], shape[1], 32)
    net, _ = make_backbone("wavenet", 32, 1, args, 3)
    with torch.no_grad():
        y = net(torch.zeros(0, 1, 32, 9).cuda(), torch.zeros(0).cuda(), torch.zeros(0, 256, 9).cuda())
    assert tuple(y.shape) == (0, 1, 32, 9)
    a, _ = make_adaptor(256, 32, 64, 2, 7, 71, [-12.0], [0.0])
    with torch.no_grad():
        assert tuple(a(torch.zeros(0, 5, 256).cuda(), infer=True).shape) == (0, 5, 32)
        assert tuple(a(torch.zeros(3, 0, 256).cuda(), infer=True).shape) == (3, 0, 32)
