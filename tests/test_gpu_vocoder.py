"""-m gpu parity of the NSF-HiFiGAN generator (dsd_vocode) against the fixtures
generated from the reference Generator (G10) and the numpy oracle.  Stated fp32 tolerance: 1e-4 of the waveform
range (oracle-vs-reference is <= 5e-5; the source's sin() of accumulated phases is the sensitive part)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from diffsinger_amd import synth  # noqa: E402
from gpu_util import dev, rel_err  # noqa: E402
from oracle import vocoder as ov  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-4
GAIN = 0.7
OVER = {
    "default": dict(),
    "small_rb2": dict(num_mels=32, upsample_rates=[4, 2, 2], upsample_kernel_sizes=[8, 4, 4], upsample_initial_channel=64,
                      resblock="2", resblock_kernel_sizes=[3, 5], resblock_dilation_sizes=[[1, 2], [2, 6]], hop_size=16),
}


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"


def build(over, wseed):
    from diffsinger_amd.vocoder import Generator
    h = dict(synth.NSF_HIFIGAN_DEFAULT)
    h.update(over)
    params = synth.synth_state_dict(synth.nsf_hifigan_param_shapes(h), seed=wseed, gain=GAIN)
    g = Generator(h)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    return g.cuda().eval(), h, params


@pytest.mark.parametrize("tag", sorted(OVER))
def test_vocoder_vs_golden(tag):
    from diffsinger_amd.vocoder import NsfHifiGAN
    g = np.load(os.path.join(GOLDEN, "g10_vocoder.npz"))
    bsz, t_len, wseed, upp = (int(v) for v in g[f"{tag}_meta"])
    gen, h, _ = build(OVER[tag], wseed)
    mel = (synth.synth_normal((bsz, t_len, h["num_mels"]), wseed + 1) * 1.5 - 5.0).astype(np.float32)
    noise = synth.synth_normal((bsz, t_len * upp, 9), wseed + 3)
    wav = NsfHifiGAN(gen).spec2wav_torch(dev(mel), f0=dev(g[f"{tag}_f0"]), rand_ini=dev(g[f"{tag}_rand_ini"]),
                                         noise=dev(noise))
    want = g[f"{tag}_wav"].reshape(-1)
    assert rel_err(wav, want) < TOL
    gen.release_native()


@pytest.mark.parametrize("tag,bsz,t_len", [("small_rb2", 1, 1), ("small_rb2", 3, 130), ("default", 1, 33)])
def test_vocoder_vs_oracle_sizes(tag, bsz, t_len):
    gen, h, params = build(OVER[tag], 410)
    upp = int(np.prod(h["upsample_rates"]))
    rng = np.random.Generator(np.random.PCG64(t_len))
    mel = (synth.synth_normal((bsz, h["num_mels"], t_len), 411) * 3.0 - 11.0).astype(np.float32)     # [B, M, T], ln-mel
    f0 = (150.0 * 2.0 ** rng.uniform(-1, 2, (bsz, t_len))).astype(np.float32)
    f0[:, ::7] = 0.0
    rand_ini = rng.random(9).astype(np.float32)
    noise = synth.synth_normal((bsz, t_len * upp, 9), 412)
    want = ov.generator_forward(params, h, mel, f0, rand_ini, noise)
    with torch.no_grad():
        got = gen(dev(mel), dev(f0), rand_ini=dev(rand_ini), noise=dev(noise))
        got_t = gen(dev(np.ascontiguousarray(mel.transpose(0, 2, 1))).transpose(1, 2), dev(f0), rand_ini=dev(rand_ini),
                    noise=dev(noise))                       # [B, T, M] storage viewed as [B, M, T]
        drawn = gen(dev(mel), dev(f0))                      # device-side draws: same shape, finite, bounded by tanh
    assert tuple(got.shape) == (bsz, 1, t_len * upp)
    assert rel_err(got, want) < TOL
    assert torch.equal(got, got_t)
    assert torch.isfinite(drawn).all() and drawn.abs().max() <= 1.0
    gen.release_native()


def test_vocoder_errors_and_weight_norm_fold():
    from diffsinger_amd.vocoder import Generator
    gen, h, params = build(OVER["small_rb2"], 420)
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="no CPU path"):
            gen(torch.zeros(1, 32, 4), torch.zeros(1, 4))
        with pytest.raises(ValueError):
            gen(torch.zeros(1, 31, 4).cuda(), torch.zeros(1, 4).cuda())
        assert tuple(gen(torch.zeros(0, 32, 4).cuda(), torch.zeros(0, 4).cuda()).shape) == (0, 1, 64)
    with pytest.raises(NotImplementedError):
        Generator(dict(h, mini_nsf=True))
    # a checkpoint that still carries weight norm: weight_g / weight_v pairs are folded on load
    sd = {}
    for k, v in params.items():
        t = torch.from_numpy(v)
        if k.endswith(".weight") and (k.startswith(("conv_pre", "ups.", "resblocks.", "conv_post"))):
            norm = t.flatten(1).norm(dim=1).reshape(-1, *([1] * (t.dim() - 1)))
            sd[k[:-6] + "weight_g"] = norm * 2.0
            sd[k[:-6] + "weight_v"] = t * 0.5 / 1.0
        else:
            sd[k] = t
    g2 = Generator(h)
    g2.load_state_dict(sd, strict=True)
    for k, v in params.items():
        if k.endswith(".weight") and k.startswith("ups."):
            assert torch.allclose(g2.state_dict()[k], torch.from_numpy(v) * 2.0, rtol=1e-5, atol=1e-7)
    gen.release_native()
