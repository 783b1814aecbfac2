"""-m gpu parity of the FastSpeech2 acoustic encoder (dsd_encode) against the
fixtures generated from the reference (G8) and the numpy oracle.  Stated fp32 tolerance: 2e-5 of the output range
(oracle-vs-reference is <= 2e-6)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from diffsinger_amd import synth  # noqa: E402
from gpu_util import dev, rel_err, set_hp  # noqa: E402
from oracle import encoder as oe  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 2e-5
ENC_HP = dict(hidden_size=256, enc_layers=4, enc_ffn_kernel_size=3, ffn_act="gelu", dropout=0.1, num_heads=2,
              use_pos_embed=True, rel_pos=True, use_rope=True, use_spk_id=False, num_spk=1, use_lang_id=False, num_lang=1)
CASES = {
    "default": (dict(), dict()),
    "padded": (dict(), dict()),
    "full": (dict(use_spk_id=True, num_spk=3, use_lang_id=True, num_lang=2, use_energy_embed=True,
                  use_breathiness_embed=True, use_key_shift_embed=True, use_speed_embed=True),
             dict(num_spk=3, num_lang=2, variances=("energy", "breathiness"), key_shift=True, speed=True)),
    "k9": (dict(enc_ffn_kernel_size=9, enc_layers=2, hidden_size=128), dict()),
    # pre-rotary checkpoints (use_rope false): RelPositionalEncoding / no positions, torch.nn.MultiheadAttention names
    "relpos": (dict(use_rope=False, rel_pos=True, enc_layers=2), dict(rope=False)),
    "nopos": (dict(use_rope=False, use_pos_embed=False, enc_layers=2), dict(rope=False)),
    "sinpos": (dict(use_rope=False, rel_pos=False, enc_layers=2), dict(rope=False, sinpos=True)),
    # TransformerFFNLayer's other activations (common_layers.py:126-136)
    "relu": (dict(ffn_act="relu", enc_layers=2), dict()),
    "swish": (dict(ffn_act="swish", enc_layers=2), dict()),
    "swiglu": (dict(ffn_act="swiglu", enc_layers=2, enc_ffn_kernel_size=5), dict(ffn_act="swiglu")),
}


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    yield
    set_hp()


def build(vocab, hp, skw, wseed):
    from diffsinger_amd.encoder import FastSpeech2Acoustic
    full = dict(ENC_HP)
    full.update(hp)
    set_hp(**full)
    m = FastSpeech2Acoustic(vocab)
    kw = dict(hidden_size=full["hidden_size"], enc_layers=full["enc_layers"], num_heads=full["num_heads"],
              ffn_kernel_size=full["enc_ffn_kernel_size"])
    kw.update(skw)
    params = synth.synth_state_dict(synth.fs2_acoustic_param_shapes(vocab, **kw), seed=wseed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    return m.cuda().eval(), params, full


@pytest.mark.parametrize("tag", sorted(CASES))
def test_encoder_vs_golden(tag):
    g = np.load(os.path.join(GOLDEN, "g8_encoder.npz"))
    vocab, hidden, layers, heads, ks, bsz, t_txt, t_mel, wseed = (int(v) for v in g[f"{tag}_meta"])
    hp, skw = CASES[tag]
    m, _, _ = build(vocab, hp, skw, wseed)
    kwargs = {k: dev(g[f"{tag}_{k}"]) for k in ("key_shift", "speed", "energy", "breathiness", "languages", "spk_embed_id")
              if f"{tag}_{k}" in g}
    with torch.no_grad():
        cond = m(dev(g[f"{tag}_tokens"]), dev(g[f"{tag}_mel2ph"]), dev(g[f"{tag}_f0"]), **kwargs)
    want = g[f"{tag}_cond"]
    if tag in ("default", "padded", "relpos", "nopos", "sinpos"):
        cond = cond[:, ::2]
    assert rel_err(cond, want) < TOL
    m.release_native()


@pytest.mark.parametrize("bsz,t_txt,t_mel", [(1, 1, 3), (1, 2, 2), (2, 65, 400), (1, 300, 2000)])
def test_encoder_vs_oracle_sizes(bsz, t_txt, t_mel):
    """One token, tokens == frames, a 65-token padded batch (two key chunks), a long 300-token segment."""
    m, params, _ = build(50, dict(), dict(), 91)
    rng = np.random.Generator(np.random.PCG64(t_txt * 7 + bsz))
    tokens = np.zeros((bsz, t_txt), np.int64)
    mel2ph = np.zeros((bsz, t_mel), np.int64)
    for b in range(bsz):
        n_tok = t_txt if b == 0 else max(1, t_txt // 2)
        tokens[b, :n_tok] = rng.integers(1, 50, n_tok)
        n_fr = t_mel if b == 0 else max(n_tok, t_mel // 2)
        durs = np.ones(n_tok, np.int64)
        durs[rng.integers(0, n_tok, n_fr - n_tok)] += 0          # keep ones; distribute the rest below
        extra = rng.multinomial(n_fr - n_tok, np.ones(n_tok) / n_tok)
        mel2ph[b, :n_fr] = np.repeat(np.arange(1, n_tok + 1), durs + extra)
    f0 = (200.0 * 2.0 ** rng.uniform(-1, 1, (bsz, t_mel))).astype(np.float32)
    want = oe.fs2_acoustic_forward(params, tokens, mel2ph, f0, num_heads=2)
    with torch.no_grad():
        got = m(dev(tokens), dev(mel2ph), dev(f0))
        got32 = m(dev(tokens).int(), dev(mel2ph).int(), dev(f0))          # int32 indices are widened
    assert rel_err(got, want) < TOL
    assert torch.equal(got, got32)
    m.release_native()


def test_encoder_spk_mix_and_errors():
    hp, skw = CASES["full"]
    m, params, _ = build(45, hp, skw, 82)
    g = np.load(os.path.join(GOLDEN, "g8_encoder.npz"))
    tag = "full"
    base = {k: dev(g[f"{tag}_{k}"]) for k in ("key_shift", "speed", "energy", "breathiness", "languages")}
    tokens, mel2ph, f0 = dev(g["full_tokens"]), dev(g["full_mel2ph"]), dev(g["full_f0"])
    mix = synth.synth_normal((2, 1, 256), 5)
    want = oe.fs2_acoustic_forward(params, g["full_tokens"], g["full_mel2ph"], g["full_f0"], num_heads=2,
                                   spk_mix_embed=mix, **{k: g[f"full_{k}"] for k in base})
    with torch.no_grad():
        got = m(tokens, mel2ph, f0, spk_mix_embed=dev(mix), **base)
    assert rel_err(got, want) < TOL
    with torch.no_grad():
        with pytest.raises(ValueError, match="spk_embed_id"):
            m(tokens, mel2ph, f0, **base)
        with pytest.raises(KeyError):
            m(tokens, mel2ph, f0, spk_embed_id=dev(g["full_spk_embed_id"]), languages=base["languages"],
              key_shift=base["key_shift"], speed=base["speed"])
        with pytest.raises(RuntimeError, match="no CPU path"):
            m(tokens.cpu(), mel2ph.cpu(), f0.cpu(), spk_embed_id=g["full_spk_embed_id"], **base)
    with pytest.raises(RuntimeError, match="inference-only"):
        m(tokens, mel2ph, f0, spk_embed_id=dev(g["full_spk_embed_id"]), **base)
    m.release_native()
    from diffsinger_amd.encoder import FastSpeech2Acoustic
    set_hp(**dict(ENC_HP, ffn_act="tanh"))
    with pytest.raises(ValueError, match="not a valid activation"):      # the reference's error (common_layers.py:135-136)
        FastSpeech2Acoustic(10)


@pytest.mark.parametrize("tag", ("ddpm_dpm", "reflow_euler"))
def test_diffsinger_acoustic_tokens_to_mel_vs_golden(tag):
    """diffsinger_amd.toplevel.DiffSingerAcoustic against the reference's own top-level model (G9): phoneme tokens,
    durations and f0 in, mel out - encoder, aux decoder, masks and the shallow loop all on the HIP library."""
    from diffsinger_amd.toplevel import DiffSingerAcoustic
    g = np.load(os.path.join(GOLDEN, "g9_acoustic_model.npz"))
    vocab, m, bsz, t_txt, t_mel, nseed = (int(v) for v in g["meta"])
    sn_args = dict(num_layers=4, num_channels=64, dilation_cycle_length=2)
    hp = dict(ENC_HP, enc_layers=2, use_shallow_diffusion=True, spec_min=g["smin"].tolist(), spec_max=g["smax"].tolist(),
              shallow_diffusion_args=dict(aux_decoder_arch="convnext", val_gt_start=False,
                                          aux_decoder_args=dict(num_channels=64, num_layers=2, kernel_size=7)),
              backbone_type="wavenet", backbone_args=sn_args, timesteps=1000, K_step=400, T_start=0.4,
              time_scale_factor=1000)
    if tag == "ddpm_dpm":
        set_hp(diffusion_type="ddpm", diff_accelerator="dpm-solver", diff_speedup=20, K_step_infer=400, **hp)
    else:
        set_hp(diffusion_type="reflow", sampling_algorithm="euler", sampling_steps=20, T_start_infer=0.4, **hp)
    model = DiffSingerAcoustic(vocab, m)
    sd = dict(model.state_dict())
    sd.update({"fs2." + k: torch.from_numpy(v) for k, v in
               synth.synth_state_dict(synth.fs2_acoustic_param_shapes(vocab, enc_layers=2), seed=9200).items()})
    sd.update({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(
        synth.convnext_param_shapes(256, m, num_channels=64, num_layers=2, prefix="aux_decoder.decoder."), seed=9201).items()})
    fn = "denoise_fn" if tag == "ddpm_dpm" else "velocity_fn"
    sd.update({f"diffusion.{fn}.{k}": torch.from_numpy(v) for k, v in synth.synth_state_dict(
        synth.backbone_param_shapes("wavenet", m, 1, hidden_size=256, **sn_args), seed=9202).items()})
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    noise = dev(synth.synth_normal((bsz, 1, m, t_mel), nseed))
    with torch.no_grad():
        out = model(dev(g["tokens"]), dev(g["mel2ph"]), dev(g["f0"]), infer=True, noise=noise)
    assert rel_err(out.aux_out, g[f"{tag}_aux"]) < 2e-5
    assert rel_err(out.diff_out, g[f"{tag}_mel"]) < 1.5e-5       # measured 1.1e-6 (profiles/r02_parity.json)
    set_hp()
