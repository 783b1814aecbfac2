"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol that
include/dsdenoise.h declares; struct layouts agree with the ctypes mirror.  No compute calls."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "dsdenoise.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dsd_[a-z_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from diffsinger_amd import _lib
    names = header_functions()
    assert len(names) >= 12
    assert sorted(_lib.EXPORTS) == names
    lib = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    assert _lib.lib().dsd_api_version() == 10


def test_struct_sizes_match_header():
    from diffsinger_amd import _lib
    assert C.sizeof(_lib.DsdConfig) == 13 * 4
    assert C.sizeof(_lib.DsdTerm) == 8
    assert C.sizeof(_lib.DsdLincomb) == 8 + 8 * _lib.DSD_MAX_TERMS
    assert C.sizeof(_lib.DsdEval) == 12 + _lib.DSD_MAX_OUT * C.sizeof(_lib.DsdLincomb)
    assert C.sizeof(_lib.DsdProgram) == 24


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from diffsinger_amd import _lib
    cfg = _lib.DsdConfig(C.sizeof(_lib.DsdConfig), 0, 128, 1, 20, 256, 256, 4, 0, 0, 0, 0, 0)
    h = C.c_void_p()
    rc = _lib.lib().dsd_create(C.byref(cfg), C.byref(h))
    assert rc < 0
    assert b"no HIP device" in _lib.lib().dsd_last_error(None)


def test_bad_config_rejected():
    from diffsinger_amd import _lib
    cfg = _lib.DsdConfig(C.sizeof(_lib.DsdConfig), 7, 128, 1, 20, 256, 256, 4, 0, 0, 0, 0, 0)
    h = C.c_void_p()
    assert _lib.lib().dsd_create(C.byref(cfg), C.byref(h)) == -1
    cfg = _lib.DsdConfig(C.sizeof(_lib.DsdConfig), 0, 128, 1, 20, 251, 256, 4, 0, 0, 0, 0, 0)     # WaveNet: any EVEN count
    assert _lib.lib().dsd_create(C.byref(cfg), C.byref(h)) == -1
    assert b"must be even" in _lib.lib().dsd_last_error(None)
    cfg = _lib.DsdConfig(C.sizeof(_lib.DsdConfig), 1, 128, 1, 6, 250, 256, 0, 2, 31, 0, 0, 0)      # LYNXNet: LayerNorm over C
    assert _lib.lib().dsd_create(C.byref(cfg), C.byref(h)) == -1
    assert b"multiple of 32" in _lib.lib().dsd_last_error(None)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "diffsinger_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            txt = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), fn


def test_cpu_tensor_raises_not_falls_back():
    import torch
    from diffsinger_amd.hparams import hparams
    hparams.update(hidden_size=256)
    from diffsinger_amd.backbones import build_backbone
    net = build_backbone(32, 1, "wavenet", dict(num_layers=2, num_channels=64, dilation_cycle_length=2, junk=1))
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU path"):
        net(torch.zeros(1, 1, 32, 8), torch.zeros(1), torch.zeros(1, 256, 8))
    with pytest.raises(RuntimeError, match="inference-only"):
        net(torch.zeros(1, 1, 32, 8), torch.zeros(1), torch.zeros(1, 256, 8))


def test_state_dict_names_match_reference_layout():
    import torch  # noqa: F401
    from diffsinger_amd import synth
    from diffsinger_amd.hparams import hparams
    hparams.update(hidden_size=256)
    from diffsinger_amd.backbones import build_backbone
    for kind, args in (("wavenet", dict(num_layers=3, num_channels=64, dilation_cycle_length=2)),
                       ("lynxnet", dict(num_layers=2, num_channels=64, expansion_factor=2, kernel_size=31,
                                        activation="PReLU", dropout_rate=0.1)),
                       ("lynxnet", dict(num_layers=2, num_channels=64, activation="SiLU"))):
        net = build_backbone(32, 1, kind, args)
        shapes = synth.backbone_param_shapes(kind, 32, 1, hidden_size=256, **args)
        sd = net.state_dict()
        assert list(sd) == list(shapes) or set(sd) == set(shapes)
        for k, v in shapes.items():
            assert tuple(sd[k].shape) == tuple(v), k


def test_aux_decoder_state_dict_and_host_checks():
    """ConvNeXt aux decoder shim: reference state_dict layout (convnext.py:24-35,63-76), registry behaviour of
    modules/aux_decoder/__init__.py:7-21 and the loud CPU refusal."""
    import torch
    from diffsinger_amd import synth
    from diffsinger_amd.aux_decoder import AUX_DECODERS, AuxDecoderAdaptor, build_aux_decoder
    assert list(AUX_DECODERS) == ["convnext"]
    dec = build_aux_decoder(256, 128, "convnext", dict(num_channels=64, num_layers=2, kernel_size=7,
                                                        dropout_rate=0.1, not_an_argument=3))
    shapes = synth.convnext_param_shapes(256, 128, num_channels=64, num_layers=2, kernel_size=7)
    sd = dec.state_dict()
    assert set(sd) == set(shapes)
    for k, v in shapes.items():
        assert tuple(sd[k].shape) == tuple(v), k
    a = AuxDecoderAdaptor(256, 128, 1, [-12.0], [0.0], "convnext", dict(num_channels=64, num_layers=2))
    assert all(k.startswith("decoder.") for k in a.state_dict())          # spec_min/max are non-persistent
    assert tuple(a.spec_min.shape) == (1, 1, 1)
    x = torch.tensor([[[-12.0, -6.0, 0.0]]])
    assert torch.allclose(a.norm_spec(x), torch.tensor([[[-1.0, 0.0, 1.0]]]))
    assert torch.allclose(a.denorm_spec(a.norm_spec(x)), x)
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU path"):
        a(torch.zeros(1, 4, 256), infer=True)
    with pytest.raises(ValueError):
        build_aux_decoder(256, 128, "convnext", dict(kernel_size=6))


def test_encoder_state_dict_matches_reference_layout():
    """FastSpeech2Acoustic shim: reference state_dict names/shapes (acoustic_encoder.py:15-63, tts_modules.py:353-383,
    common_layers.py:154-234) incl. the per-layer alias of the shared rotary `freqs` parameter."""
    import torch
    from diffsinger_amd import synth
    from diffsinger_amd.hparams import hparams
    from diffsinger_amd.encoder import FastSpeech2Acoustic
    hparams.clear()
    hparams.update(hidden_size=128, enc_layers=2, enc_ffn_kernel_size=9, ffn_act="gelu", dropout=0.1, num_heads=2,
                   use_pos_embed=True, rel_pos=True, use_rope=True, use_spk_id=True, num_spk=4, use_lang_id=True,
                   num_lang=3, use_energy_embed=True, use_tension_embed=True, use_key_shift_embed=True,
                   use_speed_embed=True)
    m = FastSpeech2Acoustic(33)
    shapes = synth.fs2_acoustic_param_shapes(33, hidden_size=128, enc_layers=2, num_heads=2, ffn_kernel_size=9,
                                             num_spk=4, num_lang=3, variances=("energy", "tension"), key_shift=True,
                                             speed=True)
    sd = m.state_dict()
    assert set(sd) == set(shapes)
    for k, v in shapes.items():
        assert tuple(sd[k].shape) == tuple(v), k
    f = sd["encoder.layers.0.op.self_attn.rotary_embed.freqs"]
    assert torch.allclose(f, 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64)))
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.ones(1, 3, dtype=torch.long), torch.ones(1, 5, dtype=torch.long), torch.ones(1, 5),
          spk_embed_id=torch.zeros(1, dtype=torch.long), languages=torch.ones(1, 3, dtype=torch.long),
          energy=torch.zeros(1, 5), tension=torch.zeros(1, 5), key_shift=torch.zeros(1, 5), speed=torch.ones(1, 5))
    hparams.clear()
    hparams.update(hidden_size=256)


def test_header_is_plain_c99():
    """include/dsdenoise.h must be consumable by a C compiler (no C++ in the boundary)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    header = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "dsdenoise.h")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", header], check=True)


def test_graft_entry_build_runs():
    """The driver's build check: compiles (or finds up to date) every HIP source and imports the package."""
    import importlib
    entry = importlib.import_module("__graft_entry__")
    entry.build()
