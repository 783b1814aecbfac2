"""-m gpu: a seeded sweep of backbone configurations against the numpy oracle, chosen to walk every kernel
selection the host makes (narrow 16-frame / 32-frame / 64-frame tiles, pipelined / resident / generic staging,
dilations up to 32, channel counts that are not multiples of 64, hidden sizes other than 256, n_feats > 1).
Tolerance: 2e-5 of the output range per evaluation, as in test_gpu_parity.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from diffsinger_amd import synth  # noqa: E402
from gpu_util import dev, rel_err, set_hp  # noqa: E402
from oracle import backbones as ob  # noqa: E402

TOL_NFE = 2e-5


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    yield
    set_hp()


def _build(kind, in_dims, n_feats, args, hidden, seed):
    from diffsinger_amd.backbones import build_backbone
    set_hp(hidden_size=hidden)
    shapes = synth.backbone_param_shapes(kind, in_dims, n_feats, hidden_size=hidden, **args)
    params = synth.synth_state_dict(shapes, seed=seed)
    net = build_backbone(in_dims, n_feats, kind, args)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    return net.cuda().eval(), params


WAVENET_SWEEP = [
    # (in_dims, n_feats, C, L, cycle, hidden, B, T)
    (128, 1, 256, 6, 6, 256, 1, 700),      # dilation up to 32 (generic path for the widest halos), narrow tiles
    (128, 1, 256, 5, 5, 256, 1, 1500),     # dilation 16: the S=80 pipelined conv, 32-frame tiles
    (128, 1, 256, 5, 5, 256, 12, 900),     # dilation 16 at 64-frame tiles (S=112)
    (64, 1, 256, 4, 4, 256, 5, 1000),      # pitch-like bins, 64-frame tiles
    (24, 2, 192, 4, 4, 256, 2, 333),       # C = 192 (K = 192: three chunks), F*M = 48 (generic in-proj)
    (20, 3, 96, 3, 2, 128, 3, 77),         # C = 96 (not a multiple of 64: generic everywhere), hidden 128
    (80, 1, 128, 3, 3, 192, 1, 2100),      # C = 128 (two chunks, resident), hidden 192, longer T
    (128, 1, 512, 2, 2, 256, 1, 260),      # C = 512 (K > 256: pipelined 1x1 GEMMs, no resident variant)
    (8, 1, 32, 2, 1, 64, 4, 19),           # tiny everything
]


@pytest.mark.parametrize("cfg", WAVENET_SWEEP, ids=[f"wn{i}" for i in range(len(WAVENET_SWEEP))])
def test_wavenet_config_sweep_vs_oracle(cfg):
    in_dims, n_feats, c, nl, cyc, hidden, bsz, t_len = cfg
    args = dict(num_layers=nl, num_channels=c, dilation_cycle_length=cyc)
    net, params = _build("wavenet", in_dims, n_feats, args, hidden, seed=100 + c + nl)
    x = synth.synth_normal((bsz, n_feats, in_dims, t_len), 31)
    cond = synth.synth_normal((bsz, hidden, t_len), 32)
    t = (np.arange(bsz) * 97.5 + 3.0).astype(np.float32)
    want = ob.wavenet_forward(params, x, t, cond, dilation_cycle_length=cyc)
    with torch.no_grad():
        out = net(dev(x), dev(t), dev(cond))
    assert rel_err(out, want) < TOL_NFE, cfg
    net.release_native()


LYNX_SWEEP = [
    # (in_dims, n_feats, C, L, expansion, k, activation, strong, hidden, B, T)
    (128, 1, 512, 2, 2, 31, "PReLU", False, 256, 1, 900),    # class default width, small grid (16-frame LN tiles)
    (128, 1, 1024, 1, 2, 31, "PReLU", True, 256, 6, 1000),   # acoustic.yaml width, 64-frame tiles
    (64, 1, 256, 2, 1, 15, "SiLU", True, 256, 2, 130),       # expansion 1, k = 15 (generic depthwise path)
    (24, 2, 192, 2, 2, 7, "ReLU", False, 128, 3, 65),        # C = 192, k = 7, hidden 128
    (16, 1, 64, 1, 4, 63, "PReLU", False, 64, 1, 40),        # widest supported depthwise kernel
]


@pytest.mark.parametrize("cfg", LYNX_SWEEP, ids=[f"lx{i}" for i in range(len(LYNX_SWEEP))])
def test_lynxnet_config_sweep_vs_oracle(cfg):
    in_dims, n_feats, c, nl, exp, ks, act, strong, hidden, bsz, t_len = cfg
    args = dict(num_layers=nl, num_channels=c, expansion_factor=exp, kernel_size=ks, activation=act, strong_cond=strong)
    net, params = _build("lynxnet", in_dims, n_feats, args, hidden, seed=200 + c + ks)
    x = synth.synth_normal((bsz, n_feats, in_dims, t_len), 41)
    cond = synth.synth_normal((bsz, hidden, t_len), 42)
    t = np.array([250.25], np.float32)
    want = ob.lynxnet_forward(params, x, t, cond, activation=act, strong_cond=strong)
    with torch.no_grad():
        out = net(dev(x), dev(t), dev(cond))
    assert rel_err(out, want) < TOL_NFE, cfg
    net.release_native()
