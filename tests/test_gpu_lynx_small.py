"""LYNXNet's pw2 with 128 rows per workgroup (lynx_layer.hip, lx_pw2q_kernel: one-utterance grids - the fork's default acoustic
configuration runs there) against the numpy oracle: forced on (DSD_LYNX_RESIDENT=1 + DSD_LYNX_PW2Q=1) at sizes the oracle
handles, for the fork's C = 1024 / inner 2048 network and the class default C = 512 / inner 1024, strong_cond on and off, dense,
cut tiles and ragged; and on its natural grid (B = 1, T = 1000).  The kernel classes of the pass are read back
(dsd_kernel_timing_classes): the test fails if another kernel served pw2."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from diffsinger_amd import synth  # noqa: E402
from gpu_util import check, dev, make_backbone, set_hp  # noqa: E402
from oracle import backbones as ob  # noqa: E402

TOL_NFE = 2e-5
SWITCHES = ("DSD_LYNX_RESIDENT", "DSD_LYNX_PW2Q", "DSD_LYNX_PW1P")
NETS = {
    "c1024_strong": dict(num_layers=3, num_channels=1024, expansion_factor=2, kernel_size=31, activation="PReLU", strong_cond=True),
    "c512_default": dict(num_layers=3, num_channels=512, expansion_factor=2, kernel_size=31, activation="PReLU", strong_cond=False),
}
GRIDS = {"dense_T211_B2": (2, 211, None), "dense_T96_B1": (1, 96, None), "ragged_B3": (3, 200, [200, 77, 141])}


@pytest.fixture(autouse=True)
def _clean_env():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    set_hp()
    saved = {k: os.environ.pop(k, None) for k in SWITCHES}
    yield
    for k in SWITCHES:
        os.environ.pop(k, None)
        if saved[k] is not None:
            os.environ[k] = saved[k]


def _run(net, x, t, cond, lengths):
    xd = dev(x)
    net.set_lengths(lengths, xd.device)
    with torch.no_grad():
        out = net(xd, dev(t), dev(cond))
        again = net(xd, dev(t), dev(cond))
    torch.cuda.synchronize()
    assert torch.equal(out, again)
    return out.cpu().numpy()


@pytest.mark.parametrize("grid", sorted(GRIDS))
@pytest.mark.parametrize("net_name", sorted(NETS))
def test_pw2_128_rows_forced_vs_oracle(net_name, grid):
    args = NETS[net_name]
    bsz, t_len, lengths = GRIDS[grid]
    os.environ["DSD_LYNX_RESIDENT"] = "1"
    os.environ["DSD_LYNX_PW2Q"] = "1"
    net, params = make_backbone("lynxnet", 128, 1, args, 57)
    x = synth.synth_normal((bsz, 1, 128, t_len), 21)
    cond = synth.synth_normal((bsz, 256, t_len), 22)
    t = (np.arange(bsz) * 173.25 + 7.5).astype(np.float32)
    out = _run(net, x, t, cond, lengths)
    net.kernel_timing(True)
    _run(net, x, t, cond, lengths)
    names = [k["name"] for k in net.kernel_classes()]
    net.kernel_timing(False)
    assert any(n.startswith("lx_pw2q_kernel") for n in names) and not any("gemm_kernel<0, 1, 7" in n or "lx_pw2d" in n for n in names), names
    fwd = lambda xx, tt, cc: ob.lynxnet_forward(params, xx, tt, cc, activation=args["activation"], strong_cond=args["strong_cond"])   # noqa: E731
    if lengths is None:
        check(out, fwd(x, t, cond), TOL_NFE, what=("pw2 128 rows", net_name, grid))
    else:
        for b, n in enumerate(lengths):
            check(out[b:b + 1, :, :, :n], fwd(x[b:b + 1, :, :, :n], t[b:b + 1], cond[b:b + 1, :, :n]), TOL_NFE,
                  what=("pw2 128 rows", net_name, grid, b))
    net.release_native()


def test_pw2_128_rows_natural_grid_vs_oracle():
    """One utterance of 1000 frames of the fork's 6 x 1024 network (here 2 layers): the library's own choice is pw1 resident + the
    128-row pw2."""
    args = dict(NETS["c1024_strong"], num_layers=2)
    net, params = make_backbone("lynxnet", 128, 1, args, 58)
    x = synth.synth_normal((1, 1, 128, 1000), 31)
    cond = synth.synth_normal((1, 256, 1000), 32)
    t = np.array([412.0], np.float32)
    out = _run(net, x, t, cond, None)
    net.kernel_timing(True)
    _run(net, x, t, cond, None)
    names = [k["name"] for k in net.kernel_classes()]
    net.kernel_timing(False)
    assert any(n.startswith("lx_pw2q_kernel<512") for n in names) and any(n.startswith("lx_pw1") for n in names), names
    check(out, ob.lynxnet_forward(params, x, t, cond, activation="PReLU", strong_cond=True), TOL_NFE, what="pw2 128 rows, B = 1, T = 1000")
    net.release_native()
