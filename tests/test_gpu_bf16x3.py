"""The opt-in split-bf16 precision mode of the fused WaveNet layer (wn_layer_x3.hip, dsd_set_precision / DSD_PRECISION=1): every
operand of the layer's two GEMMs split x = hi + lo into two bf16 values, hi.hi + hi.lo + lo.hi on v_mfma_f32_16x16x32_bf16,
fp32 accumulation.  Stated tolerance = the fp32 path's own: 2e-5 on one evaluation, 1.5e-5 on a sampler run, both max and
RMS relative (gpu_util.check) - tools/bf16x3_tolerance.py measures 9.9e-6 / 1.9e-6 for the arithmetic itself on the CPU
oracle; what the GPU adds is recorded in profiles/r03_parity.json.  Every BASELINE number is measured in fp32; this mode is a
separate workload of bench.py (--precision bf16x3)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from diffsinger_amd import synth  # noqa: E402
from gpu_util import check, dev, load_synth, make_backbone, set_hp, synth_params  # noqa: E402
from oracle import backbones as ob  # noqa: E402
from oracle import diffusion as od  # noqa: E402

TOL_NFE = 2e-5
TOL_SAMPLER = 1.5e-5
SWITCHES = ("DSD_FUSED_LAYER", "DSD_WN_PLAN", "DSD_PRECISION", "DSD_X3_WIDE", "DSD_LYNX_RESIDENT")


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


NETS = {
    "c256_cyc4": (128, dict(num_layers=5, num_channels=256, dilation_cycle_length=4)),       # halo 8
    "c256_cyc5": (64, dict(num_layers=6, num_channels=256, dilation_cycle_length=5)),        # halo 16 in layer 4
}
GRIDS = {
    "dense_T211_B2": (2, 211, None),
    "dense_T13_B3": (3, 13, None),
    "ragged_B3": (3, 200, [200, 77, 141]),
}


def _eval(net, x, t, cond, lengths):
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
def test_bf16x3_fused_layer_vs_oracle(net_name, grid):
    in_dims, args = NETS[net_name]
    bsz, t_len, lengths = GRIDS[grid]
    os.environ["DSD_FUSED_LAYER"] = "1"           # the fused kernel at a size the oracle handles
    net, params = make_backbone("wavenet", in_dims, 1, args, 42)
    x = synth.synth_normal((bsz, 1, in_dims, t_len), 21)
    cond = synth.synth_normal((bsz, 256, t_len), 22)
    t = (np.arange(bsz) * 173.25 + 7.5).astype(np.float32)
    f32 = _eval(net, x, t, cond, lengths)
    assert net.stats()["precision"] == 0
    net.set_precision("bf16x3")
    x3 = _eval(net, x, t, cond, lengths)
    st = net.stats()
    tiles = sum((n + 31) // 32 for n in lengths) if lengths else bsz * ((t_len + 31) // 32)
    assert st["precision"] == 1 and st["fused_tiles"] == tiles and st["layer_launches"] == 1, st
    assert not np.array_equal(x3, f32)            # another arithmetic really ran
    cyc = args["dilation_cycle_length"]
    if lengths is None:
        want = ob.wavenet_forward(params, x, t, cond, dilation_cycle_length=cyc)
        check(x3, want, TOL_NFE, what=("bf16x3 vs oracle", net_name, grid))
        check(x3, f32, TOL_NFE, what=("bf16x3 vs fp32 kernels", net_name, grid))
    else:
        for b, n in enumerate(lengths):
            want = ob.wavenet_forward(params, x[b:b + 1, :, :, :n], t[b:b + 1], cond[b:b + 1, :, :n], dilation_cycle_length=cyc)
            check(x3[b:b + 1, :, :, :n], want, TOL_NFE, what=("bf16x3 vs oracle", net_name, grid, b))
    net.set_precision("f32")                      # and back: the fp32 weights are the same as before
    back = _eval(net, x, t, cond, lengths)
    assert net.stats()["precision"] == 0
    assert np.array_equal(back, f32)
    net.release_native()


def test_bf16x3_config4_share_vs_oracle():
    """BASELINE config 4's per-GPU share (8 x 1000 frames, 20 x 256 WaveNet) in split-bf16, DPM-Solver++ shortened to 10 steps so
    that the oracle finishes in seconds: the fused bf16x3 kernel on its natural grid (256 tiles), hipGraph path."""
    from diffsinger_amd.diffusion import GaussianDiffusion
    set_hp(diff_accelerator="dpm-solver", diff_speedup=100, K_step_infer=1000)
    args = dict(num_layers=20, num_channels=256, dilation_cycle_length=4)
    d = GaussianDiffusion(128, 1, timesteps=1000, k_step=1000, backbone_type="wavenet", backbone_args=args,
                          spec_min=[-12.0], spec_max=[0.0])
    params = synth_params("wavenet", 128, 1, args, 42)
    load_synth(d.denoise_fn, params)
    d = d.cuda().eval()
    d.denoise_fn.set_precision("bf16x3")
    bsz, t_len = 8, 1000
    cond = synth.synth_normal((bsz, t_len, 256), 40)
    noise = synth.synth_normal((bsz, 1, 128, t_len), 41)
    out = d(dev(cond), infer=True, noise=dev(noise))
    st = d.denoise_fn.stats()
    assert st["precision"] == 1 and st["fused_tiles"] == 256 and st["layer_launches"] == 1, st
    fn = lambda x, t, c: ob.wavenet_forward(params, x, t, c, dilation_cycle_length=4)   # noqa: E731
    o = od.GaussianDiffusion(fn, 128, 1, spec_min=[-12.0], spec_max=[0.0])
    want = o.forward(cond, noise, diff_accelerator="dpm-solver", diff_speedup=100, K_step_infer=1000)
    check(out, want, TOL_SAMPLER, what="bf16x3, 8 x 1000 frames, 10 NFE")
    d.denoise_fn.release_native()


def test_bf16x3_mixed_plan_vs_oracle():
    """A mixed plan in split-bf16 mode: the fused half of the tiles on wn_layer_x3.hip, the rest on the (fp32) two-launch form."""
    in_dims, args = NETS["c256_cyc5"]
    os.environ["DSD_WN_PLAN"] = "2"
    net, params = make_backbone("wavenet", in_dims, 1, args, 46)
    net.set_precision("bf16x3")
    bsz, t_len, lengths = 3, 200, [200, 77, 141]
    x = synth.synth_normal((bsz, 1, in_dims, t_len), 61)
    cond = synth.synth_normal((bsz, 256, t_len), 62)
    t = (np.arange(bsz) * 173.25 + 7.5).astype(np.float32)
    out = _eval(net, x, t, cond, lengths)
    st = net.stats()
    tiles = sum((n + 31) // 32 for n in lengths)
    assert st["precision"] == 1 and st["layer_launches"] == 3 and st["fused_tiles"] == tiles // 2, st
    for b, n in enumerate(lengths):
        want = ob.wavenet_forward(params, x[b:b + 1, :, :, :n], t[b:b + 1], cond[b:b + 1, :, :n], dilation_cycle_length=5)
        check(out[b:b + 1, :, :, :n], want, TOL_NFE, what=("bf16x3 mixed plan", b))
    net.release_native()


# ---- LYNXNet's pointwise GEMMs in split-bf16 (lynx_x3.hip)
LX_NETS = {
    "c1024_strong": dict(num_layers=2, num_channels=1024, expansion_factor=2, kernel_size=31, activation="PReLU", strong_cond=True),
    "c512_default": dict(num_layers=3, num_channels=512, expansion_factor=2, kernel_size=31, activation="SiLU", strong_cond=False),
}


def _lx_case(net_name, bsz, t_len, lengths, expect_pw2_x3, force_resident, wide=None):
    args = LX_NETS[net_name]
    if force_resident:
        os.environ["DSD_LYNX_RESIDENT"] = "1"
    if wide is not None:
        os.environ["DSD_X3_WIDE"] = "1" if wide else "0"
    try:
        net, params = make_backbone("lynxnet", 128, 1, args, 59)
        x = synth.synth_normal((bsz, 1, 128, t_len), 21)
        cond = synth.synth_normal((bsz, 256, t_len), 22)
        t = (np.arange(bsz) * 173.25 + 7.5).astype(np.float32)
        f32 = _eval(net, x, t, cond, lengths)
        net.set_precision("bf16x3")
        out = _eval(net, x, t, cond, lengths)
        assert net.stats()["precision"] == 1 and not np.array_equal(out, f32)
        net.kernel_timing(True)
        _eval(net, x, t, cond, lengths)
        names = [k["name"] for k in net.kernel_classes()]
        net.kernel_timing(False)
        assert any(n.startswith("lx_x3_kernel<0") for n in names), names
        assert any(n.startswith("lx_x3_kernel<1") for n in names) == expect_pw2_x3, names
        if wide is not None:                                     # ", 4>" = 64-frame tiles, ", 2>" = 32-frame tiles
            assert all(n.endswith(", 4>" if wide else ", 2>") for n in names if n.startswith("lx_x3_kernel")), names
        fwd = lambda xx, tt, cc: ob.lynxnet_forward(params, xx, tt, cc, activation=args["activation"], strong_cond=args["strong_cond"])   # noqa: E731
        if lengths is None:
            check(out, fwd(x, t, cond), TOL_NFE, what=("lynx bf16x3", net_name, bsz, t_len))
        else:
            for b, n in enumerate(lengths):
                check(out[b:b + 1, :, :, :n], fwd(x[b:b + 1, :, :, :n], t[b:b + 1], cond[b:b + 1, :, :n]), TOL_NFE,
                      what=("lynx bf16x3", net_name, b))
        net.set_precision("f32")
        assert np.array_equal(_eval(net, x, t, cond, lengths), f32)
        net.release_native()
    finally:
        os.environ.pop("DSD_LYNX_RESIDENT", None)
        os.environ.pop("DSD_X3_WIDE", None)


@pytest.mark.parametrize("net_name", sorted(LX_NETS))
def test_lynx_bf16x3_small_grid_vs_oracle(net_name):
    """pw1 in split-bf16 (forced onto a grid the oracle handles; pw2 stays on the fp32 one-utterance kernels), cut tile, ragged"""
    _lx_case(net_name, 2, 211, None, False, True, wide=False)
    _lx_case(net_name, 3, 200, [200, 77, 141], False, True, wide=False)


@pytest.mark.parametrize("net_name", sorted(LX_NETS))
def test_lynx_bf16x3_wide_tiles_vs_oracle(net_name):
    """64-frame tiles (the weight stream serves twice the frames) forced at small sizes: tiles cut by the utterance end inside the
    first and the second 32-frame half, a ragged batch on the 64-frame tile list; pw2 too where the grid gives it (forced grids
    run pw1 only - the batched test below covers pw2)"""
    _lx_case(net_name, 2, 211, None, False, True, wide=True)
    _lx_case(net_name, 1, 90, None, False, True, wide=True)
    _lx_case(net_name, 3, 200, [200, 77, 141], False, True, wide=True)


def test_lynx_bf16x3_batched_grid_vs_oracle():
    """both pointwise GEMMs in split-bf16 on grids that give pw2 half a chip of workgroups: C = 1024 at 2 x 1000 frames (64 frame
    tiles x 2 row tiles), the class-default C = 512 at 5 x 1000 frames"""
    _lx_case("c1024_strong", 2, 1000, None, True, False, wide=False)
    _lx_case("c512_default", 5, 1000, None, True, False, wide=False)
    _lx_case("c1024_strong", 2, 1000, None, True, False, wide=True)          # both GEMMs on 64-frame tiles
    _lx_case("c512_default", 5, 1000, None, True, False, wide=True)
