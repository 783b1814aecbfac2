"""Every instantiation of the fused WaveNet layer kernel (wn_layer.hip: <C / 64 in {3, 4}, LDS row stride 48 | 80 = halo for
dilation <= 8 | 16, dense | ragged>) and the mixed layer plans (api.hip, wn_plan_for: whole rounds of tiles on the fused kernel,
the remainder on the row-split pair) under oracle parity IN the driver's -m gpu suite.

The library picks a layer's launch shape from the grid (one 32-frame tile per CU and more: fused), so at the small sizes the
numpy oracle finishes in seconds the fused kernel would never run: the path switches are read on every C-ABI call
(dsd_internal.h, PathOpts), and these tests set DSD_FUSED_LAYER=1 (every tile through the fused kernel) or DSD_WN_PLAN=2 (half the
tiles fused, half row-split) around their calls.  dsd_get_stats reports which plan ran (layer_launches, fused_tiles,
split_tiles); every test asserts it, so a test that silently fell back to another path fails.

One evaluation against oracle.backbones.wavenet_forward (modules/backbones/wavenet.py:18-48, 75-107): tolerance 2e-5 (max and
RMS, gpu_util.check).  Config 5's per-GPU share at FULL size (B = 8, T = 1000, configs/variance.yaml:62-110: pitch 20 x 256,
cycle 5, 64 bins; variances 10 x 192, cycle 4, 2 x 24 bins; rectified flow, euler 20) runs on its natural path against the
oracle's 20-NFE loops: tolerance 1.5e-5."""
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
SWITCHES = ("DSD_FUSED_LAYER", "DSD_WN_PLAN", "DSD_ROWSPLIT", "DSD_EDGE", "DSD_FUSED16")


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


# (in_dims, n_feats, backbone args): the six (C, halo) shapes of the kernel
NETS = {
    "c256_cyc4": (128, 1, dict(num_layers=5, num_channels=256, dilation_cycle_length=4)),       # <4, 48>: dilation 1..8
    "c256_cyc5": (64, 1, dict(num_layers=6, num_channels=256, dilation_cycle_length=5)),        # <4, 80> in layer 4 (dilation 16)
    "c192_cyc4": (24, 2, dict(num_layers=5, num_channels=192, dilation_cycle_length=4)),        # <3, 48>
    "c192_cyc5": (24, 2, dict(num_layers=6, num_channels=192, dilation_cycle_length=5)),        # <3, 80>
    "c128_cyc4": (80, 1, dict(num_layers=5, num_channels=128, dilation_cycle_length=4)),        # <2, 48>
    "c128_cyc5": (80, 1, dict(num_layers=6, num_channels=128, dilation_cycle_length=5)),        # <2, 80>
}
# (B, T, lengths): T cut inside a tile, T < dilation (a tile that is all halo on one side), several tiles, ragged lists
GRIDS = {
    "dense_T211_B2": (2, 211, None),            # 7 tiles per item, the last cut at 19 frames
    "dense_T13_B3": (3, 13, None),              # T < dilation 16: every tap of a dilation-16 layer but the centre is padding
    "dense_T96_B1": (1, 96, None),              # whole tiles
    "ragged_B3": (3, 200, [200, 77, 141]),      # per-item ends inside tiles, different tile counts
    "ragged_short": (2, 64, [5, 64]),           # an item shorter than every dilation > 4
    # a mixed plan's segment boundary INSIDE an item: 15 tiles split 7 | 8 (dense), 13 tiles split 6 | 7 (ragged)
    "dense_T160_B3": (3, 160, None),
    "ragged_mid": (3, 160, [160, 96, 141]),
}


def _forward(net, x, t, cond, lengths):
    xd = dev(x)
    if lengths is not None:
        net.set_lengths(lengths, xd.device)
    with torch.no_grad():
        out = net(xd, dev(t), dev(cond))
        again = net(xd, dev(t), dev(cond))
    torch.cuda.synchronize()
    assert torch.equal(out, again)
    return out.cpu().numpy()


def _check_vs_oracle(out, params, x, t, cond, cycle, lengths, what):
    if lengths is None:
        check(out, ob.wavenet_forward(params, x, t, cond, dilation_cycle_length=cycle), TOL_NFE, what=what)
        return
    for b, n in enumerate(lengths):                 # frames past an item's end are the caller's to mask (toplevel.py:104)
        want = ob.wavenet_forward(params, x[b:b + 1, :, :, :n], t[b:b + 1], cond[b:b + 1, :, :n], dilation_cycle_length=cycle)
        check(out[b:b + 1, :, :, :n], want, TOL_NFE, what=what + (b,))


def _inputs(in_dims, n_feats, bsz, t_len, seed):
    x = synth.synth_normal((bsz, n_feats, in_dims, t_len), seed)
    cond = synth.synth_normal((bsz, 256, t_len), seed + 1)
    t = (np.arange(bsz) * 173.25 + 7.5).astype(np.float32)
    return x, t, cond


@pytest.mark.parametrize("grid", sorted(GRIDS))
@pytest.mark.parametrize("net_name", sorted(NETS))
def test_fused_layer_forced_vs_oracle(net_name, grid):
    in_dims, n_feats, args = NETS[net_name]
    bsz, t_len, lengths = GRIDS[grid]
    os.environ["DSD_FUSED_LAYER"] = "1"
    net, params = make_backbone("wavenet", in_dims, n_feats, args, 42)
    x, t, cond = _inputs(in_dims, n_feats, bsz, t_len, 21)
    out = _forward(net, x, t, cond, lengths)
    st = net.stats()
    tiles = sum((n + 31) // 32 for n in lengths) if lengths else bsz * ((t_len + 31) // 32)
    assert st["layer_launches"] == 1 and st["fused_tiles"] == tiles and st["split_tiles"] == 0, st
    assert st["kernels_per_nfe"] in (args["num_layers"] + 1, args["num_layers"] + 3), st
    _check_vs_oracle(out, params, x, t, cond, args["dilation_cycle_length"], lengths, ("fused forced", net_name, grid))
    # ... and the same evaluation on the path the library picks by itself at this size (two launches per layer) agrees
    os.environ.pop("DSD_FUSED_LAYER")
    out2 = _forward(net, x, t, cond, lengths)
    assert net.stats()["layer_launches"] == 2
    if lengths is None:
        check(out, out2, 8e-6, what=("fused vs two launches", net_name, grid))
    else:
        for b, n in enumerate(lengths):
            check(out[b:b + 1, :, :, :n], out2[b:b + 1, :, :, :n], 8e-6, what=("fused vs two launches", net_name, grid, b))
    net.release_native()


@pytest.mark.parametrize("grid", sorted(GRIDS))
@pytest.mark.parametrize("net_name", ["c256_cyc4", "c256_cyc5", "c192_cyc4", "c192_cyc5"])
def test_fused_layer_16_frame_tiles_forced_vs_oracle(net_name, grid):
    """wn_layer16_kernel (the fused layer on 16-frame tiles: the plan for 65 ... 128 32-frame tiles) forced on the small grids: tiles cut
    inside their 16 frames, T < dilation, ragged lists on the 16-frame tile list; against the oracle and against the 32-frame kernel"""
    in_dims, n_feats, args = NETS[net_name]
    bsz, t_len, lengths = GRIDS[grid]
    os.environ["DSD_FUSED16"] = "1"
    net, params = make_backbone("wavenet", in_dims, n_feats, args, 42)
    x, t, cond = _inputs(in_dims, n_feats, bsz, t_len, 23)
    out = _forward(net, x, t, cond, lengths)
    st = net.stats()
    tiles = sum((n + 15) // 16 for n in lengths) if lengths else bsz * ((t_len + 15) // 16)
    assert st["layer_launches"] == 1 and st["fused_tiles"] == tiles and st["split_tiles"] == 0, st
    net.kernel_timing(True)
    _forward(net, x, t, cond, lengths)
    names = [k["name"] for k in net.kernel_classes()]
    net.kernel_timing(False)
    assert any(n.startswith("wn_layer16_kernel<") for n in names) and not any(n.startswith("wn_layer_kernel<") for n in names), names
    _check_vs_oracle(out, params, x, t, cond, args["dilation_cycle_length"], lengths, ("fused 16-frame tiles forced", net_name, grid))
    os.environ.pop("DSD_FUSED16")
    os.environ["DSD_FUSED_LAYER"] = "1"
    out2 = _forward(net, x, t, cond, lengths)
    if lengths is None:
        check(out, out2, 8e-6, what=("16- vs 32-frame fused tiles", net_name, grid))
    else:
        for b, n in enumerate(lengths):
            check(out[b:b + 1, :, :, :n], out2[b:b + 1, :, :, :n], 8e-6, what=("16- vs 32-frame fused tiles", net_name, grid, b))
    net.release_native()


def test_fused_16_frame_tiles_natural_b4_vs_oracle():
    """B = 4, T = 1000 (128 32-frame tiles = 252 16-frame tiles for 256 CUs): the plan the library picks by itself is one launch
    per layer on 16-frame tiles; one evaluation against the oracle"""
    in_dims, n_feats, args = NETS["c256_cyc4"]
    net, params = make_backbone("wavenet", in_dims, n_feats, args, 42)
    x, t, cond = _inputs(in_dims, n_feats, 4, 1000, 29)
    out = _forward(net, x, t, cond, None)
    st = net.stats()
    assert st["layer_launches"] == 1 and st["fused_tiles"] == 4 * 63 and st["split_tiles"] == 0, st
    check(out, ob.wavenet_forward(params, x, t, cond, dilation_cycle_length=4), TOL_NFE, what="fused 16-frame tiles, natural plan at B = 4")
    net.release_native()


@pytest.mark.parametrize("hook", ["3", "4"])
@pytest.mark.parametrize("grid", ["dense_T211_B2", "ragged_B3", "dense_T160_B3", "ragged_mid"])
@pytest.mark.parametrize("net_name", ["c256_cyc4", "c256_cyc5"])
def test_mixed_plan_with_16_frame_segment_forced_vs_oracle(net_name, grid, hook):
    """Plans that put a 16-frame fused segment beside another launch shape, at any size (DSD_WN_PLAN=3: 16-frame fused tiles for the
    first half of the 32-frame tile order, the two-launch path for the rest; 4: the 32-frame fused kernel first, 16-frame tiles for the
    rest) - the segment boundary inside an item (dense_T160_B3, ragged_mid) and at item ends"""
    in_dims, n_feats, args = NETS[net_name]
    bsz, t_len, lengths = GRIDS[grid]
    os.environ["DSD_WN_PLAN"] = hook
    net, params = make_backbone("wavenet", in_dims, n_feats, args, 42)
    x, t, cond = _inputs(in_dims, n_feats, bsz, t_len, 27)
    out = _forward(net, x, t, cond, lengths)
    st = net.stats()
    assert st["layer_launches"] == (3 if hook == "3" else 2) and st["fused_tiles"] > 0, st
    net.kernel_timing(True)
    _forward(net, x, t, cond, lengths)
    names = [k["name"] for k in net.kernel_classes()]
    net.kernel_timing(False)
    assert any(n.startswith("wn_layer16_kernel<") for n in names), names
    _check_vs_oracle(out, params, x, t, cond, args["dilation_cycle_length"], lengths, ("mixed plan with a 16-frame segment", hook, net_name, grid))
    net.release_native()


def test_mixed_plan_16_frame_head_natural_b5_vs_oracle():
    """B = 5, T = 1000 (160 32-frame tiles): the library's own plan is one round of 16-frame fused tiles (256 of them = the first
    130 32-frame tiles) and the two-launch path for the other 30"""
    in_dims, n_feats, args = NETS["c256_cyc4"]
    net, params = make_backbone("wavenet", in_dims, n_feats, args, 42)
    x, t, cond = _inputs(in_dims, n_feats, 5, 1000, 31)
    out = _forward(net, x, t, cond, None)
    st = net.stats()
    assert st["layer_launches"] == 3 and st["fused_tiles"] == 256 and st["split_tiles"] == 30, st
    check(out, ob.wavenet_forward(params, x, t, cond, dilation_cycle_length=4), TOL_NFE, what="16-frame head + two-launch rest, natural plan at B = 5")
    net.release_native()


def test_mixed_plan_16_frame_head_natural_ragged_vs_oracle():
    """A ragged batch of 5 utterances (157 32-frame tiles, 311 16-frame tiles): the library's own plan puts one round of 16-frame
    fused tiles in front - its end falls inside item 4 - and the rest on the two-launch path; every item against the oracle run alone"""
    in_dims, n_feats, args = NETS["c256_cyc5"]
    lengths = [1000, 990, 1000, 1000, 1000 - 45]
    net, params = make_backbone("wavenet", in_dims, n_feats, args, 42)
    x, t, cond = _inputs(in_dims, n_feats, 5, 1000, 33)
    out = _forward(net, x, t, cond, lengths)
    st = net.stats()
    assert st["layer_launches"] == 3 and st["fused_tiles"] == 255 and st["split_tiles"] == 157 - 129, st      # 251 + 2 * 2 16-frame tiles = 129 32-frame tiles
    _check_vs_oracle(out, params, x, t, cond, args["dilation_cycle_length"], lengths, ("16-frame head, ragged natural plan",))
    net.release_native()


@pytest.mark.parametrize("grid", ["dense_T211_B2", "dense_T13_B3", "ragged_B3", "ragged_short", "dense_T160_B3", "ragged_mid"])
@pytest.mark.parametrize("net_name", ["c256_cyc4", "c256_cyc5"])
def test_mixed_plan_forced_vs_oracle(net_name, grid):
    """DSD_WN_PLAN=2: the first half of the tiles on the fused kernel, the rest on the row-split pair, both reading the layer's
    input buffer and writing the other one - the mechanism of the mixed plans at a size the oracle handles."""
    in_dims, n_feats, args = NETS[net_name]
    bsz, t_len, lengths = GRIDS[grid]
    os.environ["DSD_WN_PLAN"] = "2"
    net, params = make_backbone("wavenet", in_dims, n_feats, args, 43)
    x, t, cond = _inputs(in_dims, n_feats, bsz, t_len, 31)
    out = _forward(net, x, t, cond, lengths)
    st = net.stats()
    tiles = sum((n + 31) // 32 for n in lengths) if lengths else bsz * ((t_len + 31) // 32)
    assert st["layer_launches"] == 3 and st["fused_tiles"] == tiles // 2 and st["split_tiles"] == tiles - tiles // 2, st
    _check_vs_oracle(out, params, x, t, cond, args["dilation_cycle_length"], lengths, ("mixed forced", net_name, grid))
    net.release_native()


def test_mixed_plan_natural_b9_vs_oracle():
    """B = 9 utterances of 1000 frames = 288 tiles: the library's own plan is one round of the fused kernel (256 tiles) and the
    row-split pair over the last 32 - asserted, then one evaluation of a 3-layer net against the oracle, and item 8 (which
    the row-split pair computed) against the same item run alone."""
    args = dict(num_layers=3, num_channels=256, dilation_cycle_length=3)
    net, params = make_backbone("wavenet", 128, 1, args, 44)
    bsz, t_len = 9, 1000
    x, t, cond = _inputs(128, 1, bsz, t_len, 41)
    out = _forward(net, x, t, cond, None)
    st = net.stats()
    assert st["layer_launches"] == 3 and st["fused_tiles"] == 256 and st["split_tiles"] == 32, st
    check(out, ob.wavenet_forward(params, x, t, cond, dilation_cycle_length=3), TOL_NFE, what="B = 9 mixed plan")
    alone = _forward(net, x[8:9], t[8:9], cond[8:9], None)
    check(alone, out[8:9], 4e-6, what="item 8: row-split segment of the batch vs alone")
    net.release_native()


def test_ragged_equals_alone_through_fused_and_mixed():
    """An item of a ragged batch is computed exactly as if it ran alone - bit for bit - on the fused kernel and on a mixed plan
    (the same instantiation computes the same tile either way; the alone run is forced onto the same path)."""
    in_dims, n_feats, args = NETS["c256_cyc5"]
    lengths = [200, 77, 141]
    x, t, cond = _inputs(in_dims, n_feats, 3, 200, 51)
    for switch, val in (("DSD_FUSED_LAYER", "1"),):
        os.environ[switch] = val
        net, _ = make_backbone("wavenet", in_dims, n_feats, args, 45)
        batch = _forward(net, x, t, cond, lengths)
        for b, n in enumerate(lengths):
            net.set_lengths(None, torch.device("cuda"))
            alone = _forward(net, x[b:b + 1, :, :, :n], t[b:b + 1], cond[b:b + 1, :, :n], None)
            assert np.array_equal(alone, batch[b:b + 1, :, :, :n]), (switch, b)
        net.release_native()
        os.environ.pop(switch)


def test_config5_per_gpu_share_full_size_vs_oracle():
    """BASELINE config 5's per-GPU share at FULL size: 8 utterances of 1000 frames through the pitch denoiser (WaveNet 20 x 256,
    cycle 5, 64 repeat bins) and the energy + breathiness denoiser (10 x 192, cycle 4, 2 x 24 bins), rectified flow, euler
    20 steps each (configs/variance.yaml:62-110) - the kernels bench.py --workload variance_reflow20 --batch 8 times:
    wn_layer_kernel<4, 48>, <4, 80>, <3, 48> - against the oracle's two 20-NFE loops (~20 s of host time)."""
    from diffsinger_amd.diffusion import MultiVarianceRectifiedFlow, PitchRectifiedFlow
    bsz, t_len = 8, 1000
    cond = synth.synth_normal((bsz, t_len, 256), 80)
    cond_t = np.ascontiguousarray(np.swapaxes(cond, 1, 2))
    set_hp(sampling_algorithm="euler", sampling_steps=20)
    pargs = dict(num_layers=20, num_channels=256, dilation_cycle_length=5)
    p = PitchRectifiedFlow(vmin=-8.0, vmax=8.0, cmin=-12.0, cmax=12.0, repeat_bins=64, backbone_type="wavenet",
                           backbone_args=pargs)
    params = synth_params("wavenet", 64, 1, pargs, 81)
    load_synth(p.velocity_fn, params)
    p = p.cuda().eval()
    noise = synth.synth_normal((bsz, 1, 64, t_len), 82)
    out = p(dev(cond), infer=True, noise=dev(noise))
    st = p.velocity_fn.stats()
    assert st["layer_launches"] == 1 and st["fused_tiles"] == 256, st
    assert st["kernels_per_nfe"] in (20 + 1, 20 + 3), st
    fn = lambda x, t, c: ob.wavenet_forward(params, x, t, c, dilation_cycle_length=5)       # noqa: E731
    nf, smin, smax = od.repetitive_spec_ranges(-8.0, 8.0)
    o = od.RectifiedFlow(fn, 64, nf, spec_min=smin, spec_max=smax)
    want = od.pitch_denorm(o, o.inference(cond_t, noise, sampling_algorithm="euler", sampling_steps=20), -12.0, 12.0)
    assert tuple(out.shape) == want.shape == (bsz, t_len)
    check(out, want, TOL_SAMPLER, what="config 5 pitch, B = 8, T = 1000, euler 20")
    p.velocity_fn.release_native()
    vargs = dict(num_layers=10, num_channels=192, dilation_cycle_length=4)
    ranges, clamps = [(-96.0, -12.0), (-96.0, -20.0)], [(-96.0, 0.0), (-96.0, 0.0)]
    m = MultiVarianceRectifiedFlow(ranges=ranges, clamps=clamps, repeat_bins=24, backbone_type="wavenet", backbone_args=vargs)
    params2 = synth_params("wavenet", 24, 2, vargs, 83)
    load_synth(m.velocity_fn, params2)
    m = m.cuda().eval()
    noise2 = synth.synth_normal((bsz, 2, 24, t_len), 84)
    outs = m(dev(cond), infer=True, noise=dev(noise2))
    st = m.velocity_fn.stats()
    assert st["layer_launches"] == 1 and st["fused_tiles"] == 256, st
    assert st["kernels_per_nfe"] in (10 + 1, 10 + 3), st
    fn2 = lambda x, t, c: ob.wavenet_forward(params2, x, t, c, dilation_cycle_length=4)     # noqa: E731
    nf, smin, smax = od.repetitive_spec_ranges([r[0] for r in ranges], [r[1] for r in ranges])
    orf = od.RectifiedFlow(fn2, 24, nf, spec_min=smin, spec_max=smax)
    want2 = od.multivar_denorm(orf, orf.inference(cond_t, noise2, sampling_algorithm="euler", sampling_steps=20), clamps)
    assert len(outs) == 2
    for i, (a, w) in enumerate(zip(outs, want2)):
        check(a, w, TOL_SAMPLER, what=("config 5 variances, B = 8, T = 1000, euler 20", i))
    m.velocity_fn.release_native()
