"""The two wave layouts of the row-split conv (wn_rowsplit.hip: K halves x four row waves, K quarters x two 32-row waves),
each forced on (DSD_RS_CONV_Q, read per launch) on grids of either side of the 256-workgroup rule that picks between them
(and the 48-frame tiles of the K-quarter layout on the dense launches that select them):
one evaluation against the numpy oracle, at the acoustic shape (dilation <= 8), the pitch shape (dilation 16: the 80-float
row stride), a ragged batch and tiles cut by the utterance end; and the two layouts against each other - they add the same
products in a different order, so they may differ by rounding only."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from diffsinger_amd import synth  # noqa: E402
from gpu_util import check, dev, make_backbone, set_hp  # noqa: E402
from oracle import backbones as ob  # noqa: E402

TOL_NFE = 2e-5


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    set_hp()
    yield
    os.environ.pop("DSD_RS_CONV_Q", None)


def _run(layout, in_dims, args, bsz, t_len, lengths=None):
    os.environ["DSD_RS_CONV_Q"] = layout
    try:
        net, params = make_backbone("wavenet", in_dims, 1, args, 42)
        x = synth.synth_normal((bsz, 1, in_dims, t_len), 21)
        cond = synth.synth_normal((bsz, 256, t_len), 22)
        t = (np.arange(bsz) * 211.5 + 3.25).astype(np.float32)
        xd = dev(x)
        if lengths is not None:
            net.set_lengths(lengths, xd.device)
        with torch.no_grad():
            out = net(xd, dev(t), dev(cond))
            again = net(xd, dev(t), dev(cond))
        torch.cuda.synchronize()
        assert torch.equal(out, again)
        stats = net.stats()
        net.release_native()
        return out.cpu().numpy(), params, x, t, cond, stats
    finally:
        os.environ.pop("DSD_RS_CONV_Q", None)


CASES = {
    "acoustic_T1000": (128, dict(num_layers=8, num_channels=256, dilation_cycle_length=4), 1, 1000, None),     # 256 workgroups
    "acoustic_T777": (128, dict(num_layers=4, num_channels=256, dilation_cycle_length=4), 1, 777, None),       # last tile cut at 9 frames
    "acoustic_T2048": (128, dict(num_layers=4, num_channels=256, dilation_cycle_length=4), 1, 2048, None),     # 512 workgroups
    "pitch_T900": (64, dict(num_layers=5, num_channels=256, dilation_cycle_length=5), 1, 900, None),           # dilation 16 in layer 4
    "ragged_B2": (128, dict(num_layers=4, num_channels=256, dilation_cycle_length=4), 2, 640, [640, 333]),     # tile list, per-item ends
    # 48-frame tiles (K-quarter layout only): dense launches that they turn into one round of workgroups
    "acoustic_T1100_bn48": (128, dict(num_layers=4, num_channels=256, dilation_cycle_length=4), 1, 1100, None),  # 23 tiles, last cut at 44
    "acoustic_T1088_bn48": (128, dict(num_layers=4, num_channels=256, dilation_cycle_length=4), 1, 1088, None),  # tiles reach 16 past 64 m
    "acoustic_T1536_bn48": (128, dict(num_layers=4, num_channels=256, dilation_cycle_length=4), 1, 1536, None),  # 32 full tiles
    "pitch_T1300_bn48": (64, dict(num_layers=5, num_channels=256, dilation_cycle_length=5), 1, 1300, None),      # dilation 16
    "acoustic_B2_T700_bn48": (128, dict(num_layers=4, num_channels=256, dilation_cycle_length=4), 2, 700, None), # 2 x 15 tiles
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_rowsplit_conv_layouts_vs_oracle(name):
    if os.environ.get("DSD_FUSED_LAYER") == "1" or os.environ.get("DSD_ROWSPLIT") == "0":
        pytest.skip("a forced-path run that takes the row-split pair out of the loop")
    in_dims, args, bsz, t_len, lengths = CASES[name]
    outs = {}
    for layout in ("0", "1"):
        out, params, x, t, cond, stats = _run(layout, in_dims, args, bsz, t_len, lengths)
        assert stats["kernels_per_nfe"] == 2 * args["num_layers"] + 3, stats          # two launches per layer: the split path
        want = ob.wavenet_forward(params, x, t, cond, dilation_cycle_length=args["dilation_cycle_length"])
        if lengths is not None:                       # frames past an item's end are the caller's to mask (toplevel.py:104)
            for b, n in enumerate(lengths):
                want_b = ob.wavenet_forward(params, x[b:b + 1, :, :, :n], t[b:b + 1], cond[b:b + 1, :, :n],
                                            dilation_cycle_length=args["dilation_cycle_length"])
                check(out[b:b + 1, :, :, :n], want_b, TOL_NFE, what=("row-split conv layout " + layout, name, b))
        else:
            check(out, want, TOL_NFE, what=("row-split conv layout " + layout, name))
        outs[layout] = out
    if lengths is None:
        check(outs["1"], outs["0"], 2e-6, what=("K quarters vs K halves", name))
    else:
        for b, n in enumerate(lengths):
            check(outs["1"][b:b + 1, :, :, :n], outs["0"][b:b + 1, :, :, :n], 2e-6, what=("K quarters vs K halves", name, b))


# ---- wide row tiles (wn_rows.hip): 128 / 256 rows per workgroup, forced with DSD_RS_ROWS (+ DSD_FUSED_LAYER=0: the whole layer on them)
WIDE_CASES = {
    "acoustic_B2_T1000": (128, dict(num_layers=8, num_channels=256, dilation_cycle_length=4), 2, 1000, None),   # 64 tiles
    "acoustic_T777": (128, dict(num_layers=4, num_channels=256, dilation_cycle_length=4), 1, 777, None),         # last tile cut at 9 frames
    "acoustic_T13_B3": (128, dict(num_layers=5, num_channels=256, dilation_cycle_length=5), 3, 13, None),        # T < dilation 16
    "pitch_T900": (64, dict(num_layers=5, num_channels=256, dilation_cycle_length=5), 1, 900, None),             # dilation 16: 80-float stride
    "ragged_B3": (128, dict(num_layers=4, num_channels=256, dilation_cycle_length=4), 3, 640, [640, 333, 70]),   # tile list, per-item ends
    "pitch_ragged_B2": (64, dict(num_layers=5, num_channels=256, dilation_cycle_length=5), 2, 200, [200, 9]),    # dilation 16, an item of 9 frames
    "acoustic_T160_B3": (128, dict(num_layers=4, num_channels=256, dilation_cycle_length=4), 3, 160, None),     # mixed plan: 15 tiles split 7 | 8, inside item 1
}


@pytest.mark.parametrize("rows", ["128", "256"])
@pytest.mark.parametrize("name", sorted(WIDE_CASES))
def test_wide_row_tiles_vs_oracle(name, rows):
    in_dims, args, bsz, t_len, lengths = WIDE_CASES[name]
    saved = {k: os.environ.pop(k, None) for k in ("DSD_RS_ROWS", "DSD_FUSED_LAYER", "DSD_WN_PLAN")}
    os.environ["DSD_RS_ROWS"] = rows
    os.environ["DSD_FUSED_LAYER"] = "0"
    try:
        net, params = make_backbone("wavenet", in_dims, 1, args, 42)
        x = synth.synth_normal((bsz, 1, in_dims, t_len), 21)
        cond = synth.synth_normal((bsz, 256, t_len), 22)
        t = (np.arange(bsz) * 211.5 + 3.25).astype(np.float32)
        xd = dev(x)
        if lengths is not None:
            net.set_lengths(lengths, xd.device)
        with torch.no_grad():
            out = net(xd, dev(t), dev(cond))
            again = net(xd, dev(t), dev(cond))
        torch.cuda.synchronize()
        assert torch.equal(out, again)
        st = net.stats()
        tiles = sum((n + 31) // 32 for n in lengths) if lengths else bsz * ((t_len + 31) // 32)
        assert st["layer_launches"] == 2 and st["split_tiles"] == tiles and st["fused_tiles"] == 0, st
        out = out.cpu().numpy()
        cyc = args["dilation_cycle_length"]
        if lengths is None:
            check(out, ob.wavenet_forward(params, x, t, cond, dilation_cycle_length=cyc), TOL_NFE, what=("wide rows", rows, name))
        else:
            for b, n in enumerate(lengths):
                want_b = ob.wavenet_forward(params, x[b:b + 1, :, :, :n], t[b:b + 1], cond[b:b + 1, :, :n], dilation_cycle_length=cyc)
                check(out[b:b + 1, :, :, :n], want_b, TOL_NFE, what=("wide rows", rows, name, b))
        # ... and as the remainder segment of a mixed plan (first half of the tiles on the fused kernel)
        os.environ.pop("DSD_FUSED_LAYER")
        os.environ["DSD_WN_PLAN"] = "2"
        with torch.no_grad():
            mixed = net(xd, dev(t), dev(cond))
        torch.cuda.synchronize()
        st = net.stats()
        assert st["fused_tiles"] == tiles // 2 and st["split_tiles"] == tiles - tiles // 2, st
        mixed = mixed.cpu().numpy()
        if lengths is None:
            check(mixed, out, 8e-6, what=("wide rows: mixed plan vs whole layer", rows, name))
        else:
            for b, n in enumerate(lengths):
                check(mixed[b:b + 1, :, :, :n], out[b:b + 1, :, :, :n], 8e-6, what=("wide rows: mixed plan vs whole layer", rows, name, b))
        net.release_native()
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
