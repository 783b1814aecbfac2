#!/usr/bin/env python3
"""What a split-bf16 ("bf16x3") mode of the layer GEMMs would cost in accuracy - CPU emulation on the numpy oracle, no GPU.

Each fp32 operand is split x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (round to nearest even), and a product A.B is
evaluated as hi(A).hi(B) + hi(A).lo(B) + lo(A).hi(B) with fp32 accumulation - what three v_mfma_f32_*_bf16 per fp32 MFMA would
compute (the lo.lo term, ~2^-16 relative, is dropped).  Applied to the dilated conv, the conditioner-free 1x1 convs of every
residual layer and the in / skip / output projections of the 20 x 256 WaveNet; compared with the plain fp32 oracle on one
evaluation and on the 50-NFE DPM-Solver++ loop of BASELINE configs[1] (T = 200 so that it runs in a minute).
Prints max / rms relative differences; DESIGN.md section 9 quotes them."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffsinger_amd import synth  # noqa: E402
from oracle import backbones as ob  # noqa: E402
from oracle import diffusion as od  # noqa: E402

F32 = np.float32


def bf16(x):
    u = np.ascontiguousarray(x, F32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16          # round to nearest even on the upper 16 bits
    return r.astype(np.uint32).view(F32).reshape(np.shape(x))


def split(x):
    hi = bf16(x)
    return hi, bf16((x - hi).astype(F32))


def matmul3(a, b):
    ah, al = split(a)
    bh, bl = split(b)
    return (np.matmul(ah, bh) + np.matmul(ah, bl) + np.matmul(al, bh)).astype(F32)


def conv1x1_x3(x, w, b):
    return (matmul3(w[:, :, 0], x) + b[None, :, None]).astype(F32)


def dilated_conv3_x3(x, w, b, dil):
    bsz, cin, t = x.shape
    xp = np.zeros((bsz, cin, t + 2 * dil), dtype=F32)
    xp[:, :, dil:dil + t] = x
    y = np.zeros((bsz, w.shape[0], t), dtype=F32)
    for k in range(3):
        y += matmul3(np.ascontiguousarray(w[:, :, k]), xp[:, :, k * dil:k * dil + t])
    return (y + b[None, :, None]).astype(F32)


def errs(a, b):
    d = np.asarray(a, np.float64) - np.asarray(b, np.float64)
    return np.abs(d).max() / np.abs(b).max(), np.sqrt((d * d).mean()) / np.sqrt((np.asarray(b, np.float64) ** 2).mean())


def main():
    args = dict(num_layers=20, num_channels=256, dilation_cycle_length=4)
    params = synth.synth_state_dict(synth.backbone_param_shapes("wavenet", 128, 1, hidden_size=256, **args), seed=42)
    t_len = 200
    x = synth.synth_normal((1, 1, 128, t_len), 21)
    cond = synth.synth_normal((1, 256, t_len), 22)
    t = np.array([412.0], F32)
    plain = (ob._conv1x1, ob._dilated_conv3)

    def fwd(xx, tt, cc):
        return ob.wavenet_forward(params, xx, tt, cc, dilation_cycle_length=4)

    ref1 = fwd(x, t, cond)
    ob._conv1x1, ob._dilated_conv3 = conv1x1_x3, dilated_conv3_x3
    try:
        x3_1 = fwd(x, t, cond)
    finally:
        ob._conv1x1, ob._dilated_conv3 = plain
    print("one evaluation, bf16x3 vs fp32 oracle: max rel %.3e  rms rel %.3e" % errs(x3_1, ref1))

    condT = np.ascontiguousarray(np.swapaxes(cond, 1, 2))
    noise = synth.synth_normal((1, 1, 128, t_len), 1)
    o = od.GaussianDiffusion(fwd, 128, 1, spec_min=[-12.0], spec_max=[0.0])
    ref = o.forward(condT, noise, diff_accelerator="dpm-solver", diff_speedup=20, K_step_infer=1000)
    ob._conv1x1, ob._dilated_conv3 = conv1x1_x3, dilated_conv3_x3
    try:
        got = o.forward(condT, noise, diff_accelerator="dpm-solver", diff_speedup=20, K_step_infer=1000)
    finally:
        ob._conv1x1, ob._dilated_conv3 = plain
    print("DPM-Solver++ 1000->50 (50 NFE), bf16x3 vs fp32 oracle: max rel %.3e  rms rel %.3e" % errs(got, ref))
    # for scale: plain bf16 operands (one product, what a naive bf16 mode would do)
    ob._conv1x1 = lambda xx, w, b: (np.matmul(bf16(w[:, :, 0]), bf16(xx)) + b[None, :, None]).astype(F32)

    def dc1(xx, w, b, dil):
        return plain[1](bf16(xx), bf16(w), b, dil)
    ob._dilated_conv3 = dc1
    try:
        b1 = fwd(x, t, cond)
    finally:
        ob._conv1x1, ob._dilated_conv3 = plain
    print("one evaluation, plain bf16 operands vs fp32 oracle: max rel %.3e  rms rel %.3e" % errs(b1, ref1))


if __name__ == "__main__":
    main()
