"""numpy executor of schedule.Program with exactly the semantics libdsdenoise gives it
(per evaluation: all linear combinations read the PRE-evaluation buffers, then write)."""
import numpy as np

from diffsinger_amd import schedule

F32 = np.float32


def run_program(prog, model, x_init, cond, noise=None):
    """model(x[B,F,M,T], t[B] float32, cond) -> [B,F,M,T]; returns the result buffer [B,F,M,T]."""
    bufs = [np.zeros_like(x_init, dtype=F32) for _ in range(prog.n_bufs)]
    bufs[0] = np.asarray(x_init, dtype=F32).copy()
    bsz = x_init.shape[0]
    for ev in prog.evals:
        eps = model(bufs[ev.x_buf], np.full((bsz,), F32(ev.t), dtype=F32), cond)
        new = []
        for dst, terms in ev.outs:
            acc = np.zeros_like(x_init, dtype=F32)
            for src, coef in terms:
                if src == schedule.MODEL:
                    v = eps
                elif src >= 0:
                    v = bufs[src]
                else:
                    v = noise[schedule.NOISE_BASE - src]
                acc = (acc + F32(coef) * v).astype(F32)
            new.append((dst, acc))
        for dst, acc in new:
            bufs[dst] = acc
    return bufs[prog.result_buf]


def finish(x, spec_min=-12.0, spec_max=0.0):
    """x.transpose(2,3).squeeze(1) + denorm_spec (ddpm.py:350,382-383)."""
    x = np.swapaxes(x, 2, 3)
    if x.shape[1] == 1:
        x = x[:, 0]
    return ((x + F32(1)) / F32(2) * F32(spec_max - spec_min) + F32(spec_min)).astype(F32)
