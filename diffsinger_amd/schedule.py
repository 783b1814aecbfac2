"""Host-side schedulers: every sampler of the reference becomes a *program* for libdsdenoise.

A program is a list of backbone evaluations; each evaluation names the state buffer fed to the
backbone, the model time, and up to three linear combinations (with scalar coefficients) of
{model output, state buffers, injected noise} that the library fuses into the last GEMM's
epilogue (include/dsdenoise.h, "Sampling programs").  The device therefore does only NFE + axpy;
all the scalar arithmetic the reference does with ~10-40 tiny device kernels per step
(`interpolate_fn`: sort + gather, dpm_solver_pytorch.py:1253-1292) happens here, once, on the host.

Scalar arithmetic mirrors the reference's own fp32 op order, using torch CPU scalar tensors where the
reference uses torch tensors: lambda/sigma are ill-conditioned in fp32 near t -> 0
(1 - exp(2*log_alpha) with log_alpha ~ -5e-5 keeps ~3 significant digits), so "the same formula in
float64" would NOT reproduce the reference's trajectory.  Composite coefficients (products of the
reference's scalars that multiply one tensor) are then formed in float64 and rounded once.

Restated from (not copied):
  ddpm.py:123-204,221-351 (DDPM / DDIM / PLMS + dispatcher), dpm_solver_pytorch.py:94-167,271-282,
  433-442,547-580,796-831,1171-1213 (DPM-Solver++ 2M), uni_pc.py:77-120,471-588,590-672 (UniPC bh2),
  reflow.py:66-138 (euler / rk2 / rk4 / rk5).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np
import torch

MODEL = -1            # DSD_SRC_MODEL
NOISE_BASE = -1000    # DSD_SRC_NOISE_BASE


def noise_src(k: int) -> int:
    return NOISE_BASE - k


# --------------------------------------------------------------------------------------------
# tiny linear-expression algebra over {MODEL, buffers, noise}
# --------------------------------------------------------------------------------------------
class Lin:
    __slots__ = ("c",)

    def __init__(self, c: Dict[int, float] | None = None):
        self.c = dict(c or {})

    @staticmethod
    def of(src: int, coef: float = 1.0) -> "Lin":
        return Lin({src: float(coef)})

    def __add__(self, o: "Lin") -> "Lin":
        r = dict(self.c)
        for k, v in o.c.items():
            r[k] = r.get(k, 0.0) + v
        return Lin(r)

    def __sub__(self, o: "Lin") -> "Lin":
        return self + (o * -1.0)

    def __mul__(self, s: float) -> "Lin":
        s = float(s)
        return Lin({k: v * s for k, v in self.c.items()})

    __rmul__ = __mul__

    def __truediv__(self, s: float) -> "Lin":
        return self * (1.0 / float(s))

    def terms(self) -> List[Tuple[int, float]]:
        # model output first, then buffers, then noise; exact zeros are dropped
        ks = sorted(self.c, key=lambda k: (k != MODEL, k < 0, abs(k)))
        out = [(k, self.c[k]) for k in ks if self.c[k] != 0.0]
        return out or [(ks[0], 0.0)]


@dataclass
class Eval:
    x_buf: int
    t: float
    outs: List[Tuple[int, List[Tuple[int, float]]]] = field(default_factory=list)

    def emit(self, dst: int, expr: Lin) -> None:
        self.outs.append((dst, expr.terms()))


@dataclass
class Program:
    n_bufs: int
    result_buf: int
    evals: List[Eval]
    n_noise: int = 0

    @property
    def nfe(self) -> int:
        return len(self.evals)

    def key(self) -> tuple:
        return (self.n_bufs, self.result_buf, self.n_noise,
                tuple((e.x_buf, float(np.float32(e.t)),
                       tuple((d, tuple((s, float(np.float32(c))) for s, c in ts)) for d, ts in e.outs))
                      for e in self.evals))


def _f(x) -> float:
    """python float (float64) of an fp32 scalar tensor / numpy scalar."""
    if isinstance(x, torch.Tensor):
        return float(x.reshape(-1)[0].item())
    return float(x)


# --------------------------------------------------------------------------------------------
# DDPM-family tables (ddpm.py:64-115)
# --------------------------------------------------------------------------------------------
def linear_beta_schedule(timesteps, max_beta=0.01):
    return np.linspace(1e-4, max_beta, timesteps)


def cosine_beta_schedule(timesteps, s=0.008):
    steps = timesteps + 1
    x = np.linspace(0, steps, steps)
    ac = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    ac = ac / ac[0]
    return np.clip(1 - (ac[1:] / ac[:-1]), a_min=0, a_max=0.999)


BETA_SCHEDULE = {"linear": linear_beta_schedule, "cosine": cosine_beta_schedule}


class DDPMTables:
    """The registered buffers of GaussianDiffusion, as float32 numpy arrays (float64 maths, one rounding)."""

    def __init__(self, betas):
        betas = np.asarray(betas, dtype=np.float64)
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        f = lambda a: np.asarray(a, dtype=np.float32)
        self.betas = f(betas)
        self.alphas_cumprod = f(ac)
        self.alphas_cumprod_prev = f(ac_prev)
        self.sqrt_alphas_cumprod = f(np.sqrt(ac))
        self.sqrt_one_minus_alphas_cumprod = f(np.sqrt(1.0 - ac))
        self.log_one_minus_alphas_cumprod = f(np.log(1.0 - ac))
        self.sqrt_recip_alphas_cumprod = f(np.sqrt(1.0 / ac))
        self.sqrt_recipm1_alphas_cumprod = f(np.sqrt(1.0 / ac - 1))
        pv = betas * (1.0 - ac_prev) / (1.0 - ac)
        self.posterior_variance = f(pv)
        self.posterior_log_variance_clipped = f(np.log(np.maximum(pv, 1e-20)))
        self.posterior_mean_coef1 = f(betas * np.sqrt(ac_prev) / (1.0 - ac))
        self.posterior_mean_coef2 = f((1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac))

    @classmethod
    def from_arrays(cls, arrays) -> "DDPMTables":
        """The tables as they STAND in a module's registered buffers (a loaded checkpoint may carry another schedule
        than the current hparams; the reference reads the buffers, ddpm.py:117-167)."""
        tb = cls.__new__(cls)
        for name in cls.NAMES:
            setattr(tb, name, np.ascontiguousarray(np.asarray(arrays[name], dtype=np.float32)))
        return tb

    NAMES = ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
             "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
             "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
             "posterior_mean_coef1", "posterior_mean_coef2")


X, TMP = 0, 1


def ddpm_ancestral_program(tb: DDPMTables, t_max: int, t_lo: int = 0, noise_index0: int = 0) -> Program:
    """p_sample for i = t_max-1 ... t_lo (ddpm.py:123-156,347-349). One injected noise tensor per step
    with i > 0 (the reference draws one for i == 0 too and multiplies it by zero)."""
    evals, k = [], noise_index0
    for i in reversed(range(t_lo, t_max)):
        sr, srm1 = float(tb.sqrt_recip_alphas_cumprod[i]), float(tb.sqrt_recipm1_alphas_cumprod[i])
        c1, c2 = float(tb.posterior_mean_coef1[i]), float(tb.posterior_mean_coef2[i])
        x_recon = Lin.of(X, sr) - Lin.of(MODEL, srm1)
        mean = x_recon * c1 + Lin.of(X, c2)
        ev = Eval(X, float(i))
        if i > 0:
            sigma = float(np.exp(np.float32(0.5) * tb.posterior_log_variance_clipped[i]))
            mean = mean + Lin.of(noise_src(k), sigma)
        k += 1
        ev.emit(X, mean)
        evals.append(ev)
    return Program(1, X, evals, n_noise=k)


def ddim_program(tb: DDPMTables, t_max: int, interval: int) -> Program:
    """p_sample_ddim (ddpm.py:158-167,334-343)."""
    evals = []
    ac = tb.alphas_cumprod.astype(np.float64)
    for i in reversed(range(0, t_max, interval)):
        a_t, a_prev = ac[i], ac[max(i - interval, 0)]
        c_eps = np.sqrt((1 - a_prev) / a_prev) - np.sqrt((1 - a_t) / a_t)
        expr = (Lin.of(X, 1.0 / np.sqrt(a_t)) + Lin.of(MODEL, c_eps)) * np.sqrt(a_prev)
        ev = Eval(X, float(i))
        ev.emit(X, expr)
        evals.append(ev)
    return Program(1, X, evals)


def plms_program(tb: DDPMTables, t_max: int, interval: int) -> Program:
    """p_sample_plms (ddpm.py:169-204,323-333): Adams-Bashforth 1..4 on eps, warm-up step with a 2nd NFE."""
    ac = tb.alphas_cumprod.astype(np.float64)
    hist0 = 2                       # ring of 4 eps buffers: 2..5

    def x_pred(x: Lin, n: Lin, i: int) -> Lin:
        a_t, a_prev = ac[i], ac[max(i - interval, 0)]
        a_t_sq, a_prev_sq = np.sqrt(a_t), np.sqrt(a_prev)
        c_x = 1.0 / (a_t_sq * (a_t_sq + a_prev_sq))
        c_n = 1.0 / (a_t_sq * (np.sqrt((1 - a_prev) * a_t) + np.sqrt((1 - a_t) * a_prev)))
        return x + (x * c_x - n * c_n) * (a_prev - a_t)

    evals: List[Eval] = []
    n_hist = 0
    for i in reversed(range(0, t_max, interval)):
        slot = hist0 + (n_hist % 4)
        prev = [hist0 + ((n_hist - j) % 4) for j in (1, 2, 3)]
        eps = Lin.of(MODEL)
        if n_hist == 0:
            e1 = Eval(X, float(i))
            e1.emit(slot, eps)
            e1.emit(TMP, x_pred(Lin.of(X), eps, i))
            evals.append(e1)
            e2 = Eval(TMP, float(max(i - interval, 0)))
            prime = (Lin.of(slot) + eps) / 2.0
            e2.emit(X, x_pred(Lin.of(X), prime, i))
            evals.append(e2)
        else:
            if n_hist == 1:
                prime = (eps * 3.0 - Lin.of(prev[0])) / 2.0
            elif n_hist == 2:
                prime = (eps * 23.0 - Lin.of(prev[0]) * 16.0 + Lin.of(prev[1]) * 5.0) / 12.0
            else:
                prime = (eps * 55.0 - Lin.of(prev[0]) * 59.0 + Lin.of(prev[1]) * 37.0 - Lin.of(prev[2]) * 9.0) / 24.0
            ev = Eval(X, float(i))
            ev.emit(slot, eps)
            ev.emit(X, x_pred(Lin.of(X), prime, i))
            evals.append(ev)
        n_hist += 1
    return Program(6, X, evals)


# --------------------------------------------------------------------------------------------
# discrete VP noise schedule in fp32 (dpm_solver_pytorch.py:94-167 / uni_pc.py:77-120)
# --------------------------------------------------------------------------------------------
class VPSchedule:
    def __init__(self, betas_f32: torch.Tensor, clip: bool):
        betas = torch.as_tensor(betas_f32, dtype=torch.float32)
        log_alphas = 0.5 * torch.log(1 - betas).cumsum(dim=0)
        if clip:   # numerical_clip_alpha, clipped_lambda = -5.1
            log_sigmas = 0.5 * torch.log(1. - torch.exp(2. * log_alphas))
            lambs = log_alphas - log_sigmas
            idx = int(torch.searchsorted(torch.flip(lambs, [0]), torch.tensor(-5.1)))
            if idx > 0:
                log_alphas = log_alphas[:-idx]
        self.T = 1.0
        self.log_alpha = log_alphas.to(torch.float32).contiguous()
        self.total_N = int(self.log_alpha.shape[0])
        self.t_array = torch.linspace(0., 1., self.total_N + 1)[1:].to(torch.float32).contiguous()

    def log_mean_coeff(self, t: torch.Tensor) -> torch.Tensor:
        """piecewise-linear interpolation of log_alpha at scalar t, same neighbour selection and the same
        fp32 expression as interpolate_fn (dpm_solver_pytorch.py:1253-1292)."""
        t = t.reshape(()).to(torch.float32)
        xp, yp, k = self.t_array, self.log_alpha, self.total_N
        idx = int(torch.searchsorted(xp, t, right=False))      # ties: x sorts before equal key points
        if idx == 0:
            x0, x1, y0, y1 = xp[0], xp[1], yp[0], yp[1]
        elif idx == k:
            x0, x1, y0, y1 = xp[k - 2], xp[k - 1], yp[k - 2], yp[k - 1]
        else:
            x0, x1, y0, y1 = xp[idx - 1], xp[idx], yp[idx - 1], yp[idx]
        return y0 + (t - x0) * (y1 - y0) / (x1 - x0)

    def alpha(self, t):
        return torch.exp(self.log_mean_coeff(t))

    def std(self, t):
        return torch.sqrt(1. - torch.exp(2. * self.log_mean_coeff(t)))

    def lam(self, t):
        lm = self.log_mean_coeff(t)
        return lm - 0.5 * torch.log(1. - torch.exp(2. * lm))

    def model_time(self, t):
        return (t - 1. / self.total_N) * self.total_N

    def time_uniform_steps(self, steps: int) -> torch.Tensor:
        return torch.linspace(self.T, 1. / self.total_N, steps + 1).to(torch.float32)


def dpm_solver_pp_program(betas_f32, steps: int) -> Program:
    """DPM_Solver(algorithm_type='dpmsolver++').sample(order=2, 'time_uniform', 'multistep')."""
    assert steps >= 2
    ns = VPSchedule(betas_f32, clip=True)
    ts = ns.time_uniform_steps(steps)
    XB, MA, MB = 0, 1, 2

    def data_pred(i, x: Lin) -> Lin:          # x0 = (x - sigma * eps) / alpha   (:433-442)
        a, s = _f(ns.alpha(ts[i])), _f(ns.std(ts[i]))
        return (x - Lin.of(MODEL, s)) / a

    def first_update(x: Lin, i_s, i_t, m_s: Lin) -> Lin:     # :569-580
        h = ns.lam(ts[i_t]) - ns.lam(ts[i_s])
        sig = ns.std(ts[i_t]) / ns.std(ts[i_s])
        coef = ns.alpha(ts[i_t]) * torch.expm1(-h)
        return x * _f(sig) - m_s * _f(coef)

    def second_update(x: Lin, i1, i0, i_t, m1: Lin, m0: Lin) -> Lin:    # :796-831
        lam1, lam0, lam_t = ns.lam(ts[i1]), ns.lam(ts[i0]), ns.lam(ts[i_t])
        h_0, h = lam0 - lam1, lam_t - lam0
        r0 = h_0 / h
        phi_1 = torch.expm1(-h)
        a_phi = ns.alpha(ts[i_t]) * phi_1
        sig = ns.std(ts[i_t]) / ns.std(ts[i0])
        d1_0 = (m0 - m1) * _f(1. / r0)
        return x * _f(sig) - m0 * _f(a_phi) - d1_0 * _f(0.5 * a_phi)

    evals: List[Eval] = []
    slots = [MA, MB]
    # eval i is the model call at ts[i] on x_i; it also produces x_{i+1}
    for i in range(steps):
        ev = Eval(XB, _f(ns.model_time(ts[i])))
        x = Lin.of(XB)
        m_new = data_pred(i, x)
        step = i + 1                                   # the update that leads to ts[step]
        if i == 0:
            nxt = first_update(x, 0, 1, m_new)
        else:
            order = min(2, steps + 1 - step) if steps < 10 else 2       # lower_order_final (:1198)
            if order == 1:
                nxt = first_update(x, i, step, m_new)
            else:
                nxt = second_update(x, i - 1, i, step, Lin.of(slots[(i - 1) % 2]), m_new)
        if i < steps - 1:
            ev.emit(slots[i % 2], m_new)
        ev.emit(XB, nxt)
        evals.append(ev)
    return Program(3, XB, evals)


def unipc_program(betas_f32, steps: int) -> Program:
    """UniPC(variant='bh2').sample(order=2, 'time_uniform', 'multistep'), data prediction."""
    assert steps >= 2
    ns = VPSchedule(betas_f32, clip=False)
    ts = ns.time_uniform_steps(steps)
    XP, XT, MA, MB = 0, 1, 2, 3
    slots = [MA, MB]

    def coeffs(i0, i_t, order, i1=None, use_corrector=True):
        """scalar pieces of multistep_uni_pc_bh_update (uni_pc.py:471-588) from ts[i0] to ts[i_t]."""
        lam0, lam_t = ns.lam(ts[i0]), ns.lam(ts[i_t])
        h = lam_t - lam0
        sig = ns.std(ts[i_t]) / ns.std(ts[i0])
        alpha_t = ns.alpha(ts[i_t])
        rks = []
        if order == 2:
            rks.append((ns.lam(ts[i1]) - lam0) / h)
        rks.append(torch.tensor(1.0))
        rks_t = torch.stack([r.reshape(()) for r in rks]).to(torch.float32)
        hh = -h
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        b_h = torch.expm1(hh)
        fact = 1
        rows, bvec = [], []
        for i in range(1, order + 1):
            rows.append(torch.pow(rks_t, i - 1))
            bvec.append((h_phi_k * fact / b_h).reshape(1))
            fact *= (i + 1)
            h_phi_k = h_phi_k / hh - 1 / fact
        r_mat, b_vec = torch.stack(rows), torch.cat(bvec)
        rhos_c = None
        if use_corrector:
            rhos_c = torch.tensor([0.5]) if order == 1 else torch.linalg.solve(r_mat, b_vec)
        return dict(sig=_f(sig), a_hphi1=_f(alpha_t * h_phi_1), a_bh=_f(alpha_t * b_h),
                    rk=_f(rks[0]) if order == 2 else None,
                    rhos_c=None if rhos_c is None else [float(v) for v in rhos_c])

    evals: List[Eval] = []
    # eval 0: model at ts[0] on x_0 -> m_0; predictor to ts[1] (order 1: x_pred = x_t_)
    ev = Eval(XP, _f(ns.model_time(ts[0])))
    a0, s0 = _f(ns.alpha(ts[0])), _f(ns.std(ts[0]))
    m0 = (Lin.of(XP) - Lin.of(MODEL, s0)) / a0
    c = coeffs(0, 1, 1)
    x_t_ = Lin.of(XP) * c["sig"] - m0 * c["a_hphi1"]
    ev.emit(slots[0], m0)
    ev.emit(XT, x_t_)
    ev.emit(XP, x_t_)
    evals.append(ev)
    pending = dict(c=c, order=1, m0_slot=slots[0], m1_slot=None)     # corrector data for the step landing on ts[1]
    for i in range(1, steps):
        # eval i: model at ts[i] on the predicted x (XP) -> m_t; corrector -> x_i; then predictor to ts[i+1]
        ev = Eval(XP, _f(ns.model_time(ts[i])))
        a_i, s_i = _f(ns.alpha(ts[i])), _f(ns.std(ts[i]))
        m_t = (Lin.of(XP) - Lin.of(MODEL, s_i)) / a_i
        pc = pending["c"]
        m_prev0 = Lin.of(pending["m0_slot"])
        corr = Lin()
        if pending["order"] == 2:
            d1 = (Lin.of(pending["m1_slot"]) - m_prev0) / pc["rk"]
            corr = d1 * pc["rhos_c"][0]
        x_i = Lin.of(XT) - (corr + (m_t - m_prev0) * pc["rhos_c"][-1]) * pc["a_bh"]
        # predictor towards ts[i+1] (uni_pc.py:628-642)
        step = i + 1
        order = min(2, steps + 1 - step)
        use_corr = step != steps
        c = coeffs(i, step, order, i1=i - 1, use_corrector=use_corr)
        x_t_ = x_i * c["sig"] - m_t * c["a_hphi1"]
        x_pred = x_t_
        if order == 2:
            d1 = (m_prev0 - m_t) / c["rk"]          # (model_prev_1 - model_prev_0) / rk, prev_0 is now m_t
            x_pred = x_t_ - d1 * (0.5 * c["a_bh"])
        new_slot = slots[i % 2]
        if use_corr:
            ev.emit(new_slot, m_t)
            ev.emit(XT, x_t_)
        ev.emit(XP, x_pred)
        evals.append(ev)
        pending = dict(c=c, order=order, m0_slot=new_slot, m1_slot=slots[(i - 1) % 2])
    return Program(4, XP, evals)


# --------------------------------------------------------------------------------------------
# rectified flow (reflow.py:66-138)
# --------------------------------------------------------------------------------------------
def reflow_program(algorithm: str, steps: int, t_start: float, time_scale_factor) -> Program:
    if algorithm not in ("euler", "rk2", "rk4", "rk5"):
        raise ValueError(f"Unsupported algorithm for Rectified Flow: {algorithm}.")
    dt = (1.0 - t_start) / max(1, steps)
    dts = torch.tensor([dt]).to(torch.float32)
    K1, K2, K3, K4, K5 = 2, 3, 4, 5, 6
    evals: List[Eval] = []

    def tt(t, off):       # time_scale_factor * (t + off*dt) in the reference's fp32 order
        return _f(time_scale_factor * (t + off * dt)) if off else _f(time_scale_factor * t)

    x, m = Lin.of(X), Lin.of(MODEL)
    for i in range(steps):
        t = t_start + i * dts
        if algorithm == "euler":
            e = Eval(X, tt(t, 0)); e.emit(X, x + m * dt); evals.append(e)
        elif algorithm == "rk2":
            e = Eval(X, tt(t, 0)); e.emit(TMP, x + m * (0.5 * dt)); evals.append(e)
            e = Eval(TMP, tt(t, 0.5)); e.emit(X, x + m * dt); evals.append(e)
        elif algorithm == "rk4":
            e = Eval(X, tt(t, 0)); e.emit(K1, m); e.emit(TMP, x + m * (0.5 * dt)); evals.append(e)
            e = Eval(TMP, tt(t, 0.5)); e.emit(K2, m); e.emit(TMP, x + m * (0.5 * dt)); evals.append(e)
            e = Eval(TMP, tt(t, 0.5)); e.emit(K3, m); e.emit(TMP, x + m * dt); evals.append(e)
            e = Eval(TMP, tt(t, 1.0))
            e.emit(X, x + (Lin.of(K1) + Lin.of(K2) * 2.0 + Lin.of(K3) * 2.0 + m) * (dt / 6.0)); evals.append(e)
        else:
            k1, k2, k3, k4, k5 = (Lin.of(b) for b in (K1, K2, K3, K4, K5))
            e = Eval(X, tt(t, 0)); e.emit(K1, m); e.emit(TMP, x + m * (0.25 * dt)); evals.append(e)
            e = Eval(TMP, tt(t, 0.25)); e.emit(K2, m); e.emit(TMP, x + (m + k1) * (0.125 * dt)); evals.append(e)
            e = Eval(TMP, tt(t, 0.25)); e.emit(K3, m); e.emit(TMP, x + (m * 2.0 - k2) * (0.5 * dt)); evals.append(e)
            e = Eval(TMP, tt(t, 0.5)); e.emit(K4, m); e.emit(TMP, x + (k1 * 3.0 + m * 9.0) * (0.0625 * dt)); evals.append(e)
            e = Eval(TMP, tt(t, 0.75)); e.emit(K5, m)
            e.emit(TMP, x + (k1 * -3.0 + k2 * 2.0 + k3 * 12.0 - k4 * 12.0 + m * 8.0) * (dt / 7.0)); evals.append(e)
            e = Eval(TMP, tt(t, 1.0))
            e.emit(X, x + (k1 * 7.0 + k3 * 32.0 + k4 * 12.0 + k5 * 32.0 + m * 7.0) * (dt / 90.0)); evals.append(e)
    return Program(7 if algorithm == "rk5" else (5 if algorithm == "rk4" else 2), X, evals)


# --------------------------------------------------------------------------------------------
# Runtime inputs of the ONNX deployment twins (deployment/modules/diffusion.py:105-131,
# deployment/modules/rectified_flow.py:37-68): `steps` and `depth` arrive per call instead of through hparams
# --------------------------------------------------------------------------------------------
def onnx_ddpm_plan(timesteps: int, k_step: int, factors, steps: int, depth=None) -> Tuple[int, int]:
    """(t_max, speedup) as GaussianDiffusionONNX.forward derives them.

    depth None (no shallow source): speed-up = timesteps // steps snapped DOWN to a factor of `timesteps`
    (`timestep_factors`), loop over [0, k_step).  Otherwise depth * timesteps is rounded (fp32, half to even, as
    torch.round), capped at k_step, the speed-up is depth // steps - NOT snapped - and the depth is rounded down to a
    multiple of it."""
    if depth is None:
        speedup = max(1, timesteps // steps)
        f = torch.as_tensor(factors)
        return k_step, int(f[int(torch.sum(f <= speedup)) - 1])
    d = torch.round(torch.as_tensor(depth, dtype=torch.float32) * timesteps).long()
    depth_i = min(int(d), k_step)
    speedup = max(1, depth_i // steps)
    return depth_i // speedup * speedup, speedup


def reflow_onnx_program(steps: int, t_start, time_scale_factor) -> Program:
    """RectifiedFlowONNX's euler loop: dt = (1 - t_start) / max(1, steps) and the step times i * dt + t_start are fp32
    tensor arithmetic there (rectified_flow.py:58-62), not the Python-float arithmetic of reflow.py:117-126."""
    ts = torch.as_tensor(t_start, dtype=torch.float32)
    dt = (1.0 - ts) / max(1, steps)
    times = torch.arange(steps, dtype=torch.long).float() * dt + ts
    x, m = Lin.of(X), Lin.of(MODEL)
    evals: List[Eval] = []
    for i in range(steps):
        e = Eval(X, _f(times[i] * time_scale_factor))
        e.emit(X, x + m * _f(dt))
        evals.append(e)
    return Program(2, X, evals)
