"""numpy fp32 restatement of the reference sampling loops (TEST ORACLE).

Follows:
  * beta schedules, buffers       modules/core/ddpm.py:28-52, 64-115
  * q_sample / p_sample / DDIM / PLMS / inference dispatch
                                  modules/core/ddpm.py:123-351
  * norm_spec / denorm_spec and the Repetitive / Pitch / MultiVariance
    variants                      modules/core/ddpm.py:379-505, reflow.py:140-261
  * NoiseScheduleVP, model_wrapper time mapping, DPM-Solver++ multistep order 2
                                  inference/dpm_solver_pytorch.py:94-167, 271-282,
                                  433-442, 547-580, 796-831, 1171-1213, 1253-1292
  * UniPC bh2 multistep order 2   inference/uni_pc.py:77-120, 471-588, 590-672
  * RectifiedFlow euler/rk2/rk4/rk5
                                  modules/core/reflow.py:66-138

All randomness is injected (x_T and the per-step noise of ancestral DDPM) so
the oracle, the reference and the HIP path can be run on identical inputs.
The backbone is a callable `fn(x[B,F,M,T], t[B] or [1], cond[B,H,T]) -> [B,F,M,T]`.
"""
from __future__ import annotations

from collections import deque

import numpy as np

F32 = np.float32


def _f(v):
    return np.asarray(v, dtype=F32)


# ----------------------------------------------------------------------------
# schedules (ddpm.py:28-52)
# ----------------------------------------------------------------------------
def linear_beta_schedule(timesteps, max_beta=0.01):
    return np.linspace(1e-4, max_beta, timesteps)


def cosine_beta_schedule(timesteps, s=0.008):
    steps = timesteps + 1
    x = np.linspace(0, steps, steps)
    ac = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = 1 - (ac[1:] / ac[:-1])
    return np.clip(betas, a_min=0, a_max=0.999)


BETA_SCHEDULE = {"linear": linear_beta_schedule, "cosine": cosine_beta_schedule}


def torch_linspace_f32(start, end, steps):
    """torch.linspace(start, end, steps) in float32, scalar formula of ATen's
    RangeFactories kernel: first half counts up from start, second half down from end."""
    start, end = F32(start), F32(end)
    step = F32((end - start) / F32(steps - 1))
    out = np.empty(steps, dtype=F32)
    half = steps // 2
    for i in range(steps):
        if i < half:
            out[i] = F32(start + F32(step * F32(i)))
        else:
            out[i] = F32(end - F32(step * F32(steps - i - 1)))
    return out


# ----------------------------------------------------------------------------
# piecewise-linear interpolation (dpm_solver_pytorch.py:1253-1292)
# ----------------------------------------------------------------------------
def interpolate_fn(x, xp, yp):
    """Scalar x against key points xp/yp (1-D float32, xp ascending).

    The reference sorts [x, xp...] and gathers neighbours; this is the same
    selection written with a binary search (ties: x sorts first)."""
    x = F32(x)
    k = xp.shape[0]
    x_idx = int(np.searchsorted(xp, x, side="left"))

    def sorted_all(i):
        if i < x_idx:
            return xp[i]
        if i == x_idx:
            return x
        return xp[i - 1]

    cand = x_idx - 1
    if x_idx == 0:
        start_idx, start_idx2 = 1, 0
    elif x_idx == k:
        start_idx, start_idx2 = k - 2, k - 2
    else:
        start_idx, start_idx2 = cand, cand
    end_idx = start_idx + 2 if start_idx == cand else start_idx + 1
    start_x, end_x = F32(sorted_all(start_idx)), F32(sorted_all(end_idx))
    start_y, end_y = F32(yp[start_idx2]), F32(yp[start_idx2 + 1])
    return F32(start_y + F32(F32(F32(x - start_x) * F32(end_y - start_y)) / F32(end_x - start_x)))


class NoiseScheduleVP:
    """Discrete VP schedule; `clip` selects the DPM-Solver flavour (lambda clipping at
    -5.1, dpm_solver_pytorch.py:114-125) or the UniPC flavour (none, uni_pc.py:77-86)."""

    def __init__(self, betas_f32, clip):
        betas = _f(betas_f32)
        log_alphas = (F32(0.5) * np.cumsum(np.log((F32(1) - betas).astype(F32)), dtype=F32)).astype(F32)
        if clip:
            log_sigmas = F32(0.5) * np.log(F32(1) - np.exp(F32(2) * log_alphas))
            lambs = (log_alphas - log_sigmas).astype(F32)
            idx = int(np.searchsorted(lambs[::-1], F32(-5.1), side="left"))
            if idx > 0:
                log_alphas = log_alphas[:-idx]
        self.T = 1.0
        self.log_alpha_array = log_alphas.astype(F32)
        self.total_N = int(log_alphas.shape[0])
        self.t_array = torch_linspace_f32(0.0, 1.0, self.total_N + 1)[1:]

    def marginal_log_mean_coeff(self, t):
        return interpolate_fn(t, self.t_array, self.log_alpha_array)

    def marginal_alpha(self, t):
        return F32(np.exp(self.marginal_log_mean_coeff(t)))

    def marginal_std(self, t):
        return F32(np.sqrt(F32(1) - np.exp(F32(2) * self.marginal_log_mean_coeff(t))))

    def marginal_lambda(self, t):
        lm = self.marginal_log_mean_coeff(t)
        log_std = F32(F32(0.5) * np.log(F32(1) - np.exp(F32(2) * lm)))
        return F32(lm - log_std)

    def model_time(self, t):
        # get_model_input_time (dpm_solver_pytorch.py:271-282): (t - 1/N) * N
        return F32(F32(F32(t) - F32(1.0 / self.total_N)) * F32(self.total_N))


# ----------------------------------------------------------------------------
# DPM-Solver++ (2M) and UniPC (bh2, order 2)
# ----------------------------------------------------------------------------
def dpm_solver_pp_2m(fn, x, cond, ns: NoiseScheduleVP, steps, trace=None):
    """DPM_Solver.sample(method='multistep', order=2, skip_type='time_uniform'),
    algorithm_type='dpmsolver++', solver_type='dpmsolver' (dpm_solver_pytorch.py:1171-1213)."""
    bsz = x.shape[0]
    t0 = 1.0 / ns.total_N
    ts = torch_linspace_f32(ns.T, t0, steps + 1)
    assert steps >= 2

    def model_fn(xx, t):      # data prediction (:433-442)
        t_in = np.full((bsz,), ns.model_time(t), dtype=F32)
        eps = fn(xx, t_in, cond)
        alpha, sigma = ns.marginal_alpha(t), ns.marginal_std(t)
        return ((xx - sigma * eps) / alpha).astype(F32)

    def first_update(xx, s, t, model_s):                         # :569-580
        lam_s, lam_t = ns.marginal_lambda(s), ns.marginal_lambda(t)
        h = F32(lam_t - lam_s)
        sigma_s, sigma_t = ns.marginal_std(s), ns.marginal_std(t)
        alpha_t = F32(np.exp(ns.marginal_log_mean_coeff(t)))
        phi_1 = F32(np.expm1(-h))
        return (F32(sigma_t / sigma_s) * xx - F32(alpha_t * phi_1) * model_s).astype(F32)

    def second_update(xx, m_list, t_list, t):                    # :796-831
        m1, m0 = m_list[-2], m_list[-1]
        tp1, tp0 = t_list[-2], t_list[-1]
        lam1, lam0, lam_t = ns.marginal_lambda(tp1), ns.marginal_lambda(tp0), ns.marginal_lambda(t)
        sigma0, sigma_t = ns.marginal_std(tp0), ns.marginal_std(t)
        alpha_t = F32(np.exp(ns.marginal_log_mean_coeff(t)))
        h_0 = F32(lam0 - lam1)
        h = F32(lam_t - lam0)
        r0 = F32(h_0 / h)
        d1_0 = (F32(F32(1.0) / r0) * (m0 - m1)).astype(F32)
        phi_1 = F32(np.expm1(-h))
        return (F32(sigma_t / sigma0) * xx
                - F32(alpha_t * phi_1) * m0
                - F32(F32(0.5) * F32(alpha_t * phi_1)) * d1_0).astype(F32)

    t = ts[0]
    t_list = [t]
    m_list = [model_fn(x, t)]
    t = ts[1]
    x = first_update(x, t_list[-1], t, m_list[-1])
    if trace is not None:
        trace.append(x.copy())
    t_list.append(t)
    m_list.append(model_fn(x, t))
    for step in range(2, steps + 1):
        t = ts[step]
        if steps < 10:                                           # lower_order_final (:1198)
            order = min(2, steps + 1 - step)
        else:
            order = 2
        if order == 1:
            x = first_update(x, t_list[-1], t, m_list[-1])
        else:
            x = second_update(x, m_list, t_list, t)
        if trace is not None:
            trace.append(x.copy())
        t_list[0], m_list[0] = t_list[1], m_list[1]
        t_list[1] = t
        if step < steps:
            m_list[1] = model_fn(x, t)
    return x


def unipc_bh2(fn, x, cond, ns: NoiseScheduleVP, steps, trace=None):
    """UniPC.sample(method='multistep', order=2, variant='bh2') with data prediction
    (uni_pc.py:590-672; update :471-588)."""
    bsz = x.shape[0]
    t0 = 1.0 / ns.total_N
    ts = torch_linspace_f32(ns.T, t0, steps + 1)
    assert steps >= 2

    def model_fn(xx, t):
        t_in = np.full((bsz,), ns.model_time(t), dtype=F32)
        eps = fn(xx, t_in, cond)
        alpha, sigma = ns.marginal_alpha(t), ns.marginal_std(t)
        return ((xx - sigma * eps) / alpha).astype(F32)

    def bh_update(xx, m_list, t_list, t, order, use_corrector):
        t_prev_0 = t_list[-1]
        lam0, lam_t = ns.marginal_lambda(t_prev_0), ns.marginal_lambda(t)
        m0 = m_list[-1]
        sigma0, sigma_t = ns.marginal_std(t_prev_0), ns.marginal_std(t)
        alpha_t = F32(np.exp(ns.marginal_log_mean_coeff(t)))
        h = F32(lam_t - lam0)
        rks, d1s = [], []
        for i in range(1, order):
            lam_i = ns.marginal_lambda(t_list[-(i + 1)])
            rk = F32(F32(lam_i - lam0) / h)
            rks.append(rk)
            d1s.append(((m_list[-(i + 1)] - m0) / rk).astype(F32))
        rks.append(F32(1.0))
        rks = np.asarray(rks, dtype=F32)
        hh = F32(-h)
        h_phi_1 = F32(np.expm1(hh))
        h_phi_k = F32(F32(h_phi_1 / hh) - F32(1))
        b_h = F32(np.expm1(hh))                                  # variant bh2
        fact = 1
        r_rows, b_vec = [], []
        for i in range(1, order + 1):
            r_rows.append(np.power(rks, F32(i - 1)).astype(F32))
            b_vec.append(F32(F32(h_phi_k * F32(fact)) / b_h))
            fact *= (i + 1)
            h_phi_k = F32(F32(h_phi_k / hh) - F32(1.0 / fact))
        r_mat = np.stack(r_rows).astype(F32)
        b_vec = np.asarray(b_vec, dtype=F32)
        rhos_c = None
        if use_corrector:
            rhos_c = _f([0.5]) if order == 1 else np.linalg.solve(r_mat, b_vec).astype(F32)
        x_t_ = (F32(sigma_t / sigma0) * xx - F32(alpha_t * h_phi_1) * m0).astype(F32)
        if d1s:
            pred_res = (F32(0.5) * d1s[0]).astype(F32)           # order 2: rhos_p = [0.5]
            x_t = (x_t_ - F32(alpha_t * b_h) * pred_res).astype(F32)
        else:
            x_t = x_t_
        model_t = None
        if use_corrector:
            model_t = model_fn(x_t, t)
            corr_res = (rhos_c[0] * d1s[0]).astype(F32) if d1s else F32(0)
            d1_t = (model_t - m0).astype(F32)
            x_t = (x_t_ - F32(alpha_t * b_h) * (corr_res + rhos_c[-1] * d1_t)).astype(F32)
        return x_t, model_t

    t = ts[0]
    t_list = [t]
    m_list = [model_fn(x, t)]
    t = ts[1]
    x, model_x = bh_update(x, m_list, t_list, t, 1, True)
    if trace is not None:
        trace.append(x.copy())
    t_list.append(t)
    m_list.append(model_x)
    for step in range(2, steps + 1):
        t = ts[step]
        order = min(2, steps + 1 - step)                         # lower_order_final always on
        use_corrector = step != steps
        x, model_x = bh_update(x, m_list, t_list, t, order, use_corrector)
        if trace is not None:
            trace.append(x.copy())
        t_list[0], m_list[0] = t_list[1], m_list[1]
        t_list[1] = t
        if step < steps:
            if model_x is None:
                model_x = model_fn(x, t)
            m_list[1] = model_x
    return x


# ----------------------------------------------------------------------------
# GaussianDiffusion (ddpm.py:55-383)
# ----------------------------------------------------------------------------
class GaussianDiffusion:
    def __init__(self, denoise_fn, out_dims, num_feats=1, timesteps=1000, k_step=1000,
                 spec_min=None, spec_max=None, schedule_type="linear",
                 use_shallow_diffusion=False, betas=None):
        self.denoise_fn = denoise_fn
        self.out_dims, self.num_feats = out_dims, num_feats
        if betas is None:
            betas = BETA_SCHEDULE[schedule_type](timesteps)
        betas = np.asarray(betas, dtype=np.float64)
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        self.use_shallow_diffusion = use_shallow_diffusion
        if use_shallow_diffusion:
            assert k_step <= timesteps, "K_step should not be larger than timesteps."
        self.timesteps = timesteps
        self.k_step = k_step if use_shallow_diffusion else timesteps
        self.betas = _f(betas)
        self.alphas_cumprod = _f(ac)
        self.alphas_cumprod_prev = _f(ac_prev)
        self.sqrt_alphas_cumprod = _f(np.sqrt(ac))
        self.sqrt_one_minus_alphas_cumprod = _f(np.sqrt(1.0 - ac))
        self.sqrt_recip_alphas_cumprod = _f(np.sqrt(1.0 / ac))
        self.sqrt_recipm1_alphas_cumprod = _f(np.sqrt(1.0 / ac - 1))
        pv = betas * (1.0 - ac_prev) / (1.0 - ac)
        self.posterior_variance = _f(pv)
        self.posterior_log_variance_clipped = _f(np.log(np.maximum(pv, 1e-20)))
        self.posterior_mean_coef1 = _f(betas * np.sqrt(ac_prev) / (1.0 - ac))
        self.posterior_mean_coef2 = _f((1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac))
        # spec_min/max: [1,1,M] (F == 1) or [1,F,1,M]
        smin = _f(spec_min)[None, None, ...][..., :out_dims]
        smax = _f(spec_max)[None, None, ...][..., :out_dims]
        self.spec_min = np.swapaxes(smin, -3, -2)
        self.spec_max = np.swapaxes(smax, -3, -2)

    # -- q / p ---------------------------------------------------------------
    def q_sample(self, x_start, t, noise):
        return (self.sqrt_alphas_cumprod[t] * x_start
                + self.sqrt_one_minus_alphas_cumprod[t] * noise).astype(F32)

    def p_sample(self, x, t, cond, noise):
        """ddpm.py:123-156 with the fresh randn replaced by the injected `noise`."""
        bsz = x.shape[0]
        tt = np.full((bsz,), t, dtype=np.int64)
        eps = self.denoise_fn(x, tt, cond)
        x_recon = (self.sqrt_recip_alphas_cumprod[t] * x
                   - self.sqrt_recipm1_alphas_cumprod[t] * eps).astype(F32)
        mean = (self.posterior_mean_coef1[t] * x_recon + self.posterior_mean_coef2[t] * x).astype(F32)
        nonzero = F32(0.0 if t == 0 else 1.0)
        sigma = F32(np.exp(F32(0.5) * self.posterior_log_variance_clipped[t]))
        return (mean + F32(nonzero * sigma) * noise).astype(F32)

    def p_sample_ddim(self, x, t, interval, cond):
        bsz = x.shape[0]
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[max(t - interval, 0)]
        eps = self.denoise_fn(x, np.full((bsz,), t, dtype=np.int64), cond)
        c_eps = F32(np.sqrt(F32(F32(1) - a_prev) / a_prev) - np.sqrt(F32(F32(1) - a_t) / a_t))
        return (np.sqrt(a_prev) * (x / np.sqrt(a_t) + c_eps * eps)).astype(F32)

    def _plms_x_pred(self, x, noise_t, t, interval):
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[max(t - interval, 0)]
        a_t_sq, a_prev_sq = F32(np.sqrt(a_t)), F32(np.sqrt(a_prev))
        c_x = F32(F32(1) / F32(a_t_sq * F32(a_t_sq + a_prev_sq)))
        c_n = F32(F32(1) / F32(a_t_sq * F32(np.sqrt(F32(F32(1) - a_prev) * a_t)
                                             + np.sqrt(F32(F32(1) - a_t) * a_prev))))
        x_delta = (F32(a_prev - a_t) * (c_x * x - c_n * noise_t)).astype(F32)
        return (x + x_delta).astype(F32)

    def p_sample_plms(self, x, t, interval, cond, noise_list):
        bsz = x.shape[0]
        eps = self.denoise_fn(x, np.full((bsz,), t, dtype=np.int64), cond)
        n = len(noise_list)
        if n == 0:
            x_pred = self._plms_x_pred(x, eps, t, interval)
            eps_prev = self.denoise_fn(x_pred, np.full((bsz,), max(t - interval, 0), dtype=np.int64), cond)
            prime = ((eps + eps_prev) / F32(2)).astype(F32)
        elif n == 1:
            prime = ((F32(3) * eps - noise_list[-1]) / F32(2)).astype(F32)
        elif n == 2:
            prime = ((F32(23) * eps - F32(16) * noise_list[-1] + F32(5) * noise_list[-2]) / F32(12)).astype(F32)
        else:
            prime = ((F32(55) * eps - F32(59) * noise_list[-1] + F32(37) * noise_list[-2]
                      - F32(9) * noise_list[-3]) / F32(24)).astype(F32)
        x_prev = self._plms_x_pred(x, prime, t, interval)
        noise_list.append(eps)
        return x_prev

    # -- the loop (ddpm.py:221-351) -------------------------------------------
    def inference(self, cond, noise, x_start=None, *, K_step_infer=None, diff_speedup=1,
                  diff_accelerator="ddim", step_noise=None, trace=None):
        """cond [B,H,T]; noise = x_T [B,F,M,T]; step_noise: list of [B,F,M,T] for ancestral DDPM,
        consumed in loop order.  Returns [B,T,M] (F == 1) or [B,F,T,M]."""
        bsz = noise.shape[0]
        depth = self.k_step if K_step_infer is None else K_step_infer
        speedup = diff_speedup
        if speedup > 0:
            assert depth % speedup == 0, f"Acceleration ratio must be a factor of diffusion depth {depth}."
        noise = _f(noise)
        t_max = min(depth, self.k_step) if self.use_shallow_diffusion else self.k_step
        if t_max >= self.timesteps:
            x = noise
        elif t_max > 0:
            assert x_start is not None, "Missing shallow diffusion source."
            x = self.q_sample(_f(x_start), t_max - 1, noise)
        else:
            assert x_start is not None, "Missing shallow diffusion source."
            x = _f(x_start)

        if speedup > 1 and t_max > 0:
            algorithm = diff_accelerator
            if algorithm == "dpm-solver":
                ns = NoiseScheduleVP(self.betas[:t_max], clip=True)
                x = dpm_solver_pp_2m(self.denoise_fn, x, cond, ns, t_max // speedup, trace=trace)
            elif algorithm == "unipc":
                ns = NoiseScheduleVP(self.betas[:t_max], clip=False)
                x = unipc_bh2(self.denoise_fn, x, cond, ns, t_max // speedup, trace=trace)
            elif algorithm == "pndm":
                noise_list = deque(maxlen=4)
                for i in reversed(range(0, t_max, speedup)):
                    x = self.p_sample_plms(x, i, speedup, cond, noise_list)
                    if trace is not None:
                        trace.append(x.copy())
            elif algorithm == "ddim":
                for i in reversed(range(0, t_max, speedup)):
                    x = self.p_sample_ddim(x, i, speedup, cond)
                    if trace is not None:
                        trace.append(x.copy())
            else:
                raise ValueError(f"Unsupported acceleration algorithm for DDPM: {algorithm}.")
        else:
            it = iter(step_noise if step_noise is not None else [])
            for i in reversed(range(0, t_max)):
                x = self.p_sample(x, i, cond, _f(next(it)))
                if trace is not None:
                    trace.append(x.copy())
        x = np.swapaxes(x, 2, 3)
        if x.shape[1] == 1:
            x = x[:, 0]
        return np.ascontiguousarray(x, dtype=F32)

    # -- the ONNX deployment twin's runtime inputs (deployment/modules/diffusion.py:93-161) ------
    def onnx_plan(self, steps, depth=None):
        """(t_max, speedup) of GaussianDiffusionONNX.forward(condition, x_start, depth, steps):
        without a shallow source the speed-up is timesteps // steps snapped DOWN to a factor of timesteps and the
        loop covers [0, k_step) (:112-115); with one, depth * timesteps is rounded, capped at k_step, the speed-up
        is depth // steps (not snapped) and the depth is rounded down to a multiple of it (:117-120)."""
        if depth is None:
            speedup = max(1, self.timesteps // steps)
            factors = [i for i in range(1, self.timesteps + 1) if self.timesteps % i == 0]
            speedup = [f for f in factors if f <= speedup][-1]
            return self.k_step, speedup
        depth_i = min(int(np.round(F32(depth) * F32(self.timesteps))), self.k_step)     # torch.round: half to even, as numpy
        speedup = max(1, depth_i // steps)
        return depth_i // speedup * speedup, speedup

    def forward_onnx(self, condition, noise, x_start=None, depth=None, steps=10, step_noise=None):
        """GaussianDiffusionONNX.forward (deployment/modules/diffusion.py:105-161): DDIM when the speed-up exceeds 1,
        ancestral sampling otherwise; norm / denorm in the (x - b) / k form (:93-103).  condition [B,T,H]."""
        cond = np.ascontiguousarray(np.swapaxes(_f(condition), 1, 2))
        noise = _f(noise)
        k = ((self.spec_max - self.spec_min) / F32(2)).astype(F32)
        b = ((self.spec_max + self.spec_min) / F32(2)).astype(F32)
        t_max, speedup = self.onnx_plan(steps, None if x_start is None else depth)
        if x_start is None:
            x = noise
        else:
            xs = np.swapaxes(((_f(x_start) - b) / k).astype(F32), -2, -1)
            if self.num_feats == 1:
                xs = xs[:, None]
            if t_max >= self.timesteps:
                x = noise
            elif t_max > 0:
                x = self.q_sample(xs, t_max - 1, noise)
            else:
                x = xs
        if speedup > 1:
            for i in reversed(range(0, t_max, speedup)):
                x = self.p_sample_ddim(x, i, speedup, cond)
        else:
            it = iter(step_noise if step_noise is not None else [])
            for i in reversed(range(0, t_max)):
                x = self.p_sample(x, i, cond, _f(next(it)))
        x = np.swapaxes(x, 2, 3)
        if x.shape[1] == 1:
            x = x[:, 0]
        return (np.ascontiguousarray(x, dtype=F32) * k + b).astype(F32)

    def forward(self, condition, noise, src_spec=None, **kw):
        """GaussianDiffusion.forward(infer=True) (ddpm.py:353-377). condition [B,T,H]."""
        cond = np.ascontiguousarray(np.swapaxes(_f(condition), 1, 2))
        spec = None
        if src_spec is not None:
            spec = np.swapaxes(self.norm_spec(src_spec), -2, -1)
            if self.num_feats == 1:
                spec = spec[:, None]
        x = self.inference(cond, noise, x_start=spec, **kw)
        return self.denorm_spec(x)

    def norm_spec(self, x):
        return ((_f(x) - self.spec_min) / (self.spec_max - self.spec_min) * F32(2) - F32(1)).astype(F32)

    def denorm_spec(self, x):
        return ((x + F32(1)) / F32(2) * (self.spec_max - self.spec_min) + self.spec_min).astype(F32)


# ----------------------------------------------------------------------------
# RectifiedFlow (reflow.py:13-144)
# ----------------------------------------------------------------------------
class RectifiedFlow:
    def __init__(self, velocity_fn, out_dims, num_feats=1, t_start=0.0, time_scale_factor=1000,
                 spec_min=None, spec_max=None, use_shallow_diffusion=False):
        self.velocity_fn = velocity_fn
        self.out_dims, self.num_feats = out_dims, num_feats
        self.use_shallow_diffusion = use_shallow_diffusion
        if use_shallow_diffusion:
            assert 0.0 <= t_start <= 1.0, "T_start should be in [0, 1]."
        else:
            t_start = 0.0
        self.t_start = t_start
        self.time_scale_factor = time_scale_factor
        smin = _f(spec_min)[None, None, ...][..., :out_dims]
        smax = _f(spec_max)[None, None, ...][..., :out_dims]
        self.spec_min = np.swapaxes(smin, -3, -2)
        self.spec_max = np.swapaxes(smax, -3, -2)

    def _v(self, x, t, cond):
        # velocity_fn(x, time_scale_factor * t, cond) with t a [1] float32 array
        return self.velocity_fn(x, (F32(self.time_scale_factor) * t).astype(F32), cond)

    def sample_euler(self, x, t, dt, cond):
        return (x + self._v(x, t, cond) * F32(dt)).astype(F32)

    def sample_rk2(self, x, t, dt, cond):
        k1 = self._v(x, t, cond)
        k2 = self._v((x + F32(0.5) * k1 * F32(dt)).astype(F32), (t + F32(0.5 * dt)).astype(F32), cond)
        return (x + k2 * F32(dt)).astype(F32)

    def sample_rk4(self, x, t, dt, cond):
        k1 = self._v(x, t, cond)
        k2 = self._v((x + F32(0.5) * k1 * F32(dt)).astype(F32), (t + F32(0.5 * dt)).astype(F32), cond)
        k3 = self._v((x + F32(0.5) * k2 * F32(dt)).astype(F32), (t + F32(0.5 * dt)).astype(F32), cond)
        k4 = self._v((x + k3 * F32(dt)).astype(F32), (t + F32(dt)).astype(F32), cond)
        return (x + (k1 + F32(2) * k2 + F32(2) * k3 + k4) * F32(dt) / F32(6)).astype(F32)

    def sample_rk5(self, x, t, dt, cond):
        fdt = F32(dt)
        k1 = self._v(x, t, cond)
        k2 = self._v((x + F32(0.25) * k1 * fdt).astype(F32), (t + F32(0.25 * dt)).astype(F32), cond)
        k3 = self._v((x + F32(0.125) * (k2 + k1) * fdt).astype(F32), (t + F32(0.25 * dt)).astype(F32), cond)
        k4 = self._v((x + F32(0.5) * (-k2 + F32(2) * k3) * fdt).astype(F32), (t + F32(0.5 * dt)).astype(F32), cond)
        k5 = self._v((x + F32(0.0625) * (F32(3) * k1 + F32(9) * k4) * fdt).astype(F32),
                     (t + F32(0.75 * dt)).astype(F32), cond)
        k6 = self._v((x + (F32(-3) * k1 + F32(2) * k2 + F32(12) * k3 - F32(12) * k4 + F32(8) * k5) * fdt / F32(7)).astype(F32),
                     (t + F32(dt)).astype(F32), cond)
        return (x + (F32(7) * k1 + F32(32) * k3 + F32(12) * k4 + F32(32) * k5 + F32(7) * k6) * fdt / F32(90)).astype(F32)

    def inference(self, cond, noise, x_end=None, *, T_start_infer=None, sampling_algorithm="euler",
                  sampling_steps=20, trace=None):
        noise = _f(noise)
        t_start = self.t_start if T_start_infer is None else T_start_infer
        if self.use_shallow_diffusion and t_start > 0:
            assert x_end is not None, "Missing shallow diffusion source."
            if t_start >= 1.0:
                t_start = 1.0
                x = _f(x_end)
            else:
                x = (F32(t_start) * _f(x_end) + F32(1 - t_start) * noise).astype(F32)
        else:
            t_start = 0.0
            x = noise
        if t_start < 1:
            dt = (1.0 - t_start) / max(1, sampling_steps)
            fn = {"euler": self.sample_euler, "rk2": self.sample_rk2,
                  "rk4": self.sample_rk4, "rk5": self.sample_rk5}.get(sampling_algorithm)
            if fn is None:
                raise ValueError(f"Unsupported algorithm for Rectified Flow: {sampling_algorithm}.")
            dts = _f([dt])
            for i in range(sampling_steps):
                t = (F32(t_start) + F32(i) * dts).astype(F32)
                x = fn(x, t, dt, cond)
                if trace is not None:
                    trace.append(x.copy())
        x = np.swapaxes(x, 2, 3)
        if x.shape[1] == 1:
            x = x[:, 0]
        return np.ascontiguousarray(x, dtype=F32)

    def forward(self, condition, noise, src_spec=None, **kw):
        cond = np.ascontiguousarray(np.swapaxes(_f(condition), 1, 2))
        spec = None
        if src_spec is not None:
            spec = np.swapaxes(self.norm_spec(src_spec), -2, -1)
            if self.num_feats == 1:
                spec = spec[:, None]
        x = self.inference(cond, noise, x_end=spec, **kw)
        return self.denorm_spec(x)

    def forward_onnx(self, condition, noise, x_end=None, depth=None, steps=10):
        """RectifiedFlowONNX.forward (deployment/modules/rectified_flow.py:37-68): euler only; t_start =
        max(1 - depth, self.t_start) in fp32; the step times i * dt + t_start are fp32 tensor arithmetic."""
        cond = np.ascontiguousarray(np.swapaxes(_f(condition), 1, 2))
        noise = _f(noise)
        k = ((self.spec_max - self.spec_min) / F32(2)).astype(F32)
        b = ((self.spec_max + self.spec_min) / F32(2)).astype(F32)
        if x_end is None:
            t_start = F32(0.0)
            x = noise
        else:
            t_start = max(F32(F32(1) - F32(depth)), F32(self.t_start))
            xe = np.swapaxes(((_f(x_end) - b) / k).astype(F32), -2, -1)
            if self.num_feats == 1:
                xe = xe[:, None]
            if t_start <= 0.0:
                x = noise
            elif t_start >= 1.0:
                x = xe
            else:
                x = (t_start * xe + F32(F32(1) - t_start) * noise).astype(F32)
        t_width = F32(F32(1) - t_start)
        if t_width >= 0.0:
            dt = F32(t_width / F32(max(1, steps)))
            for i in range(steps):
                t = _f([F32(F32(i) * dt) + t_start])
                x = (x + self._v(x, t, cond) * dt).astype(F32)
        x = np.swapaxes(x, 2, 3)
        if x.shape[1] == 1:
            x = x[:, 0]
        return (np.ascontiguousarray(x, dtype=F32) * k + b).astype(F32)

    norm_spec = GaussianDiffusion.norm_spec
    denorm_spec = GaussianDiffusion.denorm_spec


# ----------------------------------------------------------------------------
# repeat-bin wrappers (ddpm.py:386-505, reflow.py:147-261): the norm/denorm maps only
# ----------------------------------------------------------------------------
def repetitive_spec_ranges(vmin, vmax):
    """spec_min/spec_max lists as RepetitiveDiffusion.__init__ builds them (ddpm.py:392-394)."""
    if isinstance(vmin, (int, float)):
        return 1, [vmin], [vmax]
    return len(vmin), [[v] for v in vmin], [[v] for v in vmax]


def repetitive_norm(diff, x, repeat_bins):
    """RepetitiveDiffusion.norm_spec: [B,T] or [B,F,T] -> [B,T,R] or [B,F,T,R]."""
    x = _f(x)
    rep = np.repeat(x[..., None], repeat_bins, axis=-1)
    return GaussianDiffusion.norm_spec(diff, rep)


def repetitive_denorm(diff, x):
    return GaussianDiffusion.denorm_spec(diff, x).mean(axis=-1, dtype=F32).astype(F32)


def pitch_norm(diff, x, repeat_bins, cmin, cmax):
    return repetitive_norm(diff, np.clip(_f(x), F32(cmin), F32(cmax)), repeat_bins)


def pitch_denorm(diff, x, cmin, cmax):
    return np.clip(repetitive_denorm(diff, x), F32(cmin), F32(cmax)).astype(F32)


def _clamp_list(xs, clamps):
    out = []
    for x, c in zip(xs, clamps):
        if c is None:
            out.append(_f(x))
        else:
            lo = None if c[0] is None else F32(c[0])
            hi = None if c[1] is None else F32(c[1])
            out.append(np.clip(_f(x), lo, hi).astype(F32))
    return out


def multivar_norm(diff, xs, repeat_bins, clamps):
    xs = np.stack(_clamp_list(xs, clamps), axis=1)
    if diff.num_feats == 1:
        xs = xs[:, 0]
    return repetitive_norm(diff, xs, repeat_bins)


def multivar_denorm(diff, x, clamps):
    xs = repetitive_denorm(diff, x)
    xs = [xs] if diff.num_feats == 1 else [xs[:, i] for i in range(diff.num_feats)]
    return _clamp_list(xs, clamps)
