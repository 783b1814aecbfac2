"""GaussianDiffusion / RectifiedFlow (+ repeat-bin variants) over libdsdenoise: the coarse drop-in boundary.

Same class names, constructor arguments, attribute names (`denoise_fn` / `velocity_fn`: they are part of
the checkpoint key layout), buffers, hparams keys, assertions and exceptions as
`modules/core/ddpm.py:55-505` and `modules/core/reflow.py:13-261`.  `forward(condition, gt_spec=None,
src_spec=None, infer=True)` has the reference signature; the sampling loop itself is ONE call into the
library (dsd_sample): the host computes the solver's scalar coefficients (schedule.py), the device runs
every NFE plus the fused solver update, by default replayed from a cached hipGraph.

Inference only: `infer=False` (p_losses, ddpm.py:360-367) raises - training stays on the reference.
Randomness: x_T (and the per-step noise of ancestral DDPM) is drawn with `torch.randn` exactly where the
reference draws it, or injected through the keyword-only `noise=` / `step_noise=` arguments (tests).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import torch
from torch import nn

from . import _lib, schedule
from .backbones import build_backbone
from .hparams import hparams


def _spec_buffer(spec, out_dims):
    # ddpm.py:104-109 / reflow.py:29-34: [1,1,M] (F == 1) or [1,F,1,1]
    return torch.FloatTensor(spec)[None, None, :out_dims].transpose(-3, -2)


class _SamplerMixin:
    """Runs a schedule.Program on the backbone's native handle."""

    # hipGraph policy of the loop: "lazy" (default) = a (program, batch shape) runs eagerly the first time and is captured
    # when it comes back (a project's segments all differ in length, and a capture costs about one loop); True = capture
    # at first use (fixed shapes: benchmarks, servers with bucketed lengths); False = never
    use_graph = "lazy"
    _ANCESTRAL_CHUNK = 50       # ancestral DDPM: steps per dsd_sample call (bounds the injected-noise tensor)

    def _backbone(self):
        return self.denoise_fn if hasattr(self, "denoise_fn") else self.velocity_fn

    def _prog_cache(self):
        if not hasattr(self, "_programs"):
            object.__setattr__(self, "_programs", {})
        return self._programs

    def _cached_program(self, key, builder):
        cache = self._prog_cache()
        if key not in cache:
            if len(cache) > 32:
                cache.clear()
            prog = builder()
            cache[key] = (prog,) + _lib.program_to_c(prog)
        return cache[key]

    def _run_program(self, entry, cond, x_init, noise=None, out=None, transpose=True, scale=None, shift=None):
        prog, cprog, _keep = entry
        net = self._backbone()
        x_init = x_init.to(torch.float32).contiguous()
        b, f, m, t = x_init.shape
        if b == 0 or t == 0:
            # empty batch / zero frames: the reference's loop runs on empty tensors and returns an empty mel; there
            # is nothing to launch (still a device tensor: a CPU input raises in prepare_cond as everywhere else)
            if x_init.device.type != "cuda":
                net.native_handle(x_init.device)
            if out is not None:
                return out
            shape = ((b, t, m) if f == 1 else (b, f, t, m)) if transpose else (b, f, m, t)
            return torch.empty(shape, device=x_init.device, dtype=torch.float32)
        handle = net.prepare_cond(cond)
        if out is None:
            if transpose:
                out = torch.empty((b, t, m) if f == 1 else (b, f, t, m), device=x_init.device, dtype=torch.float32)
            else:
                out = torch.empty_like(x_init)
        flags = (_lib.DSD_SAMPLE_GRAPH if self.use_graph else 0) | (_lib.DSD_SAMPLE_TRANSPOSE if transpose else 0)
        if self.use_graph == "lazy":
            flags |= _lib.DSD_SAMPLE_GRAPH_LAZY
        nptr = None
        if prog.n_noise:
            noise = noise.to(torch.float32).contiguous()
            assert noise.shape[0] >= prog.n_noise
            nptr = C.c_void_p(noise.data_ptr())
        stream = torch.cuda.current_stream(x_init.device).cuda_stream
        _lib.check(handle, _lib.lib().dsd_sample(
            handle, C.byref(cprog), C.c_void_p(x_init.data_ptr()), nptr, C.c_void_p(out.data_ptr()),
            None if scale is None else C.c_void_p(scale.data_ptr()),
            None if shift is None else C.c_void_p(shift.data_ptr()), flags, C.c_void_p(stream)), "dsd_sample")
        return out


class GaussianDiffusion(nn.Module, _SamplerMixin):
    def __init__(self, out_dims, num_feats=1, timesteps=1000, k_step=1000,
                 backbone_type=None, backbone_args=None, betas=None,
                 spec_min=None, spec_max=None):
        super().__init__()
        self.denoise_fn: nn.Module = build_backbone(out_dims, num_feats, backbone_type, backbone_args)
        self.out_dims = out_dims
        self.num_feats = num_feats
        if betas is not None:
            betas = betas.detach().cpu().numpy() if isinstance(betas, torch.Tensor) else betas
        else:
            betas = schedule.BETA_SCHEDULE[hparams['schedule_type']](timesteps)
        self.use_shallow_diffusion = hparams.get('use_shallow_diffusion', False)
        if self.use_shallow_diffusion:
            assert k_step <= timesteps, 'K_step should not be larger than timesteps.'
        self.timesteps = timesteps
        self.k_step = k_step if self.use_shallow_diffusion else timesteps
        self._tables = schedule.DDPMTables(betas)
        for name in schedule.DDPMTables.NAMES:
            self.register_buffer(name, torch.from_numpy(getattr(self._tables, name).copy()))
        self.register_buffer('spec_min', _spec_buffer(spec_min, out_dims))
        self.register_buffer('spec_max', _spec_buffer(spec_max, out_dims))
        # for compatibility with ONNX continuous acceleration
        self.time_scale_factor = self.timesteps
        self.t_start = 1 - self.k_step / self.timesteps
        factors = torch.LongTensor([i for i in range(1, self.timesteps + 1) if self.timesteps % i == 0])
        self.register_buffer('timestep_factors', factors, persistent=False)

    # ---- closed forms kept for API parity (ddpm.py:117-135, 206-210) ---------------------------
    def q_sample(self, x_start, t, noise):
        shp = (-1,) + (1,) * (x_start.dim() - 1)
        return (self.sqrt_alphas_cumprod[t].reshape(shp) * x_start
                + self.sqrt_one_minus_alphas_cumprod[t].reshape(shp) * noise)

    def predict_start_from_noise(self, x_t, t, noise):
        shp = (-1,) + (1,) * (x_t.dim() - 1)
        return (self.sqrt_recip_alphas_cumprod[t].reshape(shp) * x_t
                - self.sqrt_recipm1_alphas_cumprod[t].reshape(shp) * noise)

    def p_losses(self, *a, **k):
        raise NotImplementedError("diffsinger_amd is inference-only; train with the reference GaussianDiffusion")

    # ---- the loop (ddpm.py:221-351) -----------------------------------------------------------
    @torch.no_grad()
    def inference(self, cond, b=1, x_start=None, device=None, *, noise=None, step_noise=None, _denorm=False):
        depth = hparams.get('K_step_infer', self.k_step)
        speedup = hparams['diff_speedup']
        if speedup > 0:
            assert depth % speedup == 0, f'Acceleration ratio must be a factor of diffusion depth {depth}.'
        if noise is None:
            noise = torch.randn(b, self.num_feats, self.out_dims, cond.shape[2], device=device)
        if self.use_shallow_diffusion:
            t_max = min(depth, self.k_step)
        else:
            t_max = self.k_step

        if t_max >= self.timesteps:
            x = noise
        elif t_max > 0:
            assert x_start is not None, 'Missing shallow diffusion source.'
            x = self.q_sample(x_start, torch.full((b,), t_max - 1, device=device, dtype=torch.long), noise)
        else:
            assert x_start is not None, 'Missing shallow diffusion source.'
            x = x_start

        scale = shift = None
        if _denorm:   # denorm_spec folded into the unpack: (x+1)/2*(max-min)+min = x*scale + shift
            rng = (self.spec_max - self.spec_min).reshape(-1)
            scale = self._expand_fm(rng / 2)
            shift = self._expand_fm(rng / 2 + self.spec_min.reshape(-1))
        tb = self._tables
        if speedup > 1 and t_max > 0:
            algorithm = hparams['diff_accelerator']
            if algorithm == 'dpm-solver':
                entry = self._cached_program(('dpm', t_max, speedup), lambda: schedule.dpm_solver_pp_program(
                    self.betas[:t_max].cpu(), t_max // speedup))
            elif algorithm == 'unipc':
                entry = self._cached_program(('unipc', t_max, speedup), lambda: schedule.unipc_program(
                    self.betas[:t_max].cpu(), t_max // speedup))
            elif algorithm == 'pndm':
                if b > 1:
                    # the reference evaluates `max(t - interval, 0)` on a [B] tensor (ddpm.py:192): B > 1 raises
                    raise RuntimeError("Boolean value of Tensor with more than one value is ambiguous "
                                       "(PNDM supports B == 1 only, as in the reference, ddpm.py:192)")
                entry = self._cached_program(('pndm', t_max, speedup), lambda: schedule.plms_program(tb, t_max, speedup))
            elif algorithm == 'ddim':
                entry = self._cached_program(('ddim', t_max, speedup), lambda: schedule.ddim_program(tb, t_max, speedup))
            else:
                raise ValueError(f"Unsupported acceleration algorithm for DDPM: {algorithm}.")
            return self._run_program(entry, cond, x, scale=scale, shift=shift)
        # ancestral sampling, one fresh randn per step (ddpm.py:149-156,347-349), run in chunks
        if t_max == 0:
            entry = self._cached_program(('noop',), lambda: schedule.Program(1, 0, []))
            return self._run_program(entry, cond, x, scale=scale, shift=shift)
        hi, k = t_max, 0
        while hi > 0:
            lo = max(hi - self._ANCESTRAL_CHUNK, 0)
            n = hi - lo
            if step_noise is not None:
                chunk = step_noise[k:k + n]
            else:
                chunk = torch.stack([torch.randn(x.shape, device=device) for _ in range(n)])
            k += n
            entry = self._cached_program(('ddpm', hi, lo), lambda: schedule.ddpm_ancestral_program(tb, hi, lo))
            last = lo == 0
            x = self._run_program(entry, cond, x, noise=chunk, transpose=last,
                                  scale=scale if last else None, shift=shift if last else None)
            hi = lo
        return x

    def _expand_fm(self, v):
        # per-(f, m) vector of length F*M from a spec_min/max-shaped tensor ([1,1,M] or [1,F,1,1])
        f, m = self.num_feats, self.out_dims
        v = v.reshape(-1)
        if v.numel() == 1:
            return v.expand(f * m).contiguous()
        if f == 1:
            return v.expand(m).contiguous() if v.numel() == 1 else v.contiguous()
        return v.reshape(f, 1).expand(f, m).reshape(-1).contiguous()

    def forward(self, condition, gt_spec=None, src_spec=None, infer=True, *, noise=None, step_noise=None, lengths=None):
        """
            conditioning diffusion, use fastspeech2 encoder output as the condition
            `lengths` [B]: ragged batch - item b is run as if alone at T = lengths[b] (dsd_set_lengths); frames beyond
            an item's length are unspecified in the result.
        """
        cond = condition.transpose(1, 2)
        b, device = condition.shape[0], condition.device
        if not infer:
            raise NotImplementedError(
                "diffsinger_amd.GaussianDiffusion is inference-only (infer=True); train with the reference module")
        if src_spec is not None:
            spec = self.norm_spec(src_spec).transpose(-2, -1)
            if self.num_feats == 1:
                spec = spec[:, None, :, :]
        else:
            spec = None
        if lengths is not None:
            self.denoise_fn.set_lengths(lengths, device)
        try:
            x = self.inference(cond, b=b, x_start=spec, device=device, noise=noise, step_noise=step_noise, _denorm=True)
        finally:
            if lengths is not None:
                self.denoise_fn.set_lengths(None, device)
        return self._finish_denorm(x)

    def norm_spec(self, x):
        return (x - self.spec_min) / (self.spec_max - self.spec_min) * 2 - 1

    def denorm_spec(self, x):
        return (x + 1) / 2 * (self.spec_max - self.spec_min) + self.spec_min

    def _finish_denorm(self, x):
        return x


class RepetitiveDiffusion(GaussianDiffusion):
    def __init__(self, vmin: float | int | list, vmax: float | int | list,
                 repeat_bins: int, timesteps=1000, k_step=1000,
                 backbone_type=None, backbone_args=None,
                 betas=None):
        assert (isinstance(vmin, (float, int)) and isinstance(vmin, (float, int))) or len(vmin) == len(vmax)
        num_feats = 1 if isinstance(vmin, (float, int)) else len(vmin)
        spec_min = [vmin] if num_feats == 1 else [[v] for v in vmin]
        spec_max = [vmax] if num_feats == 1 else [[v] for v in vmax]
        self.repeat_bins = repeat_bins
        super().__init__(
            out_dims=repeat_bins, num_feats=num_feats,
            timesteps=timesteps, k_step=k_step,
            backbone_type=backbone_type, backbone_args=backbone_args,
            betas=betas, spec_min=spec_min, spec_max=spec_max
        )

    def norm_spec(self, x):
        """[B, T] or [B, F, T] -> [B, T, R] or [B, F, T, R]"""
        repeats = [1, 1, self.repeat_bins] if self.num_feats == 1 else [1, 1, 1, self.repeat_bins]
        return super().norm_spec(x.unsqueeze(-1).repeat(repeats))

    def denorm_spec(self, x):
        """[B, T, R] or [B, F, T, R] -> [B, T] or [B, F, T]"""
        return super().denorm_spec(x).mean(dim=-1)

    def _finish_denorm(self, x):
        return x.mean(dim=-1)


class PitchDiffusion(RepetitiveDiffusion):
    def __init__(self, vmin: float, vmax: float,
                 cmin: float, cmax: float, repeat_bins,
                 timesteps=1000, k_step=1000,
                 backbone_type=None, backbone_args=None,
                 betas=None):
        self.vmin = vmin  # norm min
        self.vmax = vmax  # norm max
        self.cmin = cmin  # clip min
        self.cmax = cmax  # clip max
        super().__init__(
            vmin=vmin, vmax=vmax, repeat_bins=repeat_bins,
            timesteps=timesteps, k_step=k_step,
            backbone_type=backbone_type, backbone_args=backbone_args,
            betas=betas
        )

    def norm_spec(self, x):
        return super().norm_spec(x.clamp(min=self.cmin, max=self.cmax))

    def denorm_spec(self, x):
        return super().denorm_spec(x).clamp(min=self.cmin, max=self.cmax)

    def _finish_denorm(self, x):
        return super()._finish_denorm(x).clamp(min=self.cmin, max=self.cmax)


def _clamp_list(xs, clamps):
    out = []
    for x, c in zip(xs, clamps):
        out.append(x if c is None else x.clamp(min=c[0], max=c[1]))
    return out


class MultiVarianceDiffusion(RepetitiveDiffusion):
    def __init__(
            self, ranges: List[Tuple[float, float]],
            clamps: List[Tuple[float | None, float | None] | None],
            repeat_bins, timesteps=1000, k_step=1000,
            backbone_type=None, backbone_args=None,
            betas=None
    ):
        assert len(ranges) == len(clamps)
        self.clamps = clamps
        vmin = [r[0] for r in ranges]
        vmax = [r[1] for r in ranges]
        if len(vmin) == 1:
            vmin = vmin[0]
        if len(vmax) == 1:
            vmax = vmax[0]
        super().__init__(
            vmin=vmin, vmax=vmax, repeat_bins=repeat_bins,
            timesteps=timesteps, k_step=k_step,
            backbone_type=backbone_type, backbone_args=backbone_args,
            betas=betas
        )

    def clamp_spec(self, xs: list | tuple):
        return _clamp_list(xs, self.clamps)

    def norm_spec(self, xs: list | tuple):
        """sequence of [B, T] -> [B, F, T] -> [B, F, T, R]"""
        assert len(xs) == self.num_feats
        xs = torch.stack(self.clamp_spec(xs), dim=1)
        if self.num_feats == 1:
            xs = xs.squeeze(1)
        return super().norm_spec(xs)

    def _split(self, xs):
        xs = [xs] if self.num_feats == 1 else xs.unbind(dim=1)
        assert len(xs) == self.num_feats
        return self.clamp_spec(xs)

    def denorm_spec(self, xs):
        """[B, T, R] or [B, F, T, R] -> sequence of [B, T]"""
        return self._split(super().denorm_spec(xs))

    def _finish_denorm(self, x):
        return self._split(x.mean(dim=-1))


# ==============================================================================================
# Rectified flow (reflow.py)
# ==============================================================================================
class RectifiedFlow(nn.Module, _SamplerMixin):
    def __init__(self, out_dims, num_feats=1, t_start=0., time_scale_factor=1000,
                 backbone_type=None, backbone_args=None,
                 spec_min=None, spec_max=None):
        super().__init__()
        self.velocity_fn: nn.Module = build_backbone(out_dims, num_feats, backbone_type, backbone_args)
        self.out_dims = out_dims
        self.num_feats = num_feats
        self.use_shallow_diffusion = hparams.get('use_shallow_diffusion', False)
        if self.use_shallow_diffusion:
            assert 0. <= t_start <= 1., 'T_start should be in [0, 1].'
        else:
            t_start = 0.
        self.t_start = t_start
        self.time_scale_factor = time_scale_factor
        self.register_buffer('spec_min', _spec_buffer(spec_min, out_dims), persistent=False)
        self.register_buffer('spec_max', _spec_buffer(spec_max, out_dims), persistent=False)

    def p_losses(self, *a, **k):
        raise NotImplementedError("diffsinger_amd is inference-only; train with the reference RectifiedFlow")

    @torch.no_grad()
    def inference(self, cond, b=1, x_end=None, device=None, *, noise=None, _denorm=False):
        if noise is None:
            noise = torch.randn(b, self.num_feats, self.out_dims, cond.shape[2], device=device)
        t_start = hparams.get('T_start_infer', self.t_start)
        if self.use_shallow_diffusion and t_start > 0:
            assert x_end is not None, 'Missing shallow diffusion source.'
            if t_start >= 1.:
                t_start = 1.
                x = x_end
            else:
                x = t_start * x_end + (1 - t_start) * noise
        else:
            t_start = 0.
            x = noise

        algorithm = hparams['sampling_algorithm']
        infer_step = hparams['sampling_steps']
        scale = shift = None
        if _denorm:
            rng = (self.spec_max - self.spec_min).reshape(-1)
            scale = GaussianDiffusion._expand_fm(self, rng / 2)
            shift = GaussianDiffusion._expand_fm(self, rng / 2 + self.spec_min.reshape(-1))
        if t_start < 1:
            entry = self._cached_program(
                ('reflow', algorithm, infer_step, float(t_start), float(self.time_scale_factor)),
                lambda: schedule.reflow_program(algorithm, infer_step, t_start, self.time_scale_factor))
        else:
            entry = self._cached_program(('noop',), lambda: schedule.Program(1, 0, []))
        return self._run_program(entry, cond, x, scale=scale, shift=shift)

    def forward(self, condition, gt_spec=None, src_spec=None, infer=True, *, noise=None, lengths=None):
        """`lengths` [B]: ragged batch - item b is run as if alone at T = lengths[b] (dsd_set_lengths)."""
        cond = condition.transpose(1, 2)
        b, device = condition.shape[0], condition.device
        if not infer:
            raise NotImplementedError(
                "diffsinger_amd.RectifiedFlow is inference-only (infer=True); train with the reference module")
        if src_spec is not None:
            spec = self.norm_spec(src_spec).transpose(-2, -1)
            if self.num_feats == 1:
                spec = spec[:, None, :, :]
        else:
            spec = None
        if lengths is not None:
            self.velocity_fn.set_lengths(lengths, device)
        try:
            x = self.inference(cond, b=b, x_end=spec, device=device, noise=noise, _denorm=True)
        finally:
            if lengths is not None:
                self.velocity_fn.set_lengths(None, device)
        return self._finish_denorm(x)

    def norm_spec(self, x):
        return (x - self.spec_min) / (self.spec_max - self.spec_min) * 2 - 1

    def denorm_spec(self, x):
        return (x + 1) / 2 * (self.spec_max - self.spec_min) + self.spec_min

    def _finish_denorm(self, x):
        return x


class RepetitiveRectifiedFlow(RectifiedFlow):
    def __init__(self, vmin: float | int | list, vmax: float | int | list,
                 repeat_bins: int, time_scale_factor=1000,
                 backbone_type=None, backbone_args=None):
        assert (isinstance(vmin, (float, int)) and isinstance(vmin, (float, int))) or len(vmin) == len(vmax)
        num_feats = 1 if isinstance(vmin, (float, int)) else len(vmin)
        spec_min = [vmin] if num_feats == 1 else [[v] for v in vmin]
        spec_max = [vmax] if num_feats == 1 else [[v] for v in vmax]
        self.repeat_bins = repeat_bins
        super().__init__(
            out_dims=repeat_bins, num_feats=num_feats,
            time_scale_factor=time_scale_factor,
            backbone_type=backbone_type, backbone_args=backbone_args,
            spec_min=spec_min, spec_max=spec_max
        )

    def norm_spec(self, x):
        repeats = [1, 1, self.repeat_bins] if self.num_feats == 1 else [1, 1, 1, self.repeat_bins]
        return super().norm_spec(x.unsqueeze(-1).repeat(repeats))

    def denorm_spec(self, x):
        return super().denorm_spec(x).mean(dim=-1)

    def _finish_denorm(self, x):
        return x.mean(dim=-1)


class PitchRectifiedFlow(RepetitiveRectifiedFlow):
    def __init__(self, vmin: float, vmax: float,
                 cmin: float, cmax: float, repeat_bins,
                 time_scale_factor=1000,
                 backbone_type=None, backbone_args=None):
        self.vmin = vmin  # norm min
        self.vmax = vmax  # norm max
        self.cmin = cmin  # clip min
        self.cmax = cmax  # clip max
        super().__init__(
            vmin=vmin, vmax=vmax, repeat_bins=repeat_bins,
            time_scale_factor=time_scale_factor,
            backbone_type=backbone_type, backbone_args=backbone_args
        )

    def norm_spec(self, x):
        return super().norm_spec(x.clamp(min=self.cmin, max=self.cmax))

    def denorm_spec(self, x):
        return super().denorm_spec(x).clamp(min=self.cmin, max=self.cmax)

    def _finish_denorm(self, x):
        return super()._finish_denorm(x).clamp(min=self.cmin, max=self.cmax)


class MultiVarianceRectifiedFlow(RepetitiveRectifiedFlow):
    def __init__(
            self, ranges: List[Tuple[float, float]],
            clamps: List[Tuple[float | None, float | None] | None],
            repeat_bins, time_scale_factor=1000,
            backbone_type=None, backbone_args=None
    ):
        assert len(ranges) == len(clamps)
        self.clamps = clamps
        vmin = [r[0] for r in ranges]
        vmax = [r[1] for r in ranges]
        if len(vmin) == 1:
            vmin = vmin[0]
        if len(vmax) == 1:
            vmax = vmax[0]
        super().__init__(
            vmin=vmin, vmax=vmax, repeat_bins=repeat_bins,
            time_scale_factor=time_scale_factor,
            backbone_type=backbone_type, backbone_args=backbone_args
        )

    def clamp_spec(self, xs: list | tuple):
        return _clamp_list(xs, self.clamps)

    def norm_spec(self, xs: list | tuple):
        assert len(xs) == self.num_feats
        xs = torch.stack(self.clamp_spec(xs), dim=1)
        if self.num_feats == 1:
            xs = xs.squeeze(1)
        return super().norm_spec(xs)

    def _split(self, xs):
        xs = [xs] if self.num_feats == 1 else xs.unbind(dim=1)
        assert len(xs) == self.num_feats
        return self.clamp_spec(xs)

    def denorm_spec(self, xs):
        return self._split(super().denorm_spec(xs))

    def _finish_denorm(self, x):
        return self._split(x.mean(dim=-1))
