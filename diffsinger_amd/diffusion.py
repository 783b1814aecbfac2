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
        tables = schedule.DDPMTables(betas)
        for name in schedule.DDPMTables.NAMES:
            self.register_buffer(name, torch.from_numpy(getattr(tables, name).copy()))
        self.register_buffer('spec_min', _spec_buffer(spec_min, out_dims))
        self.register_buffer('spec_max', _spec_buffer(spec_max, out_dims))
        # for compatibility with ONNX continuous acceleration
        self.time_scale_factor = self.timesteps
        self.t_start = 1 - self.k_step / self.timesteps
        factors = torch.LongTensor([i for i in range(1, self.timesteps + 1) if self.timesteps % i == 0])
        self.register_buffer('timestep_factors', factors, persistent=False)

    @property
    def _tables(self) -> schedule.DDPMTables:
        """Host view of the schedule buffers AS THEY STAND: the buffers are persistent, so `load_state_dict` may replace the
        schedule the constructor derived from hparams - and the reference reads the buffers (ddpm.py:117-167).  Rebuilt
        (and every cached program dropped) whenever a buffer was reassigned or written in place."""
        bufs = [getattr(self, n) for n in schedule.DDPMTables.NAMES]
        key = tuple((b.data_ptr(), b._version) for b in bufs)
        if getattr(self, '_tables_key', None) != key:
            view = schedule.DDPMTables.from_arrays(
                {n: b.detach().cpu().numpy() for n, b in zip(schedule.DDPMTables.NAMES, bufs)})
            object.__setattr__(self, '_tables_view', view)
            object.__setattr__(self, '_tables_key', key)
            self._prog_cache().clear()
        return self._tables_view

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
    def _affine_out(self):
        """denorm_spec folded into the unpack: (x+1)/2*(max-min)+min = x*scale + shift, one value per (feature, bin)."""
        rng = (self.spec_max - self.spec_min).reshape(-1)
        return _expand_fm(self, rng / 2), _expand_fm(self, rng / 2 + self.spec_min.reshape(-1))

    def _start_state(self, t_max, x_start, noise, b, device):
        """x_T, the forward-diffused shallow source q(x_{t_max-1} | x_start), or the source itself (ddpm.py:232-243)."""
        if t_max >= self.timesteps:
            return noise
        assert x_start is not None, 'Missing shallow diffusion source.'
        if t_max > 0:
            return self.q_sample(x_start, torch.full((b,), t_max - 1, device=device, dtype=torch.long), noise)
        return x_start

    def _run_loop(self, cond, x, t_max, speedup, algorithm, b, device, step_noise, scale, shift):
        """The sampler dispatch of ddpm.py:245-351 on (t_max, speedup) that the caller has already decided."""
        tb = self._tables
        if speedup > 1 and t_max > 0:
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

    @torch.no_grad()
    def inference(self, cond, b=1, x_start=None, device=None, *, noise=None, step_noise=None, _denorm=False):
        depth = hparams.get('K_step_infer', self.k_step)
        speedup = hparams['diff_speedup']
        if speedup > 0:
            assert depth % speedup == 0, f'Acceleration ratio must be a factor of diffusion depth {depth}.'
        if noise is None:
            noise = torch.randn(b, self.num_feats, self.out_dims, cond.shape[2], device=device)
        t_max = min(depth, self.k_step) if self.use_shallow_diffusion else self.k_step
        x = self._start_state(t_max, x_start, noise, b, device)
        scale, shift = self._affine_out() if _denorm else (None, None)
        return self._run_loop(cond, x, t_max, speedup, hparams['diff_accelerator'] if speedup > 1 else None,
                              b, device, step_noise, scale, shift)

    def forward(self, condition, gt_spec=None, src_spec=None, infer=True, *, noise=None, step_noise=None, lengths=None):
        """
            conditioning diffusion, use fastspeech2 encoder output as the condition
            `lengths` [B]: ragged batch - item b is run as if alone at T = lengths[b] (dsd_set_lengths); frames beyond
            an item's length are unspecified in the result.
        """
        if not infer:
            raise NotImplementedError(
                "diffsinger_amd.GaussianDiffusion is inference-only (infer=True); train with the reference module")
        cond = condition.transpose(1, 2)
        b, device = condition.shape[0], condition.device
        spec = None if src_spec is None else _to_bfmt(self.norm_spec(src_spec), self.num_feats)
        with _ragged(self.denoise_fn, lengths, device):
            x = self.inference(cond, b=b, x_start=spec, device=device, noise=noise, step_noise=step_noise, _denorm=True)
        return self._finish_denorm(x)

    @torch.no_grad()
    def forward_onnx(self, condition, x_start=None, depth=None, steps: int = 10, *, noise=None, step_noise=None):
        """The runtime inputs of the ONNX deployment twin, `GaussianDiffusionONNX.forward(condition, x_start, depth, steps)`
        (deployment/modules/diffusion.py:105-161): `steps` and `depth` arrive per call, any `steps` is legal - the
        speed-up is snapped to a factor of `timesteps` (no source) or the depth rounded down to a multiple of the
        speed-up (with one) where `inference()` asserts divisibility - DDIM only (ancestral when the speed-up is 1), and
        norm / denorm are the twin's (x - b) / k and x * k + b.  Repeat-bin subclasses return what the twin's
        `denorm_spec` returns (:185-190, :217-222): the mean over the bins, unclamped and not split (`clamp_spec` is a
        separate graph there); their twins do not override `norm_spec`, so an `x_start` is taken as it comes."""
        cond = condition.transpose(1, 2)
        b, device = condition.shape[0], condition.device
        if noise is None:       # the twin is exported for one utterance per call (:109); a batch draws one x_T per item
            noise = torch.randn((b, self.num_feats, self.out_dims, cond.shape[2]), device=device)
        k, mid = (self.spec_max - self.spec_min) / 2., (self.spec_max + self.spec_min) / 2.
        t_max, speedup = schedule.onnx_ddpm_plan(self.timesteps, self.k_step, self.timestep_factors.cpu(), steps,
                                                 None if x_start is None else depth)
        if x_start is None:
            x = noise
        else:
            x = self._start_state(t_max, _to_bfmt((x_start - mid) / k, self.num_feats), noise, b, device)
        x = self._run_loop(cond, x, t_max, speedup, 'ddim', b, device, step_noise,
                           _expand_fm(self, k.reshape(-1)), _expand_fm(self, mid.reshape(-1)))
        return self._reduce_bins(x)

    def norm_spec(self, x):
        return (x - self.spec_min) / (self.spec_max - self.spec_min) * 2 - 1

    def denorm_spec(self, x):
        return (x + 1) / 2 * (self.spec_max - self.spec_min) + self.spec_min

    # hooks of the repeat-bin codecs below; identities for a plain spectrogram
    def _finish_denorm(self, x):
        return x

    def _pre_norm(self, x):
        return x

    def _reduce_bins(self, x):
        return x


def _expand_fm(mod, v):
    """per-(f, m) vector of length F*M from a spec_min/max-shaped tensor ([1,1,M] or [1,F,1,1])"""
    f, m = mod.num_feats, mod.out_dims
    v = v.reshape(-1)
    if v.numel() == 1:
        return v.expand(f * m).contiguous()
    if f == 1:
        return v.contiguous()
    return v.reshape(f, 1).expand(f, m).reshape(-1).contiguous()


def _to_bfmt(spec, num_feats):
    """normalised [B,T,M] / [B,F,T,M] -> the backbone's [B,F,M,T]"""
    spec = spec.transpose(-2, -1)
    return spec[:, None, :, :] if num_feats == 1 else spec


class _ragged:
    """`with _ragged(backbone, lengths, device)`: per-item lengths for the calls inside (dsd_set_lengths), dense after."""

    def __init__(self, net, lengths, device):
        self.net, self.lengths, self.device = net, lengths, device

    def __enter__(self):
        if self.lengths is not None:
            self.net.set_lengths(self.lengths, self.device)

    def __exit__(self, *exc):
        if self.lengths is not None:
            self.net.set_lengths(None, self.device)
        return False


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

    _affine_out = GaussianDiffusion._affine_out

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
        scale, shift = self._affine_out() if _denorm else (None, None)
        if t_start < 1:
            entry = self._cached_program(
                ('reflow', algorithm, infer_step, float(t_start), float(self.time_scale_factor)),
                lambda: schedule.reflow_program(algorithm, infer_step, t_start, self.time_scale_factor))
        else:
            entry = self._cached_program(('noop',), lambda: schedule.Program(1, 0, []))
        return self._run_program(entry, cond, x, scale=scale, shift=shift)

    def forward(self, condition, gt_spec=None, src_spec=None, infer=True, *, noise=None, lengths=None):
        """`lengths` [B]: ragged batch - item b is run as if alone at T = lengths[b] (dsd_set_lengths)."""
        if not infer:
            raise NotImplementedError(
                "diffsinger_amd.RectifiedFlow is inference-only (infer=True); train with the reference module")
        cond = condition.transpose(1, 2)
        b, device = condition.shape[0], condition.device
        spec = None if src_spec is None else _to_bfmt(self.norm_spec(src_spec), self.num_feats)
        with _ragged(self.velocity_fn, lengths, device):
            x = self.inference(cond, b=b, x_end=spec, device=device, noise=noise, _denorm=True)
        return self._finish_denorm(x)

    @torch.no_grad()
    def forward_onnx(self, condition, x_end=None, depth=None, steps: int = 10, *, noise=None):
        """`RectifiedFlowONNX.forward(condition, x_end, depth, steps)` (deployment/modules/rectified_flow.py:37-68): euler
        only, start time max(1 - depth, self.t_start) in fp32, fp32 step times, the twin's (x - b) / k norm and
        x * k + b denorm.  Repeat-bin subclasses return the mean over the bins (the twin's `denorm_spec`)."""
        cond = condition.transpose(1, 2)
        b, device = condition.shape[0], condition.device
        if noise is None:
            noise = torch.randn((b, self.num_feats, self.out_dims, cond.shape[2]), device=device)
        k, mid = (self.spec_max - self.spec_min) / 2., (self.spec_max + self.spec_min) / 2.
        if x_end is None:
            t_start, x = torch.zeros((), dtype=torch.float32), noise
        else:
            t_start = torch.maximum(1 - torch.as_tensor(depth, dtype=torch.float32).cpu(),
                                    torch.tensor(self.t_start, dtype=torch.float32))
            xe = _to_bfmt((x_end - mid) / k, self.num_feats)
            if t_start <= 0.:
                x = noise
            elif t_start >= 1.:
                x = xe
            else:
                ts = float(t_start)         # an fp32 value: the device arithmetic below is the twin's
                x = ts * xe + (1 - ts) * noise
        ts = float(t_start)
        if ts >= 1. or steps < 1:           # dt = 0: the twin still evaluates the backbone `steps` times and adds v * 0
            entry = self._cached_program(('noop',), lambda: schedule.Program(1, 0, []))
        else:
            entry = self._cached_program(('reflow_onnx', steps, ts, float(self.time_scale_factor)),
                                         lambda: schedule.reflow_onnx_program(steps, t_start, self.time_scale_factor))
        x = self._run_program(entry, cond, x, scale=_expand_fm(self, k.reshape(-1)), shift=_expand_fm(self, mid.reshape(-1)))
        return self._reduce_bins(x)

    def norm_spec(self, x):
        return (x - self.spec_min) / (self.spec_max - self.spec_min) * 2 - 1

    def denorm_spec(self, x):
        return (x + 1) / 2 * (self.spec_max - self.spec_min) + self.spec_min

    def _finish_denorm(self, x):
        return x

    def _pre_norm(self, x):
        return x

    def _reduce_bins(self, x):
        return x


# ==============================================================================================
# Repeat-bin value codecs (ddpm.py:386-505, reflow.py:147-261).  The variance model predicts curves (pitch delta, energy,
# breathiness, ...), not spectrograms: a value v per frame is encoded as R identical "bins" so that the spectrogram
# denoiser can be reused, and decoded as the mean over the bins.  The reference writes the three flavours out twice
# (once per wrapper); here each flavour is ONE mixin that sits in front of either wrapper.  A flavour supplies
#   _pre_norm(x)      what happens to the caller's value(s) before the bins are made (clamp; list -> stacked tensor)
#   _post_denorm(v)   what happens to the decoded value(s) (clamp; stacked tensor -> list)
# ==============================================================================================
def _scalar(v) -> bool:
    return isinstance(v, (float, int))


class _RepeatBins:
    def _init_bins(self, vmin, vmax, repeat_bins, **wrapper_args):
        # (the reference's check tests vmin twice - ddpm.py:391, reflow.py:151 - so a scalar vmin with a list vmax passes
        # there and fails later; both are checked here)
        assert (_scalar(vmin) and _scalar(vmax)) or len(vmin) == len(vmax)
        single = _scalar(vmin)
        self.repeat_bins = repeat_bins
        super().__init__(out_dims=repeat_bins, num_feats=1 if single else len(vmin),
                         spec_min=[vmin] if single else [[v] for v in vmin],
                         spec_max=[vmax] if single else [[v] for v in vmax], **wrapper_args)

    def norm_spec(self, x):
        """value(s) [B, T] / [B, F, T] (or what `_pre_norm` accepts) -> [B, T, R] / [B, F, T, R]"""
        v = self._pre_norm(x)
        return super().norm_spec(v.unsqueeze(-1).expand(*v.shape, self.repeat_bins))

    def denorm_spec(self, x):
        """[B, T, R] / [B, F, T, R] -> value(s)"""
        return self._post_denorm(super().denorm_spec(x).mean(dim=-1))

    def _finish_denorm(self, x):        # x arrives denormalised (fused into the sampler's last kernel)
        return self._post_denorm(x.mean(dim=-1))

    def _reduce_bins(self, x):
        return x.mean(dim=-1)

    def _post_denorm(self, v):
        return v


class _ClampedPitch(_RepeatBins):
    """one curve, clipped to [cmin, cmax] on the way in and on the way out"""

    def _init_pitch(self, vmin, vmax, cmin, cmax, repeat_bins, **wrapper_args):
        self.vmin, self.vmax = vmin, vmax       # range of the normalisation
        self.cmin, self.cmax = cmin, cmax       # clipping range
        self._init_bins(vmin, vmax, repeat_bins, **wrapper_args)

    def _pre_norm(self, x):
        return x.clamp(min=self.cmin, max=self.cmax)

    _post_denorm = _pre_norm


class _MultiCurve(_RepeatBins):
    """F curves handed over (and back) as a sequence of [B, T] tensors, each with its own optional clipping range"""

    def _init_curves(self, ranges, clamps, repeat_bins, **wrapper_args):
        assert len(ranges) == len(clamps)
        self.clamps = clamps
        lo, hi = [r[0] for r in ranges], [r[1] for r in ranges]
        self._init_bins(lo[0] if len(lo) == 1 else lo, hi[0] if len(hi) == 1 else hi, repeat_bins, **wrapper_args)

    def clamp_spec(self, xs: list | tuple):
        return [x if c is None else x.clamp(min=c[0], max=c[1]) for x, c in zip(xs, self.clamps)]

    def _pre_norm(self, xs: list | tuple):
        assert len(xs) == self.num_feats
        clipped = self.clamp_spec(xs)
        return clipped[0] if self.num_feats == 1 else torch.stack(clipped, dim=1)

    def _post_denorm(self, v):
        curves = [v] if self.num_feats == 1 else list(v.unbind(dim=1))
        assert len(curves) == self.num_feats
        return self.clamp_spec(curves)


class RepetitiveDiffusion(_RepeatBins, GaussianDiffusion):
    def __init__(self, vmin: float | int | list, vmax: float | int | list, repeat_bins: int, timesteps=1000, k_step=1000,
                 backbone_type=None, backbone_args=None, betas=None):
        self._init_bins(vmin, vmax, repeat_bins, timesteps=timesteps, k_step=k_step, backbone_type=backbone_type,
                        backbone_args=backbone_args, betas=betas)


class PitchDiffusion(_ClampedPitch, GaussianDiffusion):
    def __init__(self, vmin: float, vmax: float, cmin: float, cmax: float, repeat_bins, timesteps=1000, k_step=1000,
                 backbone_type=None, backbone_args=None, betas=None):
        self._init_pitch(vmin, vmax, cmin, cmax, repeat_bins, timesteps=timesteps, k_step=k_step,
                         backbone_type=backbone_type, backbone_args=backbone_args, betas=betas)


class MultiVarianceDiffusion(_MultiCurve, GaussianDiffusion):
    def __init__(self, ranges: List[Tuple[float, float]], clamps: List[Tuple[float | None, float | None] | None],
                 repeat_bins, timesteps=1000, k_step=1000, backbone_type=None, backbone_args=None, betas=None):
        self._init_curves(ranges, clamps, repeat_bins, timesteps=timesteps, k_step=k_step, backbone_type=backbone_type,
                          backbone_args=backbone_args, betas=betas)


class RepetitiveRectifiedFlow(_RepeatBins, RectifiedFlow):
    def __init__(self, vmin: float | int | list, vmax: float | int | list, repeat_bins: int, time_scale_factor=1000,
                 backbone_type=None, backbone_args=None):
        self._init_bins(vmin, vmax, repeat_bins, time_scale_factor=time_scale_factor, backbone_type=backbone_type,
                        backbone_args=backbone_args)


class PitchRectifiedFlow(_ClampedPitch, RectifiedFlow):
    def __init__(self, vmin: float, vmax: float, cmin: float, cmax: float, repeat_bins, time_scale_factor=1000,
                 backbone_type=None, backbone_args=None):
        self._init_pitch(vmin, vmax, cmin, cmax, repeat_bins, time_scale_factor=time_scale_factor,
                         backbone_type=backbone_type, backbone_args=backbone_args)


class MultiVarianceRectifiedFlow(_MultiCurve, RectifiedFlow):
    def __init__(self, ranges: List[Tuple[float, float]], clamps: List[Tuple[float | None, float | None] | None],
                 repeat_bins, time_scale_factor=1000, backbone_type=None, backbone_args=None):
        self._init_curves(ranges, clamps, repeat_bins, time_scale_factor=time_scale_factor, backbone_type=backbone_type,
                          backbone_args=backbone_args)
