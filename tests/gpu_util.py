"""Helpers for the -m gpu tests: HIP modules with the deterministic synthetic weights."""
import numpy as np
import torch

from diffsinger_amd import synth
from diffsinger_amd.hparams import hparams

BASE_HP = dict(hidden_size=256, schedule_type="linear", use_shallow_diffusion=False, diff_speedup=10,
               diff_accelerator="ddim", infer=False, sampling_algorithm="euler", sampling_steps=20)


def set_hp(**kw):
    hparams.clear()
    hparams.update(BASE_HP)
    hparams.update(kw)


def synth_params(kind, in_dims, n_feats, args, seed, hidden=256):
    shapes = synth.backbone_param_shapes(kind, in_dims, n_feats, hidden_size=hidden, **args)
    return synth.synth_state_dict(shapes, seed=seed)


def load_synth(module, params):
    module.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    return module.cuda().eval()


def make_backbone(kind, in_dims, n_feats, args, seed):
    from diffsinger_amd.backbones import build_backbone
    params = synth_params(kind, in_dims, n_feats, args, seed)
    return load_synth(build_backbone(in_dims, n_feats, kind, args), params), params


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rel_err(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.isfinite(a).all()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
