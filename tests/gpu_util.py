"""Helpers for the -m gpu tests: HIP modules with the deterministic synthetic weights."""
import json
import os

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


_PARITY_LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity.jsonl")
_seen = {}


def errors(a, b):
    """(max |a - b| / max |b|,  rms(a - b) / rms(b)): the second cannot be flattered by one large reference value (the
    random-weight sampler fixtures peak at ~260 on a [-12, 0] mel range)."""
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.isfinite(a).all()
    d = a - b
    return (float(np.abs(d).max() / max(np.abs(b).max(), 1e-30)),
            float(np.sqrt(np.mean(d * d)) / max(np.sqrt(np.mean(b * b)), 1e-30)))


def _record(mx, rms, tol, rms_tol):
    """Every comparison of a -m gpu run is appended to gpurun_out/parity.jsonl (tools/summarize_parity.py condenses the
    file into profiles/): a drift from 2e-6 to 4e-4 inside a 5e-4 tolerance is then visible."""
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" (")[0]
    k = _seen[test] = _seen.get(test, -1) + 1
    try:
        os.makedirs(os.path.dirname(_PARITY_LOG), exist_ok=True)
        with open(_PARITY_LOG, "a") as f:
            f.write(json.dumps({"test": test, "k": k, "max_rel": mx, "rms_rel": rms, "tol": tol, "rms_tol": rms_tol}) + "\n")
    except OSError:
        pass


def rel_err(a, b):
    mx, rms = errors(a, b)
    _record(mx, rms, None, None)
    return mx


def check(a, b, tol, rms_tol=None, what=""):
    """assert max-abs / max-ref <= tol AND rms / rms-ref <= rms_tol (default: tol)"""
    mx, rms = errors(a, b)
    rms_tol = tol if rms_tol is None else rms_tol
    _record(mx, rms, tol, rms_tol)
    assert mx <= tol and rms <= rms_tol, (what, mx, rms, tol, rms_tol)
    return mx
