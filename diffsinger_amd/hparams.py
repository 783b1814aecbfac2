"""The process-global configuration dict the hot path reads (mirror of `utils/hparams.py:13`).

The reference reads `hparams` at construction AND at call time (wavenet.py:65; ddpm.py:67,73,222-245;
reflow.py:21,107,119-120).  `use_reference_hparams()` makes this name an alias of the reference's own
dict when DiffSinger is importable, so `scripts/infer.py --steps/--depth` mutations are seen here too.
"""
hparams: dict = {}


def set_hparams(**kw):
    hparams.update(kw)
    return hparams


def use_reference_hparams():
    """Alias to `utils.hparams.hparams` of an importable DiffSinger checkout.

    Every module of this package that did `from .hparams import hparams` holds a reference to the OLD dict object; all of
    them (whichever are imported already, in any order) are rebound to the reference's dict, and the old dict's entries are
    carried over, so no reader is left behind on a dict that is never filled."""
    global hparams
    import sys
    from utils.hparams import hparams as ref  # noqa: WPS433  (reference package, optional)
    old = hparams
    if ref is old:
        return ref
    ref.update({k: v for k, v in old.items() if k not in ref})
    hparams = ref
    for name, mod in list(sys.modules.items()):
        if mod is not None and (name == "diffsinger_amd" or name.startswith("diffsinger_amd.")) \
                and getattr(mod, "hparams", None) is old:
            mod.hparams = ref
    return ref


def _merge(into: dict, new: dict):
    """Nested update: a dict value is merged key by key into an existing dict, anything else replaces."""
    for k, v in new.items():
        if isinstance(v, dict) and isinstance(into.get(k), dict):
            _merge(into[k], v)
        else:
            into[k] = v


def load_config(config_path, overrides: dict = None, root=None, update_global: bool = True) -> dict:
    """Read a DiffSinger YAML configuration the way `utils/hparams.py:56-78` does: `base_config` entries (one path or a
    list) are loaded first, depth first, each at most once - a path starting with '.' is relative to the file naming it,
    any other to `root` (default: the current directory, as in the reference) - and later files override earlier ones key
    by key through nested dicts.  The complete `config.yaml` a training run saves in its work directory has no
    `base_config` and loads as is.  `overrides` are applied last; with `update_global` the result replaces the
    process-global `hparams` contents.  YAML is read with `yaml.safe_load`."""
    import os

    import yaml
    seen = set()

    def load(path):
        path = os.path.normpath(path)
        with open(path if os.path.isabs(path) or root is None else os.path.join(root, path), encoding='utf-8') as f:
            cfg = yaml.safe_load(f) or {}
        seen.add(path)
        bases = cfg.get('base_config', [])
        out = {}
        for base in ([bases] if isinstance(bases, str) else bases):
            if base in seen:
                continue
            if base.startswith('.'):
                base = os.path.normpath(os.path.join(os.path.dirname(path), base))
            _merge(out, load(base))
        _merge(out, cfg)
        return out

    cfg = load(str(config_path))
    if overrides:
        _merge(cfg, overrides)
    if update_global:
        hparams.clear()
        hparams.update(cfg)
    return cfg
