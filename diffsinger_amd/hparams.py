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
    """Alias to `utils.hparams.hparams` of an importable DiffSinger checkout."""
    global hparams
    from utils.hparams import hparams as ref  # noqa: WPS433  (reference package, optional)
    ref.update({k: v for k, v in hparams.items() if k not in ref})
    hparams = ref
    import diffsinger_amd.backbones as _b
    import diffsinger_amd.diffusion as _d
    _b.hparams = ref
    _d.hparams = ref
    return ref
