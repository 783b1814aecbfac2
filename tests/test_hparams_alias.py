"""use_reference_hparams(): one dict object for every reader, whatever was imported first (INTEGRATION.md route B'')."""
import sys
import types


def test_every_module_follows_the_reference_dict(monkeypatch):
    import diffsinger_amd
    import importlib
    hp_mod = importlib.import_module("diffsinger_amd.hparams")     # (the package attribute of that name is the dict itself)
    # readers imported BEFORE the call, as a host application may have done
    import diffsinger_amd.backbones as b
    import diffsinger_amd.diffusion as d
    import diffsinger_amd.encoder as e
    import diffsinger_amd.harness as h
    import diffsinger_amd.toplevel as t
    import diffsinger_amd.variance as v
    import diffsinger_amd.variance_harness as vh
    readers = [diffsinger_amd, hp_mod, b, d, e, h, t, v, vh]
    old = hp_mod.hparams
    saved = dict(old)
    ref = {"hidden_size": 192, "use_shallow_diffusion": True}
    utils = types.ModuleType("utils")
    utils_hp = types.ModuleType("utils.hparams")
    utils_hp.hparams = ref
    utils.hparams = utils_hp
    monkeypatch.setitem(sys.modules, "utils", utils)
    monkeypatch.setitem(sys.modules, "utils.hparams", utils_hp)
    try:
        old["only_here"] = 7
        got = hp_mod.use_reference_hparams()
        assert got is ref and ref["only_here"] == 7 and ref["hidden_size"] == 192       # carried over, not overwritten
        for m in readers:
            assert m.hparams is ref, m.__name__
        ref["K_step_infer"] = 123               # a later mutation by the host (scripts/infer.py --depth) is seen everywhere
        assert d.hparams["K_step_infer"] == 123 and t.hparams.get("use_shallow_diffusion") is True
    finally:        # back to the package's own dict for the other tests
        for m in readers:
            m.hparams = old
        old.clear()
        old.update(saved)
