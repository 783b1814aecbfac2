"""Host-side checks of the variance model shim (no GPU): the parameter names and shapes match the reference's
DiffSingerVariance for every G12 configuration (so a checkpoint loads strictly), and the rhythm / length regulators
reproduce the reference's integers from its own predicted durations."""
import os

import numpy as np
import pytest
import torch

import variance_cases as vc
from diffsinger_amd.hparams import hparams

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("tag", list(vc.CASES))
def test_parameter_layout_matches_reference(tag):
    from diffsinger_amd.variance import DiffSingerVariance
    g = np.load(os.path.join(GOLDEN, "g12_variance_model.npz"))
    hparams.clear()
    hparams.update(vc.case_hparams(tag), infer=True)
    model = DiffSingerVariance(vc.CASES[tag]["vocab"])
    shapes = vc.sorted_param_shapes(model.named_parameters())
    assert [f"{n}:{'x'.join(map(str, sh))}" for n, sh in shapes.items()] == [str(s) for s in g[f"{tag}_params"]]
    with pytest.raises(RuntimeError, match="no CPU path|only on an MI355X"):
        with torch.no_grad():
            model(**{k: torch.from_numpy(v) for k, v in vc.case_inputs(tag).items() if not isinstance(v, dict)})


def test_regulators_reproduce_reference_integers():
    from diffsinger_amd.variance import LengthRegulator, RhythmRegulator
    g = np.load(os.path.join(GOLDEN, "g12_variance_model.npz"))
    inp = vc.case_inputs("word_reflow")
    aligned = RhythmRegulator()(torch.from_numpy(g["word_reflow_dur"]), torch.from_numpy(inp["ph2word"]),
                                torch.from_numpy(inp["word_dur"]))
    assert np.array_equal(aligned.numpy(), g["word_reflow_dur_aligned"])
    assert np.array_equal(LengthRegulator()(aligned).numpy(), g["word_reflow_mel2ph"])


def test_unsupported_configurations_raise():
    from diffsinger_amd.variance import DiffSingerVariance
    hparams.clear()
    hparams.update(vc.case_hparams("dur_only"), infer=True)
    hparams["ffn_act"] = "tanh"
    with pytest.raises(ValueError, match="not a valid activation"):      # the reference's error (common_layers.py:135-136)
        DiffSingerVariance(10)
    hparams["ffn_act"] = "gelu"
    hparams["diffusion_type"] = "flow"
    with pytest.raises(ValueError, match="Invalid diffusion type"):
        DiffSingerVariance(10)
