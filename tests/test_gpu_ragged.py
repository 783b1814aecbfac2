"""-m gpu: ragged batches (dsd_set_lengths).  The reference runs one utterance per call because padding changes results
near the end of the shorter items; with per-item lengths every convolution along time treats an item's padded frames as
its zero padding, so item b of a padded batch must come out as if it had been run alone at T = lengths[b].  Checked
against exactly that: the same module run on each item alone (truncated inputs, the same x_T), for both backbones, the
samplers, the aux decoder and the top-level acoustic decoder."""
import numpy as np
import pytest
import torch

from diffsinger_amd import synth
from diffsinger_amd.hparams import hparams
from gpu_util import dev, load_synth, make_backbone, set_hp, synth_params

pytestmark = pytest.mark.gpu
LENS = [100, 57, 83, 1]
TOL = 2e-6


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    yield
    set_hp()


def close(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max()) <= TOL * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("kind,args", [
    ("wavenet", dict(num_layers=6, num_channels=128, dilation_cycle_length=3)),
    ("lynxnet", dict(num_layers=2, num_channels=128, expansion_factor=2, kernel_size=31, activation="PReLU", strong_cond=True)),
])
def test_single_evaluation_ragged_equals_alone(kind, args):
    set_hp()
    net, _ = make_backbone(kind, 64, 1, args, 50)
    bsz, t_max = len(LENS), max(LENS)
    x = dev(synth.synth_normal((bsz, 1, 64, t_max), 1))
    cond = dev(synth.synth_normal((bsz, 256, t_max), 2))
    t = dev(np.array([10.0, 400.0, 730.0, 999.0], np.float32))
    with torch.no_grad():
        dense = net(x, t, cond).clone()
        net.set_lengths(LENS, x.device)
        ragged = net(x, t, cond).clone()
        net.set_lengths(None, x.device)
        again = net(x, t, cond)
        assert torch.equal(dense, again)                    # lengths do not stick
        leaked = False
        for b, n in enumerate(LENS):
            alone = net(x[b:b + 1, :, :, :n].contiguous(), t[b:b + 1], cond[b:b + 1, :, :n].contiguous())
            assert close(ragged[b:b + 1, :, :, :n], alone), (kind, b)
            leaked |= n < t_max and not close(dense[b:b + 1, :, :, :n], alone)
        assert leaked, "the dense batch should differ near the end of the shorter items (otherwise this test shows nothing)"
        with pytest.raises(RuntimeError, match="lengths"):
            net.set_lengths(LENS[:2], x.device)
            net(x, t, cond)
        net.set_lengths(None, x.device)
        with pytest.raises(RuntimeError, match="exceeds"):
            net.set_lengths([t_max + 1] * bsz, x.device)
            net(x, t, cond)
        net.set_lengths(None, x.device)
    net.release_native()


@pytest.mark.parametrize("sampler", ["dpm-solver", "unipc", "reflow-euler"])
def test_sampling_loop_ragged_equals_alone(sampler):
    from diffsinger_amd.diffusion import GaussianDiffusion, RectifiedFlow
    args = dict(num_layers=4, num_channels=64, dilation_cycle_length=2)
    params = synth_params("wavenet", 32, 1, args, 60)
    if sampler == "reflow-euler":
        set_hp(sampling_algorithm="euler", sampling_steps=6)
        d = RectifiedFlow(32, 1, backbone_type="wavenet", backbone_args=args, spec_min=[-8.0], spec_max=[0.0])
        load_synth(d.velocity_fn, params)
    else:
        set_hp(diff_accelerator=sampler, diff_speedup=100, K_step_infer=1000)
        d = GaussianDiffusion(32, 1, timesteps=1000, k_step=1000, backbone_type="wavenet", backbone_args=args,
                              spec_min=[-8.0], spec_max=[0.0])
        load_synth(d.denoise_fn, params)
    d = d.cuda().eval()
    bsz, t_max = len(LENS), max(LENS)
    cond = dev(synth.synth_normal((bsz, t_max, 256), 3))
    noise = dev(synth.synth_normal((bsz, 1, 32, t_max), 4))
    with torch.no_grad():
        ragged = d(cond, infer=True, noise=noise, lengths=LENS).clone()
        for b, n in enumerate(LENS):
            alone = d(cond[b:b + 1, :n].contiguous(), infer=True, noise=noise[b:b + 1, :, :, :n].contiguous())
            assert close(ragged[b:b + 1, :n], alone), (sampler, b)
    d._backbone().release_native()


def test_acoustic_decoder_ragged_equals_alone():
    """Aux decoder (ConvNeXt) -> mask -> shallow reflow loop on LYNXNet: the top-level acoustic decoder, three segments in
    one batch against the same segments one by one."""
    from diffsinger_amd.toplevel import AcousticDecoder
    lens = [96, 41, 70]
    bargs = dict(num_layers=2, num_channels=128, expansion_factor=2, kernel_size=31, activation="PReLU", strong_cond=True)
    hparams.clear()
    hparams.update(hidden_size=256, schedule_type="linear", infer=False, use_shallow_diffusion=True, diffusion_type="reflow",
                   T_start=0.4, T_start_infer=0.4, time_scale_factor=1000, sampling_algorithm="euler", sampling_steps=5,
                   timesteps=1000, K_step=400, K_step_infer=400, backbone_type="lynxnet", backbone_args=bargs,
                   spec_min=[-12.0], spec_max=[0.0],
                   shallow_diffusion_args=dict(aux_decoder_arch="convnext", val_gt_start=False,
                                               aux_decoder_args=dict(num_channels=128, num_layers=2, kernel_size=7, dropout_rate=0.1)))
    m = AcousticDecoder(32)
    sd = dict(m.state_dict())
    sd.update({"diffusion.velocity_fn." + k: torch.from_numpy(v) for k, v in synth_params("lynxnet", 32, 1, bargs, 70).items()})
    sd.update({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(
        synth.convnext_param_shapes(256, 32, num_channels=128, num_layers=2, prefix="aux_decoder.decoder."), seed=71).items()})
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    bsz, t_max = len(lens), max(lens)
    cond = dev(synth.synth_normal((bsz, t_max, 256), 5))
    mel2ph = torch.zeros((bsz, t_max), dtype=torch.long, device="cuda")
    for b, n in enumerate(lens):
        mel2ph[b, :n] = torch.arange(n, device="cuda") // 7 + 1
        cond[b, n:] = 0.0                                  # what the length regulator's gather leaves at padded frames
    noise = dev(synth.synth_normal((bsz, 1, 32, t_max), 6))
    with torch.no_grad():
        out = m(cond, mel2ph, infer=True, noise=noise, lengths=lens)
        for b, n in enumerate(lens):
            alone = m(cond[b:b + 1, :n].contiguous(), mel2ph[b:b + 1, :n].contiguous(), infer=True,
                      noise=noise[b:b + 1, :, :, :n].contiguous())
            assert close(out.aux_out[b:b + 1, :n], alone.aux_out), b
            assert close(out.diff_out[b:b + 1, :n], alone.diff_out), b
            assert float(out.diff_out[b, n:].abs().max() if n < t_max else 0.0) == 0.0      # the padding mask still applies


def test_lazy_graph_policy_captures_on_second_use():
    """The default hipGraph policy: a (program, shape) runs eagerly the first time and is captured when it comes back;
    eager, captured and replayed runs give the same bits, and a new length does not pay for a capture."""
    from diffsinger_amd.diffusion import GaussianDiffusion
    args = dict(num_layers=4, num_channels=64, dilation_cycle_length=2)
    set_hp(diff_accelerator="dpm-solver", diff_speedup=100, K_step_infer=1000)
    d = GaussianDiffusion(32, 1, timesteps=1000, k_step=1000, backbone_type="wavenet", backbone_args=args,
                          spec_min=[-8.0], spec_max=[0.0])
    load_synth(d.denoise_fn, synth_params("wavenet", 32, 1, args, 60))
    d = d.cuda().eval()
    assert d.use_graph == "lazy"
    cond = dev(synth.synth_normal((2, 90, 256), 3))
    noise = dev(synth.synth_normal((2, 1, 32, 90), 4))
    cached = lambda: d.denoise_fn.stats()["graphs_cached"]  # noqa: E731
    with torch.no_grad():
        first = d(cond, infer=True, noise=noise).clone()
        assert cached() == 0
        second = d(cond, infer=True, noise=noise).clone()
        assert cached() == 1
        third = d(cond, infer=True, noise=noise)
        assert cached() == 1 and torch.equal(first, second) and torch.equal(first, third)
        other = d(cond[:, :70].contiguous(), infer=True, noise=noise[..., :70].contiguous())     # a new length: eager again
        assert cached() == 0 and tuple(other.shape) == (2, 70, 32)
        d.use_graph = True
        d(cond, infer=True, noise=noise)
        assert cached() == 1                                # immediate capture when asked for
    d.denoise_fn.release_native()


@pytest.mark.parametrize("bsz,t_max", [(2, 90), (6, 200), (32, 300)])
def test_ragged_in_every_tile_regime(bsz, t_max):
    """16-, 32- and 64-frame tiles each have their own list of valid tiles (the tile width follows the size of the launch):
    small, medium and large batches, lengths on, just below and just above tile boundaries, one single-frame item."""
    set_hp()
    args = dict(num_layers=6, num_channels=128, dilation_cycle_length=3)
    net, _ = make_backbone("wavenet", 64, 1, args, 51)
    rng = np.random.Generator(np.random.PCG64(bsz))
    special = [t_max, 64, 63, 65, 1, 128, 127, 129, 32, 33, 16, 17]
    lens = [min(v, t_max) for v in special[:bsz]] + [int(v) for v in rng.integers(1, t_max + 1, max(0, bsz - len(special)))]
    x = dev(synth.synth_normal((bsz, 1, 64, t_max), 7))
    cond = dev(synth.synth_normal((bsz, 256, t_max), 8))
    t = dev(np.linspace(5.0, 995.0, bsz).astype(np.float32))
    with torch.no_grad():
        net.set_lengths(lens, x.device)
        ragged = net(x, t, cond).clone()
        net.set_lengths(None, x.device)
        for b in (range(bsz) if bsz <= 6 else list(range(12)) + [bsz - 1]):
            n = lens[b]
            alone = net(x[b:b + 1, :, :, :n].contiguous(), t[b:b + 1], cond[b:b + 1, :, :n].contiguous())
            assert close(ragged[b:b + 1, :, :, :n], alone), (bsz, b, n)
    net.release_native()
