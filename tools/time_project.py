#!/usr/bin/env python3
"""A whole project through the acoustic model of the reference fork's default configuration (FastSpeech2 encoder ->
ConvNeXt aux decoder -> shallow reflow, euler 20, LYNXNet 6x1024) and the NSF-HiFiGAN vocoder: 8 segments of different
lengths, tokens in, waveform out - one by one (the reference's order) and with the acoustic model on ragged batches."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from diffsinger_amd import harness, synth  # noqa: E402
from diffsinger_amd.hparams import hparams  # noqa: E402
from diffsinger_amd.toplevel import DiffSingerAcoustic  # noqa: E402
from diffsinger_amd.vocoder import Generator, NsfHifiGAN  # noqa: E402

bargs = dict(num_layers=6, num_channels=1024, expansion_factor=2, kernel_size=31, activation="PReLU", strong_cond=True)
hparams.clear()
hparams.update(hop_size=512, audio_sample_rate=44100, hidden_size=256, enc_layers=4, enc_ffn_kernel_size=3, ffn_act="gelu",
               dropout=0.1, num_heads=2, use_pos_embed=True, rel_pos=True, use_rope=True, use_spk_id=False, num_spk=1,
               use_lang_id=False, num_lang=1, use_key_shift_embed=False, use_speed_embed=False, schedule_type="linear",
               use_shallow_diffusion=True, diffusion_type="reflow", T_start=0.4, T_start_infer=0.4, time_scale_factor=1000,
               sampling_algorithm="euler", sampling_steps=20, timesteps=1000, K_step=400, K_step_infer=400,
               backbone_type="lynxnet", backbone_args=bargs, spec_min=[-12.0], spec_max=[0.0],
               shallow_diffusion_args=dict(aux_decoder_arch="convnext", val_gt_start=False,
                                           aux_decoder_args=dict(num_channels=512, num_layers=6, kernel_size=7, dropout_rate=0.1)))
phones = [f"p{i}" for i in range(40)]
table = harness.SimplePhonemeTable(phones)
model = DiffSingerAcoustic(len(table), 128)
sd = dict(model.state_dict())
sd.update({"fs2." + k: torch.from_numpy(v) for k, v in synth.synth_state_dict(synth.fs2_acoustic_param_shapes(len(table)), seed=44).items()})
sd.update({"diffusion.velocity_fn." + k: torch.from_numpy(v) for k, v in synth.synth_state_dict(
    synth.backbone_param_shapes("lynxnet", 128, 1, hidden_size=256, **bargs), seed=42).items()})
sd.update({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(
    synth.convnext_param_shapes(256, 128, prefix="aux_decoder.decoder."), seed=43).items()})
model.load_state_dict(sd, strict=True)
model = model.cuda().eval()
vh = dict(synth.NSF_HIFIGAN_DEFAULT)
gen = Generator(vh)
gen.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(
    synth.nsf_hifigan_param_shapes(vh), seed=45, gain=0.7).items()}, strict=True)
h = harness.AcousticHarness(model, NsfHifiGAN(gen.cuda().eval()), table, device="cuda")

rng = np.random.Generator(np.random.PCG64(9))
segs, offset = [], 0.0
for i, seconds in enumerate((11.6, 10.8, 9.9, 8.8, 8.1, 7.4, 6.5, 5.6)):
    n_ph = int(seconds * 10)
    w = rng.random(n_ph) + 0.3
    dur = w / w.sum() * seconds
    f0 = 220.0 * 2.0 ** rng.uniform(-0.3, 0.3, int(seconds * 100))
    segs.append(dict(offset=offset, ph_seq=" ".join(phones[j] for j in rng.integers(0, 40, n_ph)),
                     ph_dur=" ".join(str(round(float(d), 5)) for d in dur), f0_seq=" ".join(str(round(float(v), 1)) for v in f0),
                     f0_timestep="0.01", seed=100 + i))
    offset += seconds + 0.2
total = sum((11.6, 10.8, 9.9, 8.8, 8.1, 7.4, 6.5, 5.6))
for bs in (1, 8):
    for _ in range(2):
        h.run_inference(segs, batch_size=bs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        track = h.run_inference(segs, batch_size=bs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"batch_size={bs}: {dt * 1e3:.1f} ms for {total:.1f} s of audio in 8 segments (RTF {dt / total:.5f}), "
          f"track {track.shape[0] / 44100:.1f} s", flush=True)
