#!/usr/bin/env python3
"""Time the once-per-utterance neighbours of the loop: the aux decoder pass (dsd_aux_decode) and the FastSpeech2
encoder (dsd_encode).  GPU box only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsinger_amd import synth
from diffsinger_amd.hparams import hparams
hparams.update(hidden_size=256)
from diffsinger_amd.aux_decoder import AuxDecoderAdaptor

a = AuxDecoderAdaptor(256, 128, 1, [-12.0], [0.0], "convnext", dict(num_channels=512, num_layers=6, kernel_size=7))
sd = synth.synth_state_dict(synth.convnext_param_shapes(256, 128, prefix="decoder."), seed=3)
a.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
a = a.cuda().eval()
for bsz, t_len in ((1, 1000), (8, 1000), (16, 1000)):
    cond = torch.from_numpy(synth.synth_normal((bsz, t_len, 256), 1)).cuda()
    with torch.no_grad():
        for _ in range(5):
            a(cond, infer=True)
        torch.cuda.synchronize()
        n = 50
        t0 = time.perf_counter()
        for _ in range(n):
            a(cond, infer=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    st = a.decoder.stats()
    fl = st["flops_per_frame_nfe"] * bsz * t_len
    print(f"B={bsz} T={t_len}: {dt*1e3:.3f} ms/pass  {bsz*t_len/dt/1e6:.2f} M frames/s  {fl/dt/1e12:.1f} TFLOP/s", flush=True)

hparams.update(enc_layers=4, enc_ffn_kernel_size=3, ffn_act="gelu", dropout=0.1, num_heads=2, use_pos_embed=True,
               rel_pos=True, use_rope=True, use_spk_id=False, num_spk=1, use_lang_id=False, num_lang=1)
from diffsinger_amd.encoder import FastSpeech2Acoustic
enc = FastSpeech2Acoustic(60)
enc.load_state_dict({k: torch.from_numpy(v) for k, v in
                     synth.synth_state_dict(synth.fs2_acoustic_param_shapes(60), seed=4).items()}, strict=True)
enc = enc.cuda().eval()
for bsz, n_tok, t_len in ((1, 120, 1000), (8, 120, 1000), (1, 300, 2500)):
    tokens = (torch.arange(n_tok, device="cuda") % 59 + 1)[None].expand(bsz, n_tok).contiguous()
    mel2ph = (torch.arange(t_len, device="cuda") * n_tok // t_len + 1)[None].expand(bsz, t_len).contiguous()
    f0 = torch.full((bsz, t_len), 220.0, device="cuda")
    with torch.no_grad():
        for _ in range(5):
            enc(tokens, mel2ph, f0)
        torch.cuda.synchronize()
        n = 50
        t0 = time.perf_counter()
        for _ in range(n):
            enc(tokens, mel2ph, f0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    print(f"encoder B={bsz} T_txt={n_tok} T={t_len}: {dt*1e3:.3f} ms/pass", flush=True)

from diffsinger_amd.vocoder import Generator
h = dict(synth.NSF_HIFIGAN_DEFAULT)
gen = Generator(h)
gen.load_state_dict({k: torch.from_numpy(v) for k, v in
                     synth.synth_state_dict(synth.nsf_hifigan_param_shapes(h), seed=6, gain=0.7).items()}, strict=True)
gen = gen.cuda().eval()
st = None
for bsz, t_len in ((1, 1000), (4, 1000)):
    mel = torch.from_numpy(synth.synth_normal((bsz, 128, t_len), 7) * 3 - 11).cuda()
    f0 = torch.full((bsz, t_len), 220.0, device="cuda")
    noise = torch.randn((bsz, t_len * 512, 9), device="cuda")
    ri = torch.rand(9, device="cuda")
    with torch.no_grad():
        for _ in range(2):
            gen(mel, f0, rand_ini=ri, noise=noise)
        torch.cuda.synchronize()
        n = 5
        t0 = time.perf_counter()
        for _ in range(n):
            gen(mel, f0, rand_ini=ri, noise=noise)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    flops = 0
    ch = 512
    rate = 1
    for u, k in zip(h["upsample_rates"], h["upsample_kernel_sizes"]):
        ch //= 2
        flops += 2 * (2 * ch) * ch * k * rate          # transposed conv, per input frame of that stage
        rate *= u
        flops += rate * sum(2 * ch * ch * rk * 2 * 3 for rk in h["resblock_kernel_sizes"])
    flops += 2 * 128 * 512 * 7 + rate * 2 * ch * 7
    print(f"vocoder B={bsz} T={t_len}: {dt*1e3:.2f} ms/pass  RTF {dt / (bsz * t_len * 512 / 44100):.5f}  "
          f"{flops * bsz * t_len / dt / 1e12:.1f} TFLOP/s", flush=True)

# ---- variance model at the sizes of configs/variance.yaml: tokens -> durations, pitch, energy + breathiness
import importlib.util
spec = importlib.util.spec_from_file_location("variance_cases", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "variance_cases.py"))
vc = importlib.util.module_from_spec(spec); spec.loader.exec_module(vc)
from diffsinger_amd.variance import DiffSingerVariance
hp = dict(vc.ENC_HP)
hp.update(enc_layers=4, predict_dur=True, predict_pitch=True, predict_energy=True, predict_breathiness=True,
          diffusion_type="reflow", sampling_algorithm="euler", sampling_steps=20, infer=True,
          dur_prediction_args=dict(arch="fs2", hidden_size=512, dropout=0.1, num_layers=5, kernel_size=3, log_offset=1.0, loss_type="mse"),
          pitch_prediction_args=dict(pitd_norm_min=-8.0, pitd_norm_max=8.0, pitd_clip_min=-12.0, pitd_clip_max=12.0, repeat_bins=64,
                                     backbone_type="wavenet", backbone_args=dict(num_layers=20, num_channels=256, dilation_cycle_length=5)),
          variances_prediction_args=dict(total_repeat_bins=48, backbone_type="wavenet",
                                         backbone_args=dict(num_layers=10, num_channels=192, dilation_cycle_length=4)))
hparams.clear(); hparams.update(hp)
vm = DiffSingerVariance(60)
shapes = vc.sorted_param_shapes(vm.named_parameters())
vm.load_state_dict({k: torch.from_numpy(v) for k, v in vc.synth_weights(shapes, 9).items()}, strict=False)
vm = vm.cuda().eval()
for bsz, n_ph, t_len in ((1, 120, 1000), (8, 120, 1000)):
    n_word = 40
    tokens = (torch.arange(n_ph, device="cuda") % 59 + 1)[None].expand(bsz, n_ph).contiguous()
    ph2word = (torch.arange(n_ph, device="cuda") // 3 + 1)[None].expand(bsz, n_ph).contiguous()
    midi = torch.full((bsz, n_ph), 60, device="cuda", dtype=torch.long)
    word_dur = torch.full((bsz, n_word), t_len // n_word, device="cuda", dtype=torch.long)
    base_pitch = torch.full((bsz, t_len), 60.0, device="cuda")
    call = lambda: vm(tokens, midi, ph2word, word_dur=word_dur, base_pitch=base_pitch, infer=True)
    with torch.no_grad():
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            call()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        t1 = time.perf_counter()
        for _ in range(n):
            vm.fs2(tokens, midi=midi, ph2word=ph2word, word_dur=word_dur)
        torch.cuda.synchronize()
        dt_enc = (time.perf_counter() - t1) / n
    print(f"variance model B={bsz} T_ph={n_ph} T={t_len}: {dt*1e3:.3f} ms per call (encoder + duration predictor {dt_enc*1e3:.3f} ms)", flush=True)
