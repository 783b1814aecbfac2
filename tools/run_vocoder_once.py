"""Runs the NSF-HiFiGAN generator three times at B=1, T=1000 (for `rocprofv3 --kernel-trace -- python3 tools/run_vocoder_once.py`)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffsinger_amd import synth
from diffsinger_amd.vocoder import Generator
h = dict(synth.NSF_HIFIGAN_DEFAULT)
gen = Generator(h)
gen.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(synth.nsf_hifigan_param_shapes(h), seed=6, gain=0.7).items()}, strict=True)
gen = gen.cuda().eval()
bsz, t_len = 1, 1000
mel = torch.from_numpy(synth.synth_normal((bsz, 128, t_len), 7) * 3 - 11).cuda()
f0 = torch.full((bsz, t_len), 220.0, device="cuda")
noise = torch.randn((bsz, t_len * 512, 9), device="cuda"); ri = torch.rand(9, device="cuda")
with torch.no_grad():
    for _ in range(3):
        gen(mel, f0, rand_ini=ri, noise=noise)
torch.cuda.synchronize()
