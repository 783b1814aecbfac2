import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from gpu_util import make_backbone, set_hp, dev
from diffsinger_amd import synth
from oracle import backbones as ob
set_hp()
args = dict(num_layers=4, num_channels=256, dilation_cycle_length=4)
net, params = make_backbone("wavenet", 128, 1, args, 42)
for (b, t) in [(1, 96), (1, 1), (1, 1000)]:
    x = synth.synth_normal((b, 1, 128, t), 11); c = synth.synth_normal((b, 256, t), 12)
    tt = np.full((b,), 100.0, np.float32)
    with torch.no_grad():
        out = net(dev(x), dev(tt), dev(c))
    torch.cuda.synchronize()
    want = ob.wavenet_forward(params, x, tt, c, dilation_cycle_length=4)
    print(b, t, float(np.abs(out.cpu().numpy() - want).max() / np.abs(want).max()), flush=True)
