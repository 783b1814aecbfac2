"""-m gpu: the C-ABI used from plain C (examples/c_abi_denoise.c, gcc -std=c99, no Python / torch in the process):
create -> load_weight (host pointers) -> finalize -> prepare_cond -> denoise -> sample (hipGraph), compared with the
Python shim on the same inputs.  Same library, same kernels; the only difference is the SinusoidalPosEmb frequency
table, which the shim uploads as computed by torch and the C program leaves to the library's own libm (expf): the
results agree to a few 1e-7 of the output range."""
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from diffsinger_amd import _lib, schedule, synth  # noqa: E402
from gpu_util import dev, make_backbone, set_hp  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_program_matches_python_shim(tmp_path):
    assert torch.cuda.is_available()
    exe = tmp_path / "c_abi_demo"
    libdir = os.path.join(ROOT, "diffsinger_amd")
    cmd = ["gcc", "-std=c99", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "c_abi_denoise.c"), "-L" + libdir, "-ldsdenoise", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    subprocess.run(cmd, check=True)
    set_hp()
    args = dict(num_layers=4, num_channels=64, dilation_cycle_length=2)
    bsz, t_len, hidden, bins = 2, 150, 256, 32
    net, params = make_backbone("wavenet", bins, 1, args, 45)
    with open(tmp_path / "weights.bin", "wb") as f:
        f.write(struct.pack("<i", len(params)))
        for name, arr in params.items():
            nb = name.encode()
            f.write(struct.pack("<i", len(nb)) + nb + struct.pack("<i", arr.ndim) + struct.pack(f"<{arr.ndim}q", *arr.shape))
            f.write(np.ascontiguousarray(arr, np.float32).tobytes())
    cond = synth.synth_normal((bsz, hidden, t_len), 1)
    x = synth.synth_normal((bsz, 1, bins, t_len), 2)
    t = np.array([400.0, 30.5], np.float32)
    with open(tmp_path / "inputs.bin", "wb") as f:
        f.write(struct.pack("<4i", bsz, t_len, hidden, bins) + cond.tobytes() + x.tobytes() + t.tobytes())
    env = dict(os.environ, HIP_FORCE_DEV_KERNARG="1")
    res = subprocess.run([str(exe), str(tmp_path / "weights.bin"), str(tmp_path / "inputs.bin"), str(tmp_path / "out.bin")],
                         capture_output=True, text=True, env=env, timeout=120)
    assert res.returncode == 0, res.stderr
    assert f"api v{_lib.lib().dsd_api_version()}" in res.stdout and "kernels/NFE 11" in res.stdout
    got = np.fromfile(tmp_path / "out.bin", dtype=np.float32).reshape(2, bsz, 1, bins, t_len)
    # the same two calls through the Python shim
    with torch.no_grad():
        want_nfe = net(dev(x), dev(t), dev(cond)).cpu().numpy()
    evals = []
    for k in range(5):
        evals.append(schedule.Eval(0, 900.0 - 200.0 * k,
                                   [(0, [(0, float(np.float32(0.9 + 0.01 * k))), (_lib.DSD_SRC_MODEL, float(np.float32(-0.2 + 0.03 * k)))])]))
    prog = schedule.Program(1, 0, evals)

    class Runner(torch.nn.Module):
        pass
    from diffsinger_amd.diffusion import _SamplerMixin
    runner = type("R", (_SamplerMixin,), {})()
    runner.denoise_fn = net
    entry = (prog,) + _lib.program_to_c(prog)
    want_samp = runner._run_program(entry, dev(cond), dev(x), transpose=False).cpu().numpy()
    np.testing.assert_allclose(got[0], want_nfe, rtol=0, atol=2e-6 * np.abs(want_nfe).max())
    np.testing.assert_allclose(got[1], want_samp, rtol=0, atol=1e-5 * np.abs(want_samp).max())
    net.release_native()
