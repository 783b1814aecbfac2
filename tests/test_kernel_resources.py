"""The hand-scheduled kernels must stay in registers: no scratch, no spills (tools/check_resources.py; hipcc cross-compiles,
no GPU needed).  About a minute of compile time."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hot_kernels_use_no_scratch():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_resources.py")], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "wn_layer_kernel" in r.stdout and "lx_pw1_kernel" in r.stdout and "wn_conv_rs_kernel" in r.stdout
