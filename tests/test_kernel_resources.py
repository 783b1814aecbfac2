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


def test_no_store_data_overwrite_hazard():
    """No 16-byte buffer store is followed within two issue slots by a VALU write of its data registers (the raw-buffer
    store form with a register soffset gets no wait state from hipcc; seen to corrupt results on MI355X - tools/
    check_store_hazard.py).  The files that use raw buffer stores."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_store_hazard.py"), "wn_layer.hip", "wn_rowsplit.hip",
                        "wn_rows.hip", "wn_edge.hip", "lynx_layer.hip"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
