#!/usr/bin/env python3
"""Compile the hand-scheduled kernel files for gfx950 with -Rpass-analysis=kernel-resource-usage and print / check the
register, scratch and occupancy figures of every instantiation (no GPU needed: hipcc cross-compiles).

Why a check: these kernels keep whole operand sets in registers through fully unrolled loops; one loop that the compiler
does not unroll turns a register array into scratch memory and a 63 us kernel into a 280 us one without any wrong result
(it happened once: a barrier inside an unrolled K walk).  `python tools/check_resources.py` exits non-zero if any kernel
of these files uses scratch or spills; tests/test_kernel_resources.py runs it."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "diffsinger_amd", "csrc")
FILES = ["wn_layer.hip", "wn_layer_x3.hip", "wn_rowsplit.hip", "wn_rows.hip", "wn_edge.hip", "lynx_layer.hip", "lynx_x3.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffsinger_amd.build_native import FILE_FLAGS  # noqa: E402  (the per-file flags of the product build)


def analyse(path):
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-w",
                            "-Rpass-analysis=kernel-resource-usage"] + FILE_FLAGS.get(os.path.basename(path), []) +
                           ["-c", path, "-o", os.path.join(tmp, "x.o")],
                           capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-2000:])
    out, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            out.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\]| \[waves/SIMD\]| \[bytes/block\])?: (\S+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    return out


def main():
    bad = []
    for f in FILES:
        for k in analyse(os.path.join(CSRC, f)):
            name = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip() or k["name"]
            scratch, spill = int(k.get("ScratchSize", 0)), int(k.get("VGPRs Spill", 0)) + int(k.get("SGPRs Spill", 0))
            print(f"{f:16s} {name[:64]:64s} VGPR {k.get('VGPRs'):>4s} AGPR {k.get('AGPRs'):>4s} scratch {scratch:4d} "
                  f"spills {spill:3d} waves/SIMD {k.get('Occupancy')}")
            if scratch or spill:
                bad.append(name)
    if bad:
        print("scratch / spills in:", *bad, sep="\n  ", file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
