"""Build libdsdenoise.so in-tree with hipcc for gfx950 (MI355X).

The shared library is the product's only compute path.  It is built IN the source tree
(`diffsinger_amd/libdsdenoise.so`, git-ignored) so that it travels to the GPU box with the
repository snapshot; there is no JIT cache and no fallback.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdsdenoise.so")
SOURCES = ["gemm.hip", "wn_layer.hip", "wn_layer_x3.hip", "wn_rowsplit.hip", "wn_rows.hip", "wn_edge.hip", "lynx_layer.hip", "lynx_x3.hip", "aux_kernels.hip", "encoder_kernels.hip", "vocoder_kernels.hip", "tconv.hip", "api.hip"]
HEADERS = [os.path.join(CSRC, "dsd_internal.h"), os.path.join(os.path.dirname(HERE), "include", "dsdenoise.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
FILE_FLAGS = {}           # per-file extra flags (none needed today; tools/check_resources.py and the stamp tools honour them)
FLAGS += os.environ.get("DSD_EXTRA_HIPCC_FLAGS", "").split()      # diagnostic A/B builds (e.g. -DDSD_ST_AUX=0)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(src, []) + ["-c", s, "-o", o]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if out.strip() and verbose:
            print(out)
        if p.returncode != 0:
            failed = True
            print(f"[build] {src} FAILED", file=sys.stderr)
    if failed:
        raise RuntimeError("hipcc failed")
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
