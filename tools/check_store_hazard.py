#!/usr/bin/env python3
"""Scan the gfx950 ISA of the hand-written kernel files for a hazard hipcc (ROCm 7.2, AMD clang 22) does not cover:

    buffer_store_dwordx4 v[4:7], v18, s[0:3], s27 offen sc1
    v_cndmask_b32_e32 v4, 0, v12, vcc          <- overwrites the store's first data register in the very next slot

A store of more than 8 bytes reads its data registers over several cycles after issue; a VALU write to one of them in the next
one or two issue slots can land first (seen on MI355X in wn_out_rw_kernel<4, *>: the stored vector's first element was the
NEXT item's operand, nondeterministically, in ~0.4 % of the elements).  The ISA manuals list the case (VMEM store of > 64
bits of data followed by a write of its data VGPRs: 1-2 wait states); the compiler inserts the s_nop for stores without an
SGPR offset but not for the `soffset` form the raw-buffer builtins produce.  The kernels therefore issue `s_nop 1` behind
every 16-byte raw buffer store (st4 helpers); this script verifies that no such pair is left.

Usage: python tools/check_store_hazard.py            (exit 1 if a hazard pair is found; no GPU needed)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "diffsinger_amd", "csrc")
FILES = ["gemm.hip", "wn_layer.hip", "wn_rowsplit.hip", "wn_rows.hip", "wn_edge.hip", "lynx_layer.hip", "aux_kernels.hip", "tconv.hip",
         "encoder_kernels.hip", "vocoder_kernels.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
WAIT_STATES = 2


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def scan(path):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "x.s")
        r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "-w", "-S", "--cuda-device-only",
                            "-I", CSRC, "-o", out, path] + os.environ.get("DSD_EXTRA_HIPCC_FLAGS", "").split(),
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-2000:])
        lines = open(out).read().splitlines()
    bad, func = [], "?"
    ins = []
    for ln in lines:
        s = ln.strip()
        if s.endswith(":") and s.startswith("_Z"):
            func = s[:-1]
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            continue
        ins.append((func, s.split(";")[0].strip()))
    for i, (fn, s) in enumerate(ins):
        m = re.match(r"(buffer|global|flat)_store_dwordx[34]\s+(.*)", s)
        if not m:
            continue
        ops = [t.strip() for t in m.group(2).split(",")]
        data = regs(ops[0]) if m.group(1) == "buffer" else (regs(ops[1]) if len(ops) > 1 else set())
        slots = 0
        for fn2, t in ins[i + 1:i + 1 + 4]:
            if fn2 != fn:
                break
            nop = re.match(r"s_nop\s+(\d+)", t)
            if nop:
                slots += int(nop.group(1)) + 1
                continue
            if slots >= WAIT_STATES:
                break
            if re.match(r"v_(?!mfma)", t) or re.match(r"(buffer|global|flat)_load|ds_read", t):
                dst = t.split(None, 1)[1].split(",")[0].strip() if " " in t else ""
                # only VALU writes race with the store's data read (loads return much later)
                if re.match(r"v_", t) and regs(dst) & data:
                    bad.append((fn, s, t))
                    break
            slots += 1
    return bad


def main():
    total = 0
    files = sys.argv[1:] or FILES
    for f in files:
        p = os.path.join(CSRC, f)
        if not os.path.exists(p):
            continue
        bad = scan(p)
        for fn, st, nxt in bad:
            name = subprocess.run(["c++filt", fn], capture_output=True, text=True).stdout.strip() or fn
            print(f"{f}: {name[:90]}\n    {st}\n    {nxt}")
        total += len(bad)
        print(f"{f}: {len(bad)} store / VALU-overwrite pairs within {WAIT_STATES} wait states")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
