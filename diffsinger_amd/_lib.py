"""ctypes binding of libdsdenoise.so (include/dsdenoise.h).

There is exactly one compute path: the HIP library.  If it is missing or does not load, importing
this module raises - there is no eager / CPU fallback to fall through to.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libdsdenoise.so")

DSD_MAX_TERMS = 8
DSD_MAX_OUT = 3
DSD_SRC_MODEL = -1
DSD_SRC_NOISE_BASE = -1000
DSD_SAMPLE_GRAPH = 1
DSD_SAMPLE_TRANSPOSE = 2
DSD_SAMPLE_GRAPH_LAZY = 4
BACKBONE_IDS = {"wavenet": 0, "lynxnet": 1}
AUX_CONVNEXT = 2
ACT_IDS = {"PReLU": 0, "SiLU": 1, "ReLU": 2}

EXPORTS = [
    "dsd_api_version", "dsd_create", "dsd_destroy", "dsd_last_error", "dsd_load_weight",
    "dsd_finalize_weights", "dsd_prepare_cond", "dsd_denoise", "dsd_sample", "dsd_get_stats",
    "dsd_kernel_timing", "dsd_kernel_timing_read", "dsd_kernel_timing_classes", "dsd_set_precision", "dsd_aux_decode", "dsd_encoder_create", "dsd_encode", "dsd_vocoder_create", "dsd_vocode",
    "dsd_token_encoder_create", "dsd_token_encode", "dsd_predict_dur", "dsd_cond_assemble", "dsd_set_lengths",
]
POS_ROPE, POS_REL, POS_NONE, POS_SIN = 0, 1, 2, 3       # DSD_POS_*
FFN_ACTS = {"gelu": 0, "relu": 1, "swish": 2, "swiglu": 3}    # DSD_FFN_* (TransformerFFNLayer, common_layers.py:126-136)
EMBED_FLAGS = {"energy": 1, "breathiness": 2, "voicing": 4, "tension": 8, "key_shift": 16, "speed": 32}


class DsdConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "struct_size", "backbone", "in_dims", "n_feats", "num_layers", "num_channels", "hidden_size",
        "dilation_cycle_length", "expansion_factor", "kernel_size", "activation", "strong_cond", "device")]


class DsdEncoderConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("vocab_size", C.c_int32), ("hidden_size", C.c_int32),
                ("enc_layers", C.c_int32), ("num_heads", C.c_int32), ("ffn_kernel_size", C.c_int32),
                ("num_spk", C.c_int32), ("num_lang", C.c_int32), ("embed_flags", C.c_uint32), ("pos_mode", C.c_int32),
                ("device", C.c_int32), ("ffn_act", C.c_int32)]


class DsdVocoderConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("num_mels", C.c_int32), ("sampling_rate", C.c_int32),
                ("upsample_initial_channel", C.c_int32), ("n_ups", C.c_int32), ("upsample_rates", C.c_int32 * 8),
                ("upsample_kernel_sizes", C.c_int32 * 8), ("resblock", C.c_int32), ("n_kernels", C.c_int32),
                ("resblock_kernel_sizes", C.c_int32 * 8), ("n_dilations", C.c_int32 * 8),
                ("resblock_dilation_sizes", (C.c_int32 * 4) * 8), ("harmonic_num", C.c_int32), ("mini_nsf", C.c_int32),
                ("noise_sigma", C.c_float), ("device", C.c_int32)]


class DsdTokenEncoderConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("hidden_size", C.c_int32), ("enc_layers", C.c_int32), ("num_heads", C.c_int32),
                ("ffn_kernel_size", C.c_int32), ("out_dims", C.c_int32), ("dur_layers", C.c_int32), ("dur_chans", C.c_int32),
                ("dur_kernel_size", C.c_int32), ("dur_offset", C.c_float), ("pos_mode", C.c_int32), ("device", C.c_int32),
                ("ffn_act", C.c_int32)]


class _AssembleGather(C.Structure):
    _fields_ = [("table", C.c_void_p), ("batch_stride", C.c_int64), ("rows", C.c_int64), ("idx", C.c_void_p),
                ("idx_offset", C.c_int64), ("scale", C.c_float), ("row_scale", C.c_void_p)]


class _AssembleTerm(C.Structure):
    _fields_ = [("s", C.c_void_p), ("v", C.c_void_p)]


class DsdAssembleArgs(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("B", C.c_int32), ("T", C.c_int32), ("H", C.c_int32),
                ("n_gather", C.c_int32), ("n_terms", C.c_int32), ("gather", _AssembleGather * 4), ("term", _AssembleTerm * 16)]


class DsdEncodeExtras(C.Structure):
    _fields_ = [("languages", C.c_void_p), ("spk_embed_id", C.c_void_p), ("spk_mix_embed", C.c_void_p),
                ("spk_mix_bstride", C.c_int64), ("spk_mix_tstride", C.c_int64), ("key_shift", C.c_void_p),
                ("speed", C.c_void_p), ("energy", C.c_void_p), ("breathiness", C.c_void_p), ("voicing", C.c_void_p),
                ("tension", C.c_void_p)]


class DsdTerm(C.Structure):
    _fields_ = [("src", C.c_int32), ("coef", C.c_float)]


class DsdLincomb(C.Structure):
    _fields_ = [("dst", C.c_int32), ("n_terms", C.c_int32), ("terms", DsdTerm * DSD_MAX_TERMS)]


class DsdEval(C.Structure):
    _fields_ = [("x_buf", C.c_int32), ("t", C.c_float), ("n_out", C.c_int32), ("out", DsdLincomb * DSD_MAX_OUT)]


class DsdProgram(C.Structure):
    _fields_ = [("n_bufs", C.c_int32), ("result_buf", C.c_int32), ("n_evals", C.c_int32),
                ("n_noise", C.c_int32), ("evals", C.POINTER(DsdEval))]


class DsdStats(C.Structure):
    _fields_ = [("weight_bytes", C.c_int64), ("workspace_bytes", C.c_int64),
                ("flops_per_frame_nfe", C.c_int64), ("bytes_per_frame_nfe", C.c_int64),
                ("kernels_per_nfe", C.c_int32), ("graphs_cached", C.c_int32),
                ("layer_launches", C.c_int32), ("fused_tiles", C.c_int32), ("split_tiles", C.c_int32),
                ("precision", C.c_int32)]


class DsdKernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("mean_ms", C.c_double), ("launches_timed", C.c_int64),
                ("launches", C.c_int64), ("evaluations", C.c_int64), ("flops_per_launch", C.c_double),
                ("bytes_per_launch", C.c_double)]


class NativeLibraryError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m diffsinger_amd.build_native` "
            "(hipcc --offload-arch=gfx950). diffsinger_amd has no CPU or eager fallback.")
    # torch ships its own libamdhip64.so.7; importing torch first makes the dynamic loader bind this
    # library to the SAME HIP runtime instance that owns torch's device memory and streams.
    import torch  # noqa: F401
    try:
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover
        raise NativeLibraryError(f"could not load {LIB_PATH}: {e}") from e
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.dsd_api_version.restype = C.c_int
    lib.dsd_create.argtypes = [C.POINTER(DsdConfig), C.POINTER(vp)]
    lib.dsd_destroy.argtypes = [vp]
    lib.dsd_destroy.restype = None
    lib.dsd_last_error.argtypes = [vp]
    lib.dsd_last_error.restype = C.c_char_p
    lib.dsd_load_weight.argtypes = [vp, C.c_char_p, vp, C.POINTER(i64), i32, i32]
    lib.dsd_finalize_weights.argtypes = [vp]
    lib.dsd_prepare_cond.argtypes = [vp, vp, i32, i32, i64, i64, i64, vp]
    lib.dsd_denoise.argtypes = [vp, vp, vp, i32, vp, vp]
    lib.dsd_sample.argtypes = [vp, C.POINTER(DsdProgram), vp, vp, vp, vp, vp, C.c_uint32, vp]
    lib.dsd_aux_decode.argtypes = [vp, vp, i32, i32, i64, i64, i64, vp, vp, vp, vp]
    lib.dsd_vocoder_create.argtypes = [C.POINTER(DsdVocoderConfig), C.POINTER(vp)]
    lib.dsd_vocode.argtypes = [vp, vp, i32, i32, i64, i64, i64, vp, vp, vp, vp, vp, vp]
    lib.dsd_encoder_create.argtypes = [C.POINTER(DsdEncoderConfig), C.POINTER(vp)]
    lib.dsd_token_encoder_create.argtypes = [C.POINTER(DsdTokenEncoderConfig), C.POINTER(vp)]
    lib.dsd_token_encode.argtypes = [vp, vp, vp, i32, i32, vp, vp]
    lib.dsd_predict_dur.argtypes = [vp, vp, vp, i32, i32, vp, vp]
    lib.dsd_cond_assemble.argtypes = [C.POINTER(DsdAssembleArgs), vp, vp]
    lib.dsd_set_lengths.argtypes = [vp, C.POINTER(C.c_int32), i32, vp]
    lib.dsd_encode.argtypes = [vp, vp, vp, vp, i32, i32, i32, C.POINTER(DsdEncodeExtras), vp, vp]
    lib.dsd_get_stats.argtypes = [vp, C.POINTER(DsdStats)]
    lib.dsd_kernel_timing.argtypes = [vp, i32]
    lib.dsd_set_precision.argtypes = [vp, i32]
    lib.dsd_kernel_timing_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i64)]
    lib.dsd_kernel_timing_classes.argtypes = [vp, C.POINTER(DsdKernelTime), i32, C.POINTER(i32), C.POINTER(C.c_double)]
    for name in EXPORTS:
        getattr(lib, name)
    if lib.dsd_api_version() != 10:
        raise NativeLibraryError("libdsdenoise.so API version mismatch")
    return lib


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _load()
    return _LIB


def check(handle, rc, what):
    if rc == 0:
        return
    msg = lib().dsd_last_error(handle).decode("utf-8", "replace")
    raise NativeLibraryError(f"{what} failed ({rc}): {msg}")


def program_to_c(prog):
    """schedule.Program -> (DsdProgram, keepalive)."""
    n = len(prog.evals)
    arr = (DsdEval * max(n, 1))()
    for i, ev in enumerate(prog.evals):
        ce = arr[i]
        ce.x_buf = ev.x_buf
        ce.t = float(ev.t)
        if not (1 <= len(ev.outs) <= DSD_MAX_OUT):
            raise ValueError(f"eval {i}: {len(ev.outs)} outputs")
        ce.n_out = len(ev.outs)
        for o, (dst, terms) in enumerate(ev.outs):
            if not (1 <= len(terms) <= DSD_MAX_TERMS):
                raise ValueError(f"eval {i} out {o}: {len(terms)} terms")
            ce.out[o].dst = dst
            ce.out[o].n_terms = len(terms)
            for k, (src, coef) in enumerate(terms):
                ce.out[o].terms[k].src = src
                ce.out[o].terms[k].coef = float(coef)
    p = DsdProgram()
    p.n_bufs = prog.n_bufs
    p.result_buf = prog.result_buf
    p.n_evals = n
    p.n_noise = prog.n_noise
    p.evals = C.cast(arr, C.POINTER(DsdEval))
    return p, arr
