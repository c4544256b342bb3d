"""ctypes binding of ``libmra_hip.so`` (the C ABI declared in ``include/mra.h``).

There is no CPU fallback: if the library is missing this raises, and every op in
``mraudio_amd`` goes through it.  ``torch`` is imported first on purpose: the PyTorch-ROCm wheel
ships its own ``libamdhip64.so.7`` and the HIP runtime must be shared with it so that the device
pointers and streams torch hands us are valid inside the kernels' launches.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (loads the HIP runtime the extension binds to)

MRA_F32, MRA_F16, MRA_BF16 = 0, 1, 2
_ERR = {-1: "MRA_EINVAL", -2: "MRA_ESTATE", -3: "MRA_EHIP", -4: "MRA_ENOMEM", -5: "MRA_ENAME"}

LIB_NAME = "libmra_hip.so"
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)


class mra_cfg(C.Structure):
    _fields_ = [
        ("hidden", C.c_int32), ("heads", C.c_int32), ("inter", C.c_int32), ("layers", C.c_int32),
        ("cross_freq", C.c_int32), ("enc_width", C.c_int32), ("n_query", C.c_int32), ("vocab", C.c_int32),
        ("max_pos", C.c_int32), ("ln_eps", C.c_float), ("enc_ln_eps", C.c_float), ("llm_hidden", C.c_int32),
        ("op_dtype", C.c_int32),
    ]


class mra_vit_cfg(C.Structure):
    _fields_ = [("dim", C.c_int32), ("heads", C.c_int32), ("mlp", C.c_int32), ("depth", C.c_int32), ("patch", C.c_int32),
                ("img", C.c_int32), ("ln_eps", C.c_float), ("op_dtype", C.c_int32), ("residual_dtype", C.c_int32)]


# name -> (restype, argtypes); must list every symbol include/mra.h declares (tests check this)
PROTOTYPES = {
    "mra_cfg_default": (None, [C.POINTER(mra_cfg), C.c_int32]),
    "mra_last_error": (C.c_char_p, []),
    "mra_version": (C.c_char_p, []),
    "mra_qformer_create": (C.c_int, [C.POINTER(mra_cfg), C.POINTER(C.c_void_p)]),
    "mra_qformer_destroy": (None, [C.c_void_p]),
    "mra_qformer_load": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.c_int32,
                                   C.c_void_p]),
    "mra_qformer_missing": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "mra_modality_ln": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                  C.c_void_p]),
    "mra_qformer_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "mra_qformer_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                      C.c_void_p]),
    "mra_qformer_pair_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "mra_qformer_forward_pair": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mra_qformer_set_kv_events": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mra_qformer_set_kv_done_event": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mra_qformer_set_cross_mode": (C.c_int, [C.c_void_p, C.c_int32]),
    "mra_qformer_set_cross_precision": (C.c_int, [C.c_void_p, C.c_int32]),
    "mra_qformer_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32]),
    "mra_qformer_prepare": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mra_kv_cache_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32]),
    "mra_kv_project": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "mra_llm_proj": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t,
                               C.c_void_p]),
    "mra_cosine_score": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "mra_fuse_logits": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_float), C.c_int32, C.c_int32, C.c_void_p,
                                  C.c_void_p]),
    "mra_span_from_logits": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    "mra_qformer_grad_bytes": (C.c_size_t, [C.c_void_p]),
    "mra_qformer_load_flat": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mra_qformer_grad_offset": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_int64)]),
    "mra_qformer_enable_training": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mra_qformer_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_float,
                                        C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_void_p]),
    "mra_qformer_train_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "mra_qformer_forward_train": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mra_qformer_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mra_qformer_flops": (C.c_double, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "mra_vit_cfg_default": (None, [C.POINTER(mra_vit_cfg)]),
    "mra_vit_create": (C.c_int, [C.POINTER(mra_vit_cfg), C.POINTER(C.c_void_p)]),
    "mra_vit_destroy": (None, [C.c_void_p]),
    "mra_vit_load": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.c_int32, C.c_void_p]),
    "mra_vit_missing": (C.c_int, [C.c_void_p]),
    "mra_vit_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int32]),
    "mra_vit_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mra_vit_flops": (C.c_double, [C.c_void_p, C.c_int32]),
    "mra_vit_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32]),
    "mra_debug_gemm_launches": (C.c_int64, [C.c_int32, C.c_int32]),
}

# GemmFamily / GemmEpi codes of mra_debug_gemm_launches (csrc/kernels.h)
GF_V1_64, GF_V1_128, GF_WS_256, GF_P8_256, GF_WS_128x384, GF_WS_176x384, GF_K128_64x128, GF_P8_TAIL, GF_P8_MIXED, GF_K128_64x64 = 0, 1, 3, 4, 5, 6, 7, 8, 9, 10
EPI_OP, EPI_GELU_OP, EPI_RES_F32, EPI_F32, EPI_KV, EPI_SOFTPART, EPI_RES_OP = 0, 1, 2, 3, 4, 5, 8
EPI_RES_F32_STAT, EPI_LNF_OP, EPI_LNF_GELU_OP, EPI_RES_OP_STAT = 10, 11, 12, 13     # the ViT's folded LayerNorms (csrc/kernels.h)


def gemm_launches(family: int, epi: int) -> int:
    return int(lib().mra_debug_gemm_launches(family, epi))

_lib = None


class MraError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """The loaded extension; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MraError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C mraudio_amd/csrc). "
                "mraudio_amd has no CPU fallback."
            )
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().mra_last_error().decode("utf-8", "replace")
        raise MraError(f"{what}: {_ERR.get(rc, rc)}: {msg}")


def ptr(t) -> C.c_void_p:
    """Device (or host) address of a tensor, None -> NULL."""
    return C.c_void_p(0 if t is None else t.data_ptr())


def current_stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def mra_dtype(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return MRA_F32
    if dt == torch.float16:
        return MRA_F16
    if dt == torch.bfloat16:
        return MRA_BF16
    raise MraError(f"unsupported dtype {dt}")


def torch_dtype(code: int) -> torch.dtype:
    return {MRA_F32: torch.float32, MRA_F16: torch.float16, MRA_BF16: torch.bfloat16}[code]
