"""ctypes binding of libsea_hip.so (C ABI: include/sea_hip.h).

The library is the product's only compute path: if it is missing or cannot be loaded, every operator raises —
there is no CPU or eager-PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch  # imported first so that the HIP runtime torch ships is the one the library binds to

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsea_hip.so")

SEA_F32, SEA_BF16 = 0, 1
ABI_VERSION = 8   # include/sea_hip.h SEA_ABI_VERSION
MAX_GROUPS = 16
MAX_ATTN_PROBLEMS = 8
MAX_NORM_GROUPS = 16
MAX_SILU_GROUPS = 24

_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class SeaDropout(C.Structure):
    _fields_ = [("seed", C.c_uint32), ("stream", C.c_uint32), ("thr", _i32), ("mode", _i32)]


class SeaGemmGroup(C.Structure):
    _fields_ = [("A", _vp), ("W", _vp), ("bias", _vp), ("R", _vp), ("C32", _vp), ("Cact", _vp), ("Z", _vp),
                ("a_seg_stride", _i64),
                ("lda", _i32), ("ldw", _i32), ("ldr", _i32), ("ldc32", _i32), ("ldcact", _i32), ("ldz", _i32),
                ("M", _i32), ("N", _i32), ("K", _i32), ("n_seg", _i32),
                ("act", _i32), ("bias_scale", _f32), ("drop", SeaDropout),
                ("silu_c", _vp), ("silu_w1", _vp), ("silu_b1", _vp)]


class SeaQkvGroup(C.Structure):
    _fields_ = [("A", _vp), ("W", _vp), ("bias", _vp), ("Qout", _vp), ("Kout", _vp), ("Vtout", _vp), ("Vout", _vp),
                ("lda", _i32), ("ldw", _i32), ("M", _i32), ("N", _i32), ("K", _i32), ("col0", _i32)]


class SeaQkvCommon(C.Structure):
    _fields_ = [("rope", _vp), ("H", _i32), ("hd", _i32), ("T", _i32), ("pos0", _i32), ("cap", _i32), ("q_scale", _f32)]


class SeaAttnProblem(C.Structure):
    _fields_ = [("Q", _vp), ("K", _vp), ("Vt", _vp), ("O", _vp), ("LSE", _vp)]


class SeaAttnParams(C.Structure):
    _fields_ = [("p", SeaAttnProblem * MAX_ATTN_PROBLEMS), ("n_problems", _i32),
                ("B", _i32), ("H", _i32), ("hd", _i32), ("Tq", _i32), ("Tk", _i32), ("cap", _i32),
                ("q_pos0", _i32), ("src_len", _i32), ("ldo", _i32), ("drop", SeaDropout)]


class SeaNormGroup(C.Structure):
    _fields_ = [("X", _vp), ("mod", _vp), ("gamma", _vp), ("beta", _vp), ("Y32", _vp), ("Yact", _vp),
                ("mean", _vp), ("rstd", _vp),
                ("ldx", _i32), ("ldmod", _i32), ("ldy32", _i32), ("ldyact", _i32),
                ("addend", _vp), ("Xout", _vp), ("ldadd", _i32), ("ldxout", _i32)]


class SeaSiluGroup(C.Structure):
    _fields_ = [("w1", _vp), ("b1", _vp), ("Hid", _vp), ("K2", _i32), ("ld", _i32)]


class SeaIbParams(C.Structure):
    _fields_ = [("X", _vp * 8), ("n_fields", _i32), ("ldx", _i32),
                ("c", _vp), ("w1", _vp), ("b1", _vp), ("lnw", _vp), ("lnb", _vp), ("w2", _vp), ("b2", _vp),
                ("M", _i32), ("E", _i32), ("h", _i32), ("drop", SeaDropout), ("mode", _i32), ("pad_", _i32)]


class SeaWgradGroup(C.Structure):
    _fields_ = [("dY", _vp), ("X", _vp), ("dW", _vp), ("db", _vp), ("lddy", _i32), ("ldx", _i32), ("lddw", _i32),
                ("M", _i32), ("N", _i32), ("K", _i32), ("overwrite", _i32), ("pad_", _i32)]


class SeaNormBwdGroup(C.Structure):
    _fields_ = [("dY", _vp), ("X", _vp), ("mod", _vp), ("gamma", _vp), ("beta", _vp), ("mean", _vp), ("rstd", _vp),
                ("dX32", _vp), ("dXact", _vp), ("dmod", _vp), ("dgamma", _vp), ("dbeta", _vp),
                ("lddy", _i32), ("ldx", _i32), ("ldmod", _i32), ("lddx32", _i32), ("lddxact", _i32), ("lddmod", _i32)]


class SeaSiluBwdGroup(C.Structure):
    _fields_ = [("dHid", _vp), ("w1", _vp), ("b1", _vp), ("dw1", _vp), ("db1", _vp), ("K2", _i32), ("ld", _i32)]


class SeaIbBwdParams(C.Structure):
    _fields_ = [("dX", _vp * 8), ("n_fields", _i32), ("ldx", _i32),
                ("c", _vp), ("w1", _vp), ("b1", _vp), ("lnw", _vp), ("lnb", _vp), ("w2", _vp),
                ("dw1", _vp), ("db1", _vp), ("dlnw", _vp), ("dlnb", _vp), ("dw2", _vp), ("db2", _vp),
                ("M", _i32), ("E", _i32), ("h", _i32), ("drop", SeaDropout), ("mode", _i32), ("pad_", _i32),
                ("ws", _vp), ("ws_floats", _i64), ("dhid", _vp)]


class SeaAttnBwdProblem(C.Structure):
    _fields_ = [("Q", _vp), ("K", _vp), ("V", _vp), ("O", _vp), ("dO", _vp), ("LSE", _vp), ("delta", _vp), ("dQ", _vp), ("dK", _vp),
                ("dV", _vp)]


class SeaAttnBwdParams(C.Structure):
    _fields_ = [("p", SeaAttnBwdProblem * MAX_ATTN_PROBLEMS), ("rope", _vp), ("n_problems", _i32),
                ("B", _i32), ("H", _i32), ("hd", _i32), ("Tq", _i32), ("Tk", _i32), ("cap", _i32), ("q_pos0", _i32), ("src_len", _i32),
                ("ldo", _i32), ("lddo", _i32), ("lddq", _i32), ("lddk", _i32), ("lddv", _i32), ("q_scale", _f32), ("drop", SeaDropout)]


OP_GEMM, OP_QKV, OP_ATTN, OP_NORM, OP_SILU, OP_IB, OP_CONVERT, OP_GEMM_NORM, OP_XTAIL, OP_MLP1, OP_MLP2, OP_GEMM_FEW, OP_QKV_FEW, OP_CHAIN, OP_ADALN, OP_MLPB, OP_AQKV, OP_SPLITK = 1, 2, 3, 4, 5, 6, 8, 9, 10, 11, 13, 14, 15, 16, 17, 18, 19, 20
FEW_MAX_GROUPS = 8     # sea_gemm_fewrows / sea_qkv_rope_fewrows (gemv.hip)
FEW_K = (512, 1024, 2048, 4096, 8192, 16384)


class SeaLaunchRec(C.Structure):
    _fields_ = [("op", _i32), ("n", _i32), ("dtype", _i32), ("i0", _i32), ("i1", _i32), ("i2", _i32), ("i3", _i32), ("f0", _f32),
                ("p0", _vp), ("p1", _vp), ("l0", _i64), ("l1", _i64), ("l2", _i64), ("l3", _i64)]

MAX_GEMM_NORM_GROUPS = 8


class SeaGemmNormGroup(C.Structure):
    _fields_ = [("A", _vp), ("W", _vp), ("bias", _vp), ("R", _vp), ("C32", _vp), ("mod", _vp), ("gamma", _vp), ("beta", _vp),
                ("Y32", _vp), ("Yact", _vp), ("mean", _vp), ("rstd", _vp),
                ("lda", _i32), ("ldw", _i32), ("ldr", _i32), ("ldc32", _i32), ("ldmod", _i32), ("ldy32", _i32), ("ldyact", _i32),
                ("M", _i32), ("N", _i32), ("K", _i32),
                ("n_seg", _i32), ("a_seg_stride", _i64), ("bias_scale", _f32), ("ldcact", _i32), ("Cact", _vp),
                ("ib_c", _vp), ("ib_w1", _vp), ("ib_b1", _vp), ("ib_lnw", _vp), ("ib_lnb", _vp), ("ib_w2", _vp), ("ib_b2", _vp),
                ("ib_h", _i32), ("pad_", _i32)]


XTAIL_MAX_SEG = 4
XTAIL_MAX_GROUPS = 4
MAX_SILU_IB = 4


class SeaExchangeTail(C.Structure):
    _fields_ = [("att", _vp * XTAIL_MAX_SEG), ("Wp", _vp * XTAIL_MAX_SEG), ("Wup", _vp), ("bup", _vp), ("X", _vp), ("Xact", _vp),
                ("n_seg", _i32), ("ldatt", _i32), ("ldwp", _i32), ("ldwup", _i32), ("ldx", _i32), ("ldxact", _i32),
                ("M", _i32), ("D", _i32), ("E", _i32), ("has_down", _i32), ("bias_scale", _f32),
                ("down", SeaGemmNormGroup)]


CHAIN_MAX_PROJ = 6
CHAIN_MAX_GROUPS = 3
CHAIN_MAX_RIDERS = 8


class SeaRowChain(C.Structure):
    _fields_ = [("att", _vp * XTAIL_MAX_SEG), ("Wp", _vp * XTAIL_MAX_SEG), ("a2", _vp), ("W2", _vp), ("b2", _vp), ("Xin", _vp), ("X", _vp), ("Xact", _vp),
                ("n_seg", _i32), ("ldatt", _i32), ("ldwp", _i32), ("lda2", _i32), ("ldw2", _i32), ("ldxin", _i32), ("ldx", _i32), ("ldxact", _i32),
                ("M", _i32), ("D", _i32), ("E", _i32), ("has_down", _i32), ("n_proj", _i32), ("bias_scale", _f32),
                ("down", SeaGemmNormGroup), ("proj", SeaQkvGroup * CHAIN_MAX_PROJ)]


MAX_ADALN_GROUPS = 16


class SeaAdalnGroup(C.Structure):
    _fields_ = [("A", _vp), ("W", _vp), ("bias", _vp), ("X", _vp), ("gamma", _vp), ("beta", _vp), ("Yact", _vp), ("Y32", _vp), ("mean", _vp), ("rstd", _vp),
                ("lda", _i32), ("ldw", _i32), ("ldx", _i32), ("ldyact", _i32), ("ldy32", _i32), ("M", _i32), ("d", _i32), ("K", _i32)]


MAX_SPLITK_GROUPS = 8


class SeaSplitkGroup(C.Structure):
    _fields_ = [("P", _vp), ("bias", _vp), ("R", _vp), ("C32", _vp), ("Cact", _vp), ("p_stride", _i64), ("S", _i32), ("M", _i32), ("N", _i32), ("ldp", _i32),
                ("ldr", _i32), ("ldc32", _i32), ("ldcact", _i32), ("bias_scale", _f32)]


MAX_AQKV_GROUPS = 4
AQKV_MAX_SILU = 8


class SeaAdalnQkv(C.Structure):
    _fields_ = [("X", _vp), ("cond", _vp), ("w1", _vp), ("b1", _vp), ("W2c", _vp), ("b2c", _vp), ("gamma", _vp), ("beta", _vp), ("Wqkv", _vp), ("bqkv", _vp),
                ("Q", _vp), ("K", _vp), ("Vt", _vp), ("ldx", _i32), ("ldw2c", _i32), ("ldw", _i32), ("M", _i32), ("E", _i32), ("N3", _i32),
                ("w13", _vp), ("b13", _vp), ("W3", _vp), ("b3", _vp), ("mod3", _vp), ("ldw3", _i32), ("ldmod3", _i32)]


MAX_MLP_GROUPS = 8


class SeaMlpGroup(C.Structure):
    _fields_ = [("A", _vp), ("W1", _vp), ("b1", _vp), ("lnw", _vp), ("lnb", _vp), ("Hg", _vp),
                ("lda", _i32), ("ldw", _i32), ("ldh", _i32), ("M", _i32), ("E", _i32), ("S", _i32),
                ("X32", _vp), ("addend", _vp), ("Xout", _vp), ("mod", _vp), ("gamma", _vp), ("beta", _vp),
                ("ldx32", _i32), ("ldadd", _i32), ("ldxout", _i32), ("ldmod", _i32), ("norm_eps", _f32), ("pad_", _i32)]


class SeaMlp2Group(C.Structure):
    _fields_ = [("Hg", _vp), ("W2", _vp), ("b2", _vp), ("R", _vp), ("Wproj", _vp), ("bproj", _vp), ("gamma", _vp), ("beta", _vp), ("mod", _vp),
                ("Y32", _vp), ("Yact", _vp),
                ("ldh", _i32), ("ldw2", _i32), ("ldr", _i32), ("ldwp", _i32), ("ldmod", _i32), ("ldy32", _i32), ("ldyact", _i32),
                ("M", _i32), ("E", _i32), ("S", _i32)]


class SeaStepPatch(C.Structure):
    _fields_ = [("addr", _vp), ("kind", _i32), ("pad_", _i32), ("base", C.c_int64), ("stride", C.c_int64)]


KV_MAX_FIELDS = 4


class SeaKvNorm(C.Structure):
    _fields_ = [("gamma", _vp), ("beta", _vp), ("mod", _vp), ("ldmod", _i32), ("pad_", _i32)]


class SeaKvField(C.Structure):
    _fields_ = [("ln0", SeaKvNorm), ("ln_cross", SeaKvNorm), ("ln2", SeaKvNorm),
                ("Wqkv", _vp), ("bqkv", _vp), ("Wo", _vp), ("Wdown", _vp), ("bdown", _vp), ("Wup", _vp), ("bup", _vp),
                ("W1", _vp), ("b1", _vp), ("lnw", _vp), ("lnb", _vp), ("W2", _vp), ("b2", _vp), ("Wproj", _vp), ("bproj", _vp),
                ("Ks", _vp), ("Vs", _vp)]


class SeaKvPair(C.Structure):
    _fields_ = [("Wq", _vp), ("bq", _vp), ("Wkv", _vp), ("bkv", _vp), ("Wp", _vp), ("Kc", _vp), ("Vc", _vp)]


class SeaKvLayer(C.Structure):
    _fields_ = [("f", SeaKvField * KV_MAX_FIELDS), ("p", (SeaKvPair * KV_MAX_FIELDS) * KV_MAX_FIELDS), ("ib", _vp)]


class SeaKvGlobal(C.Structure):
    _fields_ = [("F", _i32), ("E", _i32), ("D", _i32), ("S", _i32), ("H", _i32), ("B", _i32), ("L", _i32), ("cap", _i32),
                ("exchange", _i32), ("ib_after_cross", _i32), ("final_ln", SeaKvNorm * KV_MAX_FIELDS),
                ("rope_self", _vp), ("rope_cross", _vp), ("traj", _vp), ("xl", _vp * 2),
                ("att_e", _vp), ("xr", _vp), ("xq", _vp), ("x3", _vp), ("hbuf", _vp), ("nd_old", _vp), ("oc", _vp), ("qc", _vp), ("ml", _vp),
                ("handoff", _vp), ("err", _vp), ("handoff_words", _i64)]


MAX_WGRAD_GROUPS = 32
MAX_NORM_BWD_GROUPS = 8
MAX_SILU_BWD_GROUPS = 24

_lib: Optional[C.CDLL] = None


class NativeLibraryError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raise loudly if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m sea_amd.build` (hipcc, gfx950). "
            "sea_amd has no CPU or eager fallback.")
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    L.sea_abi_version.restype = C.c_int
    L.sea_last_error.restype = C.c_char_p
    L.sea_struct_sizes.argtypes = [C.POINTER(C.c_int), C.c_int]
    L.sea_struct_sizes.restype = C.c_int
    L.sea_device_info.argtypes = [C.POINTER(C.c_int), C.c_char_p, C.c_int]
    L.sea_gemm_grouped.argtypes = [C.POINTER(SeaGemmGroup), C.c_int, C.c_int, _vp]
    L.sea_qkv_rope_grouped.argtypes = [C.POINTER(SeaQkvGroup), C.c_int, C.POINTER(SeaQkvCommon), C.c_int, _vp]
    L.sea_attention_fwd.argtypes = [C.POINTER(SeaAttnParams), C.c_int, _vp]
    L.sea_rownorm.argtypes = [C.POINTER(SeaNormGroup), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _vp]
    L.sea_gemm_fewrows.argtypes = [C.POINTER(SeaGemmGroup), C.POINTER(SeaNormGroup), C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _vp]
    L.sea_gemm_fewrows.restype = C.c_int
    L.sea_qkv_rope_fewrows.argtypes = [C.POINTER(SeaQkvGroup), C.POINTER(SeaNormGroup), C.c_int, C.POINTER(SeaQkvCommon), C.c_float, C.c_int, _vp]
    L.sea_qkv_rope_fewrows.restype = C.c_int
    L.sea_silu_outer.argtypes = [C.POINTER(SeaSiluGroup), C.c_int, _vp, C.c_int, C.c_int, _vp]
    L.sea_ib_add.argtypes = [C.POINTER(SeaIbParams), _vp]
    L.sea_convert_f32_to_act.argtypes = [_vp, _i64, _vp, _i64, _i64, _i64, C.c_int, _vp]
    L.sea_selftest_mfma.restype = C.c_int
    L.sea_wgrad_grouped.argtypes = [C.POINTER(SeaWgradGroup), C.c_int, C.c_int, _vp]
    L.sea_transpose_weights.argtypes = [_vp, _vp, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp]
    L.sea_rownorm_bwd.argtypes = [C.POINTER(SeaNormBwdGroup), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _i64, _vp]
    L.sea_silu_outer_bwd.argtypes = [C.POINTER(SeaSiluBwdGroup), C.c_int, _vp, C.c_int, C.c_int, _vp, _i64, _vp]
    L.sea_ib_bwd.argtypes = [C.POINTER(SeaIbBwdParams), _vp]
    L.sea_attention_bwd.argtypes = [C.POINTER(SeaAttnBwdParams), C.c_int, _vp]
    L.sea_dropout_mask.argtypes = [_vp, _i64, _i64, C.c_uint32, C.c_uint32, _i32, _vp]
    L.sea_dropout_mask.restype = C.c_int
    L.sea_unpatchify.argtypes = [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]
    L.sea_unpatchify.restype = C.c_int
    L.sea_gemm_rownorm.argtypes = [C.POINTER(SeaGemmNormGroup), C.c_int, C.c_float, C.c_int, _vp]
    L.sea_gemm_rownorm.restype = C.c_int
    L.sea_exchange_tail.argtypes = [C.POINTER(SeaExchangeTail), C.c_int, C.c_float, C.c_int, _vp]
    L.sea_exchange_tail.restype = C.c_int
    L.sea_gemm_adaln.argtypes = [C.POINTER(SeaAdalnGroup), C.c_int, C.c_float, C.c_int, _vp]
    L.sea_gemm_adaln.restype = C.c_int
    L.sea_row_chain.argtypes = [C.POINTER(SeaRowChain), C.c_int, C.POINTER(SeaQkvCommon), C.c_float, C.c_int, _vp]
    L.sea_row_chain.restype = C.c_int
    L.sea_row_chain_riders.argtypes = [C.POINTER(SeaRowChain), C.c_int, C.POINTER(SeaQkvCommon), C.POINTER(SeaGemmGroup), C.c_int, C.c_int, C.c_int, C.POINTER(SeaIbParams), C.c_float, C.c_int, _vp]
    L.sea_row_chain_riders.restype = C.c_int
    L.sea_patchify.argtypes = [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _vp]
    L.sea_patchify.restype = C.c_int
    L.sea_silu_outer_ib.argtypes = [C.POINTER(SeaSiluGroup), C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp]
    L.sea_silu_outer_ib.restype = C.c_int
    L.sea_mlp_fc1_ln_gelu.argtypes = [C.POINTER(SeaMlpGroup), C.c_int, C.c_float, C.c_int, _vp]
    L.sea_mlp_fc1_ln_gelu.restype = C.c_int
    L.sea_mlp_fc2_proj_norm.argtypes = [C.POINTER(SeaMlp2Group), C.c_int, C.c_float, C.c_int, _vp]
    L.sea_mlp_fc2_proj_norm.restype = C.c_int
    L.sea_mlp_block.argtypes = [C.POINTER(SeaMlpGroup), C.POINTER(SeaMlp2Group), C.c_int, C.c_float, C.c_int, _vp]
    L.sea_mlp_block.restype = C.c_int
    L.sea_adaln_qkv.argtypes = [C.POINTER(SeaAdalnQkv), C.c_int, C.POINTER(SeaQkvCommon), C.POINTER(SeaGemmGroup), C.c_int, C.POINTER(SeaSiluGroup), C.c_int, _vp, C.c_int,
                                C.POINTER(SeaIbParams), C.c_float, C.c_int, _vp]
    L.sea_adaln_qkv.restype = C.c_int
    L.sea_splitk_finish.argtypes = [C.POINTER(SeaSplitkGroup), C.c_int, C.c_int, _vp]
    L.sea_splitk_finish.restype = C.c_int
    L.sea_run_list.argtypes = [C.POINTER(SeaLaunchRec), C.c_int, _vp]
    L.sea_run_list.restype = C.c_int
    L.sea_run_list_steps.argtypes = [C.POINTER(SeaLaunchRec), C.c_int, C.POINTER(SeaStepPatch), C.c_int, C.c_int, C.c_int, _vp]
    L.sea_run_list_steps.restype = C.c_int
    L.sea_kv_rollout.argtypes = [C.POINTER(SeaKvGlobal), C.POINTER(SeaKvLayer), C.c_int, C.c_int, C.c_uint32, C.c_int, _vp]
    L.sea_kv_rollout.restype = C.c_int
    L.sea_kv_arena_words.argtypes = [C.POINTER(SeaKvGlobal)]
    L.sea_kv_arena_words.restype = C.c_int64
    L.sea_kv_debug_stamps.argtypes = [_vp]
    L.sea_kv_debug_stamps.restype = None
    for name in ("sea_attention_bwd", "sea_wgrad_grouped", "sea_transpose_weights", "sea_rownorm_bwd", "sea_silu_outer_bwd", "sea_ib_bwd"):
        getattr(L, name).restype = C.c_int
    L.sea_mse_fwd_bwd.argtypes = [_vp, _vp, _vp, _vp, _vp, C.c_int, _i64, C.c_float, _vp]
    L.sea_relative_mse.argtypes = [_vp, _vp, _vp, _i64, C.c_int, _vp]
    L.sea_adamw_flat.argtypes = [_vp, _vp, _vp, _vp, _vp, C.c_int, _i64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                 C.c_int, C.c_float, _vp]
    for name in ("sea_gemm_grouped", "sea_qkv_rope_grouped", "sea_attention_fwd", "sea_rownorm", "sea_silu_outer",
                 "sea_ib_add", "sea_convert_f32_to_act", "sea_device_info", "sea_mse_fwd_bwd", "sea_relative_mse",
                 "sea_adamw_flat"):
        getattr(L, name).restype = C.c_int
    if L.sea_abi_version() != ABI_VERSION:
        raise NativeLibraryError(f"{LIB_PATH}: ABI version {L.sea_abi_version()} != {ABI_VERSION}; rebuild (python -m sea_amd.build --force)")
    _lib = L
    return L


ABI_STRUCTS = (SeaGemmGroup, SeaQkvGroup, SeaQkvCommon, SeaAttnProblem, SeaAttnParams, SeaNormGroup, SeaSiluGroup,
               SeaIbParams, SeaWgradGroup, SeaNormBwdGroup, SeaSiluBwdGroup, SeaIbBwdParams, SeaAttnBwdProblem, SeaAttnBwdParams,
               SeaDropout, SeaLaunchRec, SeaGemmNormGroup, SeaExchangeTail, SeaMlpGroup, SeaMlp2Group, SeaKvNorm, SeaKvField, SeaKvPair, SeaKvLayer, SeaKvGlobal, SeaStepPatch, SeaRowChain, SeaAdalnGroup, SeaAdalnQkv, SeaSplitkGroup)

EXPORTED_SYMBOLS = (
    "sea_abi_version", "sea_last_error", "sea_struct_sizes", "sea_device_info", "sea_gemm_grouped", "sea_qkv_rope_grouped",
    "sea_attention_fwd", "sea_rownorm", "sea_silu_outer", "sea_ib_add", "sea_convert_f32_to_act", "sea_selftest_mfma",
    "sea_mse_fwd_bwd", "sea_relative_mse", "sea_adamw_flat",
    "sea_wgrad_grouped", "sea_transpose_weights", "sea_rownorm_bwd", "sea_silu_outer_bwd", "sea_ib_bwd",
    "sea_attention_bwd", "sea_dropout_mask", "sea_run_list", "sea_run_list_steps", "sea_unpatchify", "sea_gemm_rownorm", "sea_exchange_tail", "sea_patchify", "sea_silu_outer_ib", "sea_mlp_fc1_ln_gelu", "sea_mlp_fc2_proj_norm", "sea_kv_rollout", "sea_kv_arena_words", "sea_kv_debug_stamps",
    "sea_gemm_fewrows", "sea_qkv_rope_fewrows", "sea_row_chain", "sea_row_chain_riders", "sea_gemm_adaln", "sea_mlp_block", "sea_adaln_qkv", "sea_splitk_finish",
)


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().sea_last_error()
        raise RuntimeError(f"libsea_hip {what} failed (code {rc}): {msg.decode() if msg else '?'}")


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return SEA_F32
    if dt == torch.bfloat16:
        return SEA_BF16
    raise ValueError(f"sea_amd: activation dtype must be float32 or bfloat16, got {dt}")


def require_gpu(t: torch.Tensor, name: str = "tensor") -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"sea_amd: {name} is on {t.device}; this path runs only on an MI355X through libsea_hip.so "
            "(no CPU fallback — the CPU oracle lives under oracle/ for tests only)")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()
