"""ctypes binding of libssl4gie_hip.so (the C ABI declared in include/ssl4gie_hip.h).

The library is built in-tree by `ssl4gie_amd/csrc/Makefile` (see `__graft_entry__.build()`).  There
is NO fallback: if the shared object is missing or a call fails, we raise.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libssl4gie_hip.so")
# kernel-development builds only: `make -C ssl4gie_amd/csrc DEBUG_KNOBS=1` produces a second library with the
# ablation / time-stamp modes of the NT GEMM compiled in; it is loaded only on this explicit request (and says
# so on stderr) — the release library has no environment knob that changes results.
_dbg = os.environ.get("SSL4GIE_DEBUG_LIB", "")
if _dbg == "1" or _dbg.startswith("x"):  # "x<tag>": a tools/build_variant.sh experiment library
    LIB_PATH = os.path.join(_HERE, "libssl4gie_hip_dbg.so" if _dbg == "1" else f"libssl4gie_hip_{_dbg}.so")
    import sys as _sys
    print(f"ssl4gie_amd: SSL4GIE_DEBUG_LIB=1 -> loading the DEBUG library {LIB_PATH}", file=_sys.stderr)

# the one copy of the ABI revision on the Python side: build(), the tests and load() compare the
# library's ssl4gie_abi_version() with it (include/ssl4gie_hip.h documents the history)
ABI_VERSION = 7

PROF_KINDS = 7  # SSL4GIE_PROF_KINDS: entries of the launch profiler's arrays

F32, BF16 = 0, 1
BWD_ACCUMULATE, BWD_DEFER_WGRAD, BWD_NO_JOIN = 1, 2, 4
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESIDUAL, EPI_DGELU, EPI_BIAS_GELU_GRAD, EPI_MUL_AUX, \
    EPI_RELU_MASK_AUX, EPI_ADD_AUX, EPI_AFFINE_AUX_RELU = range(10)

vp, i32, i64, f32, sz = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_size_t


class Conv3x3Geom(C.Structure):
    """ssl4gie_conv3x3_geom: the channels-last map behind an implicit patch-matrix operand"""
    _fields_ = [("B", i32), ("H", i32), ("W", i32), ("C", i32), ("stride", i32), ("relu", i32)]


class GemmDesc(C.Structure):
    _fields_ = [
        ("M", i32), ("N", i32), ("K", i32), ("batch1", i32), ("batch2", i32),
        ("A", vp), ("sAm", i64), ("sAk", i64), ("sAb1", i64), ("sAb2", i64),
        ("B", vp), ("sBk", i64), ("sBn", i64), ("sBb1", i64), ("sBb2", i64),
        ("C", vp), ("ldc", i64), ("sCb1", i64), ("sCb2", i64),
        ("dtype_ab", i32), ("dtype_c", i32), ("alpha", f32), ("epilogue", i32),
        ("bias", vp), ("residual", vp), ("ldr", i64), ("aux", vp), ("out2", vp),
        ("accumulate", i32), ("colsum_a", vp), ("conv", C.POINTER(Conv3x3Geom)),
        ("colstats", vp), ("scale", vp), ("relu", i32),
    ]


_W_F32 = ("ln1_g", "ln1_b", "ln2_g", "ln2_b", "bqkv", "bproj", "bfc1", "bfc2")
_W_LP = ("wqkv", "wproj", "wfc1", "wfc2", "wqkv_t", "wproj_t", "wfc1_t", "wfc2_t")
_G_ALL = _W_F32 + ("wqkv", "wproj", "wfc1", "wfc2")
_ACT = ("mean1", "rstd1", "mean2", "rstd2", "h1", "qkv", "attn", "lse", "xmid", "h2", "u", "g")


class BlockWeights(C.Structure):
    _fields_ = [(n, vp) for n in _W_F32 + _W_LP]


class BlockGrads(C.Structure):
    _fields_ = [(n, vp) for n in _G_ALL]


class BlockAct(C.Structure):
    _fields_ = [(n, vp) for n in _ACT]


class BlockDims(C.Structure):
    _fields_ = [("B", i32), ("N", i32), ("D", i32), ("H", i32), ("F", i32), ("dtype", i32),
                ("eps", f32)]


# name -> (restype, argtypes); every symbol declared in include/ssl4gie_hip.h
PROTOTYPES = {
    "ssl4gie_abi_version": (i32, []),
    "ssl4gie_layernorm_fwd": (i32, [vp, vp, vp, vp, i32, vp, vp, i32, i32, f32, vp]),
    "ssl4gie_layernorm_bwd_workspace_bytes": (sz, [i32, i32]),
    "ssl4gie_layernorm_bwd": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, vp,
                                    i32, i32, vp]),
    "ssl4gie_colsum_workspace_bytes": (sz, [i32, i32]),
    "ssl4gie_colsum": (i32, [vp, i32, vp, i32, vp, i32, i32, i64, vp]),
    "ssl4gie_gemm_workspace_bytes": (sz, [C.POINTER(GemmDesc)]),
    "ssl4gie_gemm": (i32, [C.POINTER(GemmDesc), vp, sz, vp]),
    "ssl4gie_attn_workspace_bytes": (sz, [i32, i32, i32, i32, i32]),
    "ssl4gie_attn_fwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    "ssl4gie_attn_bwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    "ssl4gie_cast": (i32, [vp, vp, i32, i64, vp]),
    "ssl4gie_cast_transpose": (i32, [vp, vp, i32, i32, i32, vp]),
    "ssl4gie_cast_transpose_batch": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "ssl4gie_add_cast": (i32, [vp, vp, vp, vp, i32, i64, vp]),
    "ssl4gie_mask_argsort": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "ssl4gie_patch_gather": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i64, i32, vp]),
    "ssl4gie_tokens_assemble": (i32, [vp, i32, vp, vp, vp, i64, vp, i32, i32, i32, vp]),
    "ssl4gie_tokens_assemble_bwd": (i32, [vp, vp, i32, vp, i32, i32, i32, i32, vp]),
    "ssl4gie_decoder_assemble": (i32, [vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "ssl4gie_decoder_assemble_bwd_workspace_bytes": (sz, [i32, i32, i32]),
    "ssl4gie_decoder_assemble_bwd": (i32, [vp, vp, vp, i32, vp, i32, vp, i32, i32, i32, i32, vp]),
    "ssl4gie_mae_loss": (i32, [vp, vp, vp, vp, vp, vp, f32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "ssl4gie_im2col3x3": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i64, vp]),
    "ssl4gie_col2im3x3": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i64, vp]),
    "ssl4gie_conv3x3_direct_ok": (i32, [i32, i32, i32, i32, i32]),
    "ssl4gie_conv3x3_direct_tiles": (i32, [i32, i32, i32]),
    "ssl4gie_conv3x3_direct_fwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "ssl4gie_conv3x3_direct_wgrad_ok": (i32, [i32, i32, i32, i32, i32]),
    "ssl4gie_conv3x3_direct_wgrad_workspace_bytes": (sz, [i32, i32, i32, i32, i32]),
    "ssl4gie_conv3x3_direct_wgrad": (i32, [vp, vp, vp, vp, vp, sz, i32, i32, i32, i32, i32, i32, i32, vp]),
    "ssl4gie_conv3x3_direct_wgrad_affine": (i32, [vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, i32, i32, i32, i32, vp]),
    "ssl4gie_conv3x3_direct_fwd_affine": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "ssl4gie_stem7x7_packed_bytes": (sz, [i32, i32, i32]),
    "ssl4gie_stem7x7_pack": (i32, [vp, vp, i32, i32, i32, vp]),
    "ssl4gie_stem7x7_tiles": (i32, [i32, i32, i32]),
    "ssl4gie_stem7x7_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "ssl4gie_stem7x7_wgrad_workspace_bytes": (sz, [i32, i32, i32]),
    "ssl4gie_stem7x7_wgrad": (i32, [vp, vp, vp, vp, sz, i32, i32, i32, i32, vp]),
    "ssl4gie_bilinear2x_fwd": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "ssl4gie_bilinear2x_bwd": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "ssl4gie_pixel_shuffle": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "ssl4gie_pixel_unshuffle": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "ssl4gie_tokens_to_map": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "ssl4gie_map_to_tokens": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "ssl4gie_eltwise": (i32, [i32, vp, vp, vp, vp, i32, i64, vp]),
    "ssl4gie_depth_head_fwd": (i32, [vp, vp, vp, vp, i32, i64, i32, vp]),
    "ssl4gie_depth_head_bwd_workspace_bytes": (sz, [i64, i32]),
    "ssl4gie_depth_head_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, i64, i32, vp]),
    "ssl4gie_stem_im2col7x7": (i32, [vp, vp, i32, i32, i32, i32, i64, vp]),
    "ssl4gie_subsample2": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "ssl4gie_bn_workspace_bytes": (sz, [i64, i32]),
    "ssl4gie_bn_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, i32, i32, vp, i32, i64, i32, vp]),
    "ssl4gie_bn_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, i32, i64, i32, vp]),
    "ssl4gie_bn_stats": (i32, [vp, vp, vp, vp, i32, i64, i32, vp]),
    "ssl4gie_bn_fwd_partials_bits": (i32, [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, vp, i32, i64,
                                           i32, vp]),
    "ssl4gie_bn_bwd_bits": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, i64, i32, vp]),
    "ssl4gie_bn_bwd_xmask": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, i64, i32, vp]),
    "ssl4gie_bn_bwd_reduce_xmask": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i64, i32, vp]),
    "ssl4gie_bn_bwd_apply_xmask": (i32, [vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, i32, i64, i32, vp]),
    "ssl4gie_bn_bwd_reduce": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, i64, i32, vp]),
    "ssl4gie_bn_coef_stats": (i32, [vp, vp, vp, vp, vp, i32, vp]),
    "ssl4gie_bn_apply_bits": (i32, [vp, vp, vp, vp, vp, i32, i64, i32, vp]),
    "ssl4gie_bn_bwd_reduce_bits": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i64, i32, vp]),
    "ssl4gie_bn_bwd_apply": (i32, [vp, vp, vp, vp, vp, vp, vp, f32, vp, i32, vp, i32, i64, i32, vp]),
    "ssl4gie_ema_update": (i32, [vp, vp, f32, i64, vp]),
    "ssl4gie_maxpool3x3s2_fwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "ssl4gie_bn_maxpool3x3s2_fwd": (i32, [vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp]),
    "ssl4gie_maxpool3x3s2_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "ssl4gie_avgpool_fwd": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "ssl4gie_avgpool_bwd": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "ssl4gie_bn_coef_partials": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, f32, f32, vp, vp, i64, i32, vp]),
    "ssl4gie_bn_fwd_partials": (i32, [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, i32, vp, i32, i64,
                                      i32, vp]),
    "ssl4gie_bn_stats_partials": (i32, [vp, i32, vp, vp, vp, i64, i32, vp]),
    "ssl4gie_adamw_arena": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, f32, f32, f32, i32, i64, vp]),
    "ssl4gie_adamw_arena_lp": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, f32, f32, f32, i32, i64, vp, vp]),
    "ssl4gie_adamw_arena_range": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, f32, f32, f32, i32, i64, i64, vp, vp]),
    "ssl4gie_lars_workspace_bytes": (sz, [i32]),
    "ssl4gie_lars_arena": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, f32, f32, vp, i64, vp]),
    "ssl4gie_normalize_u8": (i32, [vp, vp, C.POINTER(C.c_float), C.POINTER(C.c_float), i32, i32, i32, vp]),
    "ssl4gie_maxpool2x2_fwd": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "ssl4gie_maxpool2x2_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "ssl4gie_gelu_map": (i32, [vp, vp, vp, i32, i64, vp]),
    "ssl4gie_map_layernorm_workspace_bytes": (sz, [i32]),
    "ssl4gie_map_layernorm_fwd": (i32, [vp, vp, vp, vp, vp, vp, f32, vp, i32, i32, i64, vp]),
    "ssl4gie_map_layernorm_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, i32, i64, vp]),
    "ssl4gie_gemm_tn_pair_workspace_bytes": (sz, [vp, vp]),
    "ssl4gie_gemm_tn_pair": (i32, [vp, vp, vp, sz, vp]),
    "ssl4gie_gemm_tn_group_workspace_bytes": (sz, [vp, i32]),
    "ssl4gie_gemm_tn_group": (i32, [vp, i32, vp, sz, vp]),
    "ssl4gie_block_wgrad_descs": (i32, [C.POINTER(BlockDims), C.POINTER(BlockAct), C.POINTER(BlockGrads), vp, vp,
                                        i32, vp]),
    "ssl4gie_wgrad_group": (i32, [vp, i32, vp, sz, i32, vp]),
    "ssl4gie_wgrad_wait": (i32, [i32, vp]),
    "ssl4gie_set_wgrad_stream": (i32, [i32]),
    "ssl4gie_set_compute_cus": (i32, [i32]),
    "ssl4gie_ssi_loss_workspace_bytes": (sz, [i32, i32, i32]),
    "ssl4gie_ssi_loss": (i32, [vp, vp, vp, vp, i32, i32, i32, f32, i32, vp, vp]),
    "ssl4gie_dice_loss_workspace_bytes": (sz, [i32]),
    "ssl4gie_dice_loss": (i32, [vp, vp, vp, vp, i32, i64, f32, vp, vp]),
    "ssl4gie_allreduce_direct_blob_bytes": (sz, []),
    "ssl4gie_allreduce_direct_init": (i32, [i32, i32, sz, vp, C.POINTER(vp)]),
    "ssl4gie_allreduce_direct_connect": (i32, [vp, vp]),
    "ssl4gie_allreduce_direct_enqueue": (i32, [vp, vp, sz, f32, vp]),
    "ssl4gie_allgather_direct_enqueue": (i32, [vp, vp, sz, vp, vp]),
    "ssl4gie_bn_combine_stats": (i32, [vp, i32, i32, f32, f32, vp, vp, vp, vp, vp, vp]),
    "ssl4gie_allreduce_direct_error": (C.c_uint, [vp]),
    "ssl4gie_allreduce_direct_set_timeout": (i32, [vp, C.c_double]),
    "ssl4gie_allreduce_direct_destroy": (i32, [vp]),
    "ssl4gie_conv3x3_weight_pack_batch": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "ssl4gie_conv3x3_weight_pack": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "ssl4gie_conv3x3_wgrad_unpack": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "ssl4gie_debug_nt256_stamps": (i32, [vp, sz]),
    "ssl4gie_prof_begin": (i32, [i32]),
    "ssl4gie_prof_collect": (i32, [vp, vp, vp]),
    "ssl4gie_prof_end": (i32, []),
    "ssl4gie_block_workspace_bytes": (sz, [C.POINTER(BlockDims)]),
    "ssl4gie_block_fwd": (i32, [C.POINTER(BlockDims), C.POINTER(BlockWeights),
                                C.POINTER(BlockAct), vp, vp, vp, vp]),
    "ssl4gie_block_bwd": (i32, [C.POINTER(BlockDims), C.POINTER(BlockWeights),
                                C.POINTER(BlockAct), C.POINTER(BlockGrads), vp, vp, vp, vp, vp,
                                i32, vp, vp]),
}

_lib = None


class HipExtensionMissing(RuntimeError):
    pass


def load():
    """Load the shared object (once).  Raises HipExtensionMissing if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipExtensionMissing(
            f"{LIB_PATH} not found: build it with `make -C ssl4gie_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "ssl4gie_amd has no CPU / eager fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.ssl4gie_abi_version()
    if got != ABI_VERSION:
        raise HipExtensionMissing(
            f"{LIB_PATH} has ABI revision {got}, this package binds revision {ABI_VERSION}: "
            "rebuild it (`make -C ssl4gie_amd/csrc`)")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        kind = {1000: "invalid argument",
                1001: "a peer of the direct all-reduce did not arrive in time (sticky)"}.get(rc, f"hipError_t {rc}")
        raise RuntimeError(f"libssl4gie_hip: {what} failed ({kind})")
