"""ctypes binding of libwae_hip.so (the C ABI declared in include/wae.h).

There is no CPU fallback: if the shared object is missing or a call fails, this raises.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WAE_LIB_PATH") or os.path.join(_HERE, "libwae_hip.so")   # WAE_LIB_PATH: A/B of two builds (tools/)

WAE_F32, WAE_BF16, WAE_F16 = 0, 1, 2
GLU_SAVE_Z, GLU_NO_OUT, GLU_WAVES4, GLU_CG2, GLU_PAIR, GLU_GENERIC = 2, 4, 8, 16, 32, 64
TM_INTERLEAVE, TM_ONE_WG = 1, 2
ERR_CLASS_ID, ERR_SPEAKER_ID, ERR_TARGET_ID, ERR_NOT_ONEHOT = 1, 2, 4, 8

c_i32, c_i64, c_f32, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p


class GluDesc(ctypes.Structure):
    _fields_ = [(n, c_i32) for n in ("dtype", "B", "T", "Rp", "Ccp", "Hp", "ktaps", "dilation", "flags")]


class ArDesc(ctypes.Structure):
    _fields_ = [(n, c_i32) for n in ("dtype", "B", "T", "L", "R", "Rp", "G", "Hp", "S", "O", "Cc", "Ccp", "ktaps", "mode",
                                     "init_idx", "scalar_input")] + [("scale", c_f32), ("n_forced", c_i32), ("coop_generic", c_i32),
                                                                       ("resident_lds", c_i32), ("resident_regs", c_i32)]


class TmDesc(ctypes.Structure):
    _fields_ = [(n, c_i32) for n in ("dtype", "B", "T", "M", "nsrc", "mode")] + [("alpha", c_f32), ("flags", c_i32)]


class TmCe(ctypes.Structure):            # include/wae.h: wae_tm_ce (modes 5 / 6 of the wide head)
    _fields_ = [("logits", c_vp), ("target", c_vp), ("nll", c_vp), ("lse", c_vp), ("lengths", c_vp), ("inv_count", c_f32),
                ("O", c_i32)]


class TnTile(ctypes.Structure):          # include/wae.h: wae_tn_tile (device array element)
    _fields_ = [("P", c_vp), ("Q", c_vp), ("onehot", c_vp), ("C", c_vp), ("p_stride", c_i64), ("q_stride", c_i64), ("ldc", c_i64),
                ("m_valid", c_i32), ("n_valid", c_i32), ("m0", c_i32), ("shift", c_i32), ("ones_col", c_i32), ("alpha", c_f32)]


class GluBwdDesc(ctypes.Structure):      # include/wae.h: wae_glu_bwd_desc
    _fields_ = [(n, c_i32) for n in ("dtype", "B", "T", "Rp", "Hp", "Sp", "ktaps", "dilation")] + [("alpha", c_f32)]


class TsJob(ctypes.Structure):           # include/wae.h: wae_ts_job (device array element)
    _fields_ = [("P", c_vp), ("Q", c_vp), ("C", c_vp), ("p_stride", c_i64), ("q_stride", c_i64), ("ldc", c_i64),
                ("m_valid", c_i32), ("n_valid", c_i32), ("shift", c_i32), ("ones_col", c_i32), ("alpha", c_f32), ("pad_", c_i32)]


class TqJob(ctypes.Structure):           # include/wae.h: wae_tq_job (device array element of wae_gemm_tn_static)
    _fields_ = [("P", c_vp), ("Q0", c_vp), ("Q1", c_vp), ("C0", c_vp), ("C1", c_vp), ("Cb", c_vp), ("p_stride", c_i64),
                ("q0_stride", c_i64), ("q1_stride", c_i64), ("ldc0", c_i64), ("ldc1", c_i64), ("m_valid", c_i32), ("n0_valid", c_i32),
                ("n1_valid", c_i32), ("shift", c_i32), ("ones_col", c_i32), ("kind", c_i32), ("alpha", c_f32), ("pad_", c_i32)]


TQ_TAPS, TQ_COND, TQ_OUTSKIP = 0, 1, 2


class TsSeg(ctypes.Structure):           # include/wae.h: wae_ts_seg
    _fields_ = [("job", c_i32), ("slab_begin", c_i32), ("slab_end", c_i32)]


class GatherJob(ctypes.Structure):       # include/wae.h: wae_gather_job (host array element)
    _fields_ = [("src", c_vp), ("map", c_vp), ("dst", c_vp), ("n", c_i64), ("src_stride", c_i64), ("dst_stride", c_i64),
                ("nbatch", c_i32), ("dtype", c_i32)]


class ScatterJob(ctypes.Structure):      # include/wae.h: wae_scatter_job (host array element)
    _fields_ = [("src", c_vp), ("map", c_vp), ("dst", c_vp), ("n", c_i64), ("src_stride", c_i64), ("dst_stride", c_i64),
                ("src_ld", c_i64), ("nbatch", c_i32), ("src_cols", c_i32), ("unique", c_i32), ("pad_", c_i32)]


MULTI_MAX = 16


class HeadDesc(ctypes.Structure):
    _fields_ = [(n, c_i32) for n in ("dtype", "B", "T", "Ku", "Sp", "Op", "O")] + [("scale", c_f32)]


# name -> (restype, argtypes); mirrors include/wae.h one to one
SIGNATURES = {
    "wae_version": (ctypes.c_char_p, []),
    "wae_last_error": (ctypes.c_char_p, []),
    "wae_weight_norm_fwd": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "wae_weight_norm_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "wae_weight_norm_bwd_range": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "wae_pack_gather": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i64, c_i64, c_i32, c_vp]),
    "wae_unpack_scatter_add": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i64, c_i64, c_i32, c_i64, c_i32, c_vp]),
    "wae_pack_gather_multi": (c_i32, [ctypes.POINTER(GatherJob), c_i32, c_vp]),
    "wae_unpack_scatter_add_multi": (c_i32, [ctypes.POINTER(ScatterJob), c_i32, c_vp]),
    "wae_enc_conv_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp] + [c_i32] * 9 + [c_vp]),
    "wae_vq_nearest": (c_i32, [c_vp] * 6 + [c_i32] * 4 + [c_f32, c_vp]),
    "wae_upsample_stage_fwd": (c_i32, [c_vp, c_vp, c_vp] + [c_i32] * 7 + [c_vp]),
    "wae_gproj_fwd": (c_i32, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp] + [c_i32] * 6 + [c_vp, c_vp]),
    "wae_gproj_bwd": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i64] + [c_i32] * 7 + [c_vp]),
    "wae_check_ids": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_i32, c_vp]),
    "wae_onehot_to_ids": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i64, c_i64, c_i64, c_vp, c_vp, c_i32, c_vp]),
    "wae_stream_delay": (c_i32, [ctypes.c_double, c_vp]),
    "wae_upsample_stage_bwd": (c_i32, [c_vp] * 5 + [c_i32] * 4 + [c_vp]),
    "wae_enc_conv_bwd": (c_i32, [c_vp] * 7 + [c_i32] * 9 + [c_vp]),
    "wae_vq_bwd": (c_i32, [c_vp] * 6 + [c_i32] * 3 + [c_f32, c_f32, c_vp]),
    "wae_vq_slice": (c_i32, [c_vp] * 6 + [c_i32] * 6 + [c_f32, c_i32, c_vp]),
    "wae_vq_ema_update": (c_i32, [c_vp] * 6 + [c_i32] * 6 + [c_f32, c_vp]),
    "wae_vq_slice_bwd": (c_i32, [c_vp] * 6 + [c_i32] * 5 + [c_f32, c_f32, c_vp]),
    "wae_first_conv_fwd": (c_i32, [c_vp] * 5 + [c_i64, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "wae_onehot_rows": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp]),
    "wae_softmax_bct_fwd": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "wae_softmax_bct_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "wae_bmm_f32": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, ctypes.c_float, c_vp]),
    "wae_glu_layer_fwd": (c_i32, [ctypes.POINTER(GluDesc), c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "wae_glu_layer_fwd_drop": (c_i32, [ctypes.POINTER(GluDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "wae_dropout_fwd": (c_i32, [c_vp, c_vp, c_i64, ctypes.c_uint64, c_f32, c_i32, c_vp]),
    "wae_dropout_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, ctypes.c_uint64, c_f32, c_f32, c_i32, c_vp]),
    "wae_glu_packed_bytes": (c_i64, [ctypes.POINTER(GluDesc)]),
    "wae_head_fwd": (c_i32, [ctypes.POINTER(HeadDesc)] + [c_vp] * 10),
    "wae_head_fwd_from_h0": (c_i32, [ctypes.POINTER(HeadDesc)] + [c_vp] * 9),
    "wae_head_bwd": (c_i32, [ctypes.POINTER(HeadDesc)] + [c_vp] * 7 + [c_f32] + [c_vp] * 5),
    "wae_head_bwd_packed_bytes": (c_i64, [ctypes.POINTER(HeadDesc)]),
    "wae_gemm_tm": (c_i32, [ctypes.POINTER(TmDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "wae_gemm_tm_ce": (c_i32, [ctypes.POINTER(TmDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, ctypes.POINTER(TmCe), c_vp]),
    "wae_gemm_tn_tiles": (c_i32, [c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "wae_glu_bwd_fused_supported": (c_i32, [c_i32, c_i32]),
    "wae_glu_bwd_fused_supported16": (c_i32, [c_i32, c_i32]),
    "wae_glu_bwd_fused": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "wae_glu_bwd_fused_dc": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "wae_gemm_tn_stream": (c_i32, [c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_vp]),
    "wae_gemm_tn_static": (c_i32, [c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_i64, c_vp]),
    "wae_sum_rows": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "wae_head_packed_bytes": (c_i64, [ctypes.POINTER(HeadDesc)]),
    "wae_dmol_loss_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_i32, c_vp]),
    "wae_dmol_sample": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_i32, c_vp]),
    "wae_clip_adam_ema": (c_i32, [c_vp] * 5 + [c_i64, c_vp, c_vp, c_i32] + [ctypes.c_double] * 7 + [c_vp]),
    "wae_ar_generate": (c_i32, [ctypes.POINTER(ArDesc), c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_i64] + [c_vp] * 7 + [c_i32]
                        + [c_vp] * 5),
    "wae_ar_generate_scalar": (c_i32, [ctypes.POINTER(ArDesc), c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_i64] + [c_vp] * 7 + [c_i32]
                               + [c_vp] * 3 + [c_f32, c_i32, c_vp, c_vp, c_vp]),
    "wae_ar_coop_msg_values": (c_i32, [ctypes.POINTER(ArDesc), c_i32]),
    "wae_ar_coop_acc_floats": (c_i64, [ctypes.POINTER(ArDesc)]),
    "wae_ar_generate_coop": (c_i32, [ctypes.POINTER(ArDesc), c_i32, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_i64] + [c_vp] * 7
                             + [c_i32] + [c_vp] * 8),
    "wae_ar_generate_coop_fused": (c_i32, [ctypes.POINTER(ArDesc), c_i32, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_i64] + [c_vp] * 7
                                   + [c_i32] + [c_vp] * 9),
    "wae_ce_logits_fwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "wae_ce_logits_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "wae_weighted_mean": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    "wae_masked_mean": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "wae_to_btc": (c_i32, [c_vp, c_vp] + [c_i32] * 5 + [c_vp]),
    "wae_to_btc_masked": (c_i32, [c_vp, c_vp] + [c_i32] * 5 + [c_vp, c_f32, c_vp]),
    "wae_from_btc": (c_i32, [c_vp, c_vp] + [c_i32] * 5 + [c_vp]),
    "wae_from_btc_scaled": (c_i32, [c_vp, c_vp] + [c_i32] * 5 + [c_f32, c_vp]),
    "wae_act_fwd": (c_i32, [c_vp, c_i64, c_i32, c_f32, c_vp]),
    "wae_act_bwd": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_f32, c_vp]),
}

_lib = None


class WaeError(RuntimeError):
    pass


def build(force=False):
    """Compile csrc/*.hip for gfx950 into libwae_hip.so (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", src, "clean"])
    subprocess.check_call(["make", "-C", src, "-j4"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WaeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the hot path)")
        # torch first: its wheel carries its own libamdhip64, and libwae_hip.so must bind to THAT runtime -- loaded the other
        # way round, /opt/rocm's copy comes in as a second HIP runtime that does not know torch's allocations
        # (hipMemcpyAsync on a torch tensor then fails with "invalid value")
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise WaeError(f"{what} failed (rc={rc}): {lib().wae_last_error().decode()}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())
