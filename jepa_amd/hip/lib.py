"""ctypes binding of libvjepa_hip.so (include/vjepa_hip.h).

The library is the product: there is NO fallback.  `load_library()` raises if the shared object is missing or a
declared symbol is absent; every wrapper raises HipKernelError on a non-zero return code.
torch must be imported first so the HIP runtime the kernels bind to is the one torch already loaded
(both carry SONAME libamdhip64.so.7).
"""
import ctypes
import os

import torch  # noqa: F401  (loads torch/lib/libamdhip64.so before our DT_NEEDED is resolved)

_HERE = os.path.dirname(os.path.abspath(__file__))
_VARIANT = os.environ.get("VJ_LIB_VARIANT", "")   # A/B builds for kernel experiments (python -m jepa_amd.build <variant> ...)
_LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", f"libvjepa_hip_{_VARIANT}.so" if _VARIANT else "libvjepa_hip.so")

P = ctypes.c_void_p
I64 = ctypes.c_int64
I32 = ctypes.c_int
F32 = ctypes.c_float



class VjLinear(ctypes.Structure):           # vj_linear_t
    _fields_ = [("w", P), ("b", P), ("wT", P), ("ldwT", I64), ("gw", P), ("gb", P), ("n_out", I64), ("k_in", I64)]


class VjNorm(ctypes.Structure):             # vj_norm_t
    _fields_ = [("g", P), ("b", P), ("gg", P), ("gb", P)]


class VjBlock(ctypes.Structure):            # vj_block_t
    _fields_ = [("norm1", VjNorm), ("qkv", VjLinear), ("proj", VjLinear), ("norm2", VjNorm), ("fc1", VjLinear),
                ("fc2", VjLinear)]


class VjLnFold(ctypes.Structure):           # vj_lnfold_t
    _fields_ = [("w_qkv", P), ("c_qkv", P), ("b_qkv", P), ("w_fc1", P), ("c_fc1", P), ("b_fc1", P)]


class VjReduceSeg(ctypes.Structure):        # vj_reduce_seg_t
    _fields_ = [("part", P), ("out", P), ("P", I64), ("N", I64), ("stride", I64)]


class VjSeg(ctypes.Structure):              # vj_seg_t
    _fields_ = [("row0", I64), ("B", I64), ("S", I64)]


LAYER_CB = ctypes.CFUNCTYPE(None, P, I32)   # vj_layer_cb_t
F64P = ctypes.POINTER(ctypes.c_double)
I64P = ctypes.POINTER(ctypes.c_int64)

# name -> (restype, argtypes); mirrors include/vjepa_hip.h one to one
SIGNATURES = {
    "vj_abi_version": (I32, []),
    "vj_last_error": (ctypes.c_char_p, []),
    "vj_gather_rows": (I32, [P, P, P, I64, I64, I64, I64, P]),
    "vj_scatter_rows": (I32, [P, P, P, I64, I64, I64, I64, P]),
    "vj_copy_rows": (I32, [P, P, I64, I64, I64, I64, I64, I64, I64, P]),
    "vj_tubelet_pack": (I32, [P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, P]),
    "vj_add_pos": (I32, [P, P, P, I64, I64, I64, P]),
    "vj_layernorm_fwd": (I32, [P, P, P, P, P, P, I64, I64, F32, P]),
    "vj_layernorm_bwd_ws_bytes": (I64, [I64]),
    "vj_layernorm_bwd": (I32, [P, P, P, P, P, P, P, P, P, F32, F32, I64, I64, P, I64, P]),
    "vj_layernorm_bwd_colsum": (I32, [P, P, P, P, P, P, P, P, P, P, F32, F32, I64, I64, P, I64, P]),
    "vj_gemm_bf16_nt": (I32, [P, I64, P, I64, P, I64, I64, I64, I64, P, P, I64, P, P, I64, I32, F32, F32, I32, P]),
    "vj_gemm_colsum_rows": (I64, [I64]),
    "vj_gemm_bf16_nt_dgelu_colsum": (I32, [P, I64, P, I64, P, I64, I64, I64, I64, P, I64, P, I64, I32, ctypes.POINTER(I32), P]),
    "vj_gemm_bf16_nt_splitk": (I32, [P, I64, P, I64, P, I64, I64, I64, I64, F32, F32, I32, P, I64, P]),
    "vj_gemm_bf16_tn_splitk": (I32, [P, I64, P, I64, P, I64, I64, I64, I64, F32, F32, P, I64, P]),
    "vj_gemm_bf16_tn_grouped": (I32, [P, I64, I64, F32, F32, P, I64, P]),
    "vj_transpose_bf16": (I32, [P, P, I64, I64, I64, I64, P]),
    "vj_transpose_multi": (I32, [P, P, I64, P]),
    "vj_transpose_colsum_ws_bytes": (I64, [I64, I64]),
    "vj_transpose_colsum_bf16": (I32, [P, P, I64, I64, I64, I64, P, F32, F32, P, I64, P]),
    "vj_colsum_ws_bytes": (I64, [I64]),
    "vj_colsum_bf16": (I32, [P, I64, I64, I64, I64, I64, I64, P, F32, F32, P, I64, P]),
    "vj_reduce_partials": (I32, [P, P, I64, I64, F32, F32, P]),
    "vj_reduce_segments": (I32, [P, I64, F32, F32, P]),
    "vj_attn_fwd": (I32, [P, P, P, I64, I64, I64, I64, F32, P]),
    "vj_attn_fwd_segs": (I32, [P, P, P, ctypes.POINTER(VjSeg), I64, I64, I64, F32, P]),
    "vj_attn_bwd_segs": (I32, [P, P, P, P, P, ctypes.POINTER(VjSeg), I64, I64, I64, F32, P, I64, P, P, P]),
    "vj_attn_bwd_ws_bytes": (I64, [I64, I64, I64]),
    "vj_attn_bwd_segs_ws_bytes": (I64, [ctypes.POINTER(VjSeg), I64, I64, I64]),
    "vj_attn_bwd": (I32, [P, P, P, P, P, I64, I64, I64, I64, F32, P, I64, P]),
    "vj_attn_bwd_colsum_rows": (I32, [I64, I64, I64, I64P, I64P]),
    "vj_attn_bwd_colsum": (I32, [P, P, P, P, P, I64, I64, I64, I64, F32, P, I64, P, P, P]),
    "vj_xattn_fwd": (I32, [P, I64, P, P, P, P, I64, I64, I64, I64, I64, F32, P]),
    "vj_xattn_bwd": (I32, [P, I64, P, P, P, P, P, I64, I64, I64, I64, I64, F32, P]),
    "vj_pred_assemble_fwd": (I32, [P, P, P, P, P, P, I64, I64, I64, I64, P]),
    "vj_target_rows": (I32, [P, P, P, P, P, I64, I64, I64, I64, F32, F32, P]),
    "vj_latent_loss_ws_bytes": (I64, []),
    "vj_latent_loss": (I32, [P, P, P, I64, F32, F32, F32, I32, P, P, I64, P]),
    "vj_token_pstd": (I32, [P, P, P, I64, I64, I64, I32, P]),
    "vj_reg_grad": (I32, [P, P, P, P, I64, I64, I64, I64, F32, P]),
    "vj_reg_finish": (I32, [P, I64, I64, P, P]),
    "vj_adamw_ema": (I32, [P, P, P, P, P, P, P, I64, F32, F32, F32, F32, F32, I64, F32, F32, P]),
    "vj_step_advance": (I32, [P, P, P]),
    "vj_adamw_ema_guarded": (I32, [P, P, P, P, P, P, P, I64, F32, F32, F32, F32, F32, F32, F32, P, I32, F32, F32, P, P]),
    "vj_grad_stats_chunks": (I64, []),
    "vj_grad_stats_multi": (I32, [P, P, P, P, I64, P, P]),
    "vj_ema_update": (I32, [P, P, P, I64, F32, P]),
    "vj_cast_f32_to_bf16": (I32, [P, P, I64, P]),
    "vj_sqnorm_ws_bytes": (I64, []),
    "vj_sqnorm_f32": (I32, [P, I64, P, I32, P, I64, P]),
    "vj_blocks_fwd_ws_bytes": (I64, [I64, I64, I64, I64, I64, I32]),
    "vj_blocks_fwd": (I32, [ctypes.POINTER(VjBlock), I64, P, P, I64, I64, I64, ctypes.POINTER(VjSeg), I64, F32, I32, I32,
                            P, I64, P]),
    "vj_blocks_fwd_lnfold": (I32, [ctypes.POINTER(VjBlock), ctypes.POINTER(VjLnFold), I64, P, P, I64, I64, I64, ctypes.POINTER(VjSeg), I64,
                                   F32, I32, P, I64, P]),
    "vj_ln_rowstats": (I32, [P, P, I64, I64, F32, P]),
    "vj_ln_fold_weights": (I32, [P, P, P, P, P, P, P, I64, I64, P]),
    "vj_gemm_bf16_nt_lnfold": (I32, [P, I64, P, I64, P, I64, I64, I64, I64, P, P, P, I32, F32, I32, P]),
    "vj_blocks_bwd_ws_bytes": (I64, [I64, I64, I64, I64]),
    "vj_blocks_bwd": (I32, [ctypes.POINTER(VjBlock), I64, P, P, P, I64, I64, I64, ctypes.POINTER(VjSeg), I64, F32, F32,
                            P, I64, P, I64, I32, P, P, LAYER_CB, P]),
    "vj_prof_enable": (I32, [I32]),
    "vj_prof_collect": (I32, [F64P, F64P, I64P, ctypes.c_char_p]),
    "vj_comm_unique_id_bytes": (I64, []),
    "vj_comm_unique_id": (I32, [P]),
    "vj_comm_init": (I32, [ctypes.POINTER(P), I32, I32, P]),
    "vj_comm_allreduce_bucket": (I32, [P, P, I64, P]),
    "vj_comm_broadcast": (I32, [P, P, I64, I32, P]),
    "vj_comm_destroy": (I32, [P]),
    "vj_set_option": (I32, [ctypes.c_char_p, I32]),
    "vj_get_option": (I32, [ctypes.c_char_p, ctypes.POINTER(I32)]),
    "vj_probe_tr16": (I32, [P, I32, P]),
    "vj_probe_copy": (I32, [P, P, I64, P]),
    "vj_probe_lds_bw": (I32, [P, I32, I32, I32, P]),
    "vj_ws_guard_check": (I32, [I64P, I64P]),
    "vj_probe_spin": (I32, [I64, P]),
    "vj_probe_spin_stamped": (I32, [I64, P, P]),
}


class HipKernelError(RuntimeError):
    pass


_lib = None


def library_path():
    return _LIB_PATH


def load_library():
    """Load libvjepa_hip.so and bind every symbol of the C ABI.  Fails loudly -- there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise HipKernelError(
            f"{_LIB_PATH} is missing: build it with `python -m jepa_amd.build` (hipcc, gfx950). "
            "jepa_amd has no fallback compute path.")
    lib = ctypes.CDLL(_LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipKernelError(f"libvjepa_hip.so does not export {name}; rebuild the library") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load_library().vj_last_error()
        raise HipKernelError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def set_option(name, value):
    """Run-time tuning switch of the library (include/vjepa_hip.h: vj_set_option); returns the previous value."""
    old = get_option(name)
    check(load_library().vj_set_option(name.encode(), int(value)), "vj_set_option")
    return old


def get_option(name):
    v = I32(0)
    check(load_library().vj_get_option(name.encode(), ctypes.byref(v)), "vj_get_option")
    return int(v.value)
