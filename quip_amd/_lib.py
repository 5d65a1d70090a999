"""ctypes loader for libquip_amd.so (the C ABI of include/quip_amd.h).

The product path has no CPU fallback: if the shared library is missing or a symbol is absent, importing the
ops fails loudly here.  Build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950).
"""
import ctypes
import os

# torch bundles its own HIP runtime (torch/lib/libamdhip64.so).  It must be in the process BEFORE
# libquip_amd.so is dlopen'ed so that kernel registration and launches bind to the runtime that owns torch's
# streams and allocations, not to a second copy from /opt/rocm.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QUIP_AMD_LIB") or os.path.join(_HERE, "csrc", "libquip_amd.so")     # QUIP_AMD_LIB: A / B builds in the labs

c_i64, c_int, c_vp, c_double, c_float = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_double, ctypes.c_float

# name -> argtypes; must list every function declared in include/quip_amd.h (tests/test_abi.py checks)
SIGNATURES = {
    "quipamd_version": [],
    "quipamd_last_error": [],
    "quipamd_pack": [c_vp, c_int, c_int, c_vp, c_i64, c_i64, c_vp],
    "quipamd_unpack": [c_vp, c_int, c_int, c_vp, c_i64, c_i64, c_vp],
    "quipamd_repack_canonical_to_stream": [c_vp, c_int, c_vp, c_i64, c_i64, c_vp],
    "quipamd_vecquant_workspace_bytes": [c_int, c_i64, c_i64],
    "quipamd_vecquant_prepare": [c_int, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp],
    "quipamd_vecquant_invalidate": [c_vp],
    "quipamd_vecquant3matmul": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp],
    "quipamd_vecquant4matmul": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp],
    "quipamd_qfnb_scale": [c_vp, c_int, c_i64, c_vp, c_vp, c_vp],
    "quipamd_gridmap": [c_vp, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_i64, c_i64, c_vp],
    "quipamd_quantize": [c_vp, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_i64, c_i64, c_vp],
    "quipamd_codes_to_weight": [c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_int, c_i64, c_i64, c_vp],
    "quipamd_dequant_gemm": [c_vp, c_int, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_int,
                             c_i64, c_i64, c_i64, c_vp],
    "quipamd_dequant_gemm_cfg": [c_vp, c_int, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_int,
                                 c_i64, c_i64, c_i64, c_vp, c_vp],
    "quipamd_dequant_gemm_grouped": [c_int, c_vp, c_int, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_int,
                                     c_i64, c_i64, c_i64, c_vp],
    "quipamd_ortho_apply_small_ops": [c_vp, c_int, c_i64, c_vp],
    "quipamd_ortho_apply_tiles": [c_vp, c_vp, c_int, c_i64, c_vp],
    "quipamd_ortho_apply_tiles_supported": [c_int, c_int],
    "quipamd_ortho_apply_bigp": [c_vp, c_vp, c_int, c_i64, c_vp],
    "quipamd_ortho_apply_bigp_supported": [c_int, c_int],
    "quipamd_tune_dequant_gemm": [c_int, c_int, c_int, c_int],
    "quipamd_ortho_apply_rows": [c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_int, c_i64,
                                 c_vp, c_int, c_i64, c_i64, c_vp, c_vp],
    "quipamd_ortho_apply_small": [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_i64, c_vp, c_int,
                                  c_i64, c_i64, c_vp],
    "quipamd_dequant_gemm_vop": [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_i64, c_i64, c_vp],
    "quipamd_ortho_apply_small_chain": [c_vp, c_vp, c_int, c_i64, c_vp],
    "quipamd_ldlq_round": [c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_i64, c_i64, c_vp],
    "quipamd_unit_lower_t": [c_vp, c_vp, c_i64, c_vp],
    "quipamd_ldlq_greedy_pass": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp],
    "quipamd_gptq_round": [c_vp, c_vp, c_int, c_vp, c_vp, c_i64, c_i64, c_vp],
    "quipamd_gptq_feedback": [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp],
    "quipamd_unit_upper_inverse": [c_vp, c_vp, c_vp, c_i64, c_vp],
    "quipamd_gptq_round_groups": [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp],
    "quipamd_decode_attention": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_i64, c_int, c_int, c_i64, c_float, c_i64, c_vp],
    "quipamd_ortho_blocked_supported": [c_int, c_int],
    "quipamd_ortho_blocked_rows": [c_vp, c_vp, c_vp],
    "quipamd_ortho_blocked_rows_multi": [c_vp, c_int, c_vp, c_vp],
    "quipamd_ortho_blocked_config": [c_int, c_int],
    "quipamd_decode_attention_config": [c_int, c_int],
    "quipamd_decode_fused_gemm": [c_vp, c_vp],
    "quipamd_decode_bigp_supported": [c_int, c_int],
    "quipamd_decode_bigp_u": [c_vp, c_int, c_int, c_i64, c_vp, c_i64, c_vp],
    "quipamd_decode_bigp_v_gemm": [c_vp, c_vp],
    "quipamd_decode_head": [c_vp, c_vp],
    "quipamd_decode_embed": [c_vp, c_i64, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_int, c_i64, c_vp, c_i64, c_i64, c_vp],
    "quipamd_decode_u_only": [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_i64, c_i64, c_vp],
    "quipamd_argmax_rows": [c_vp, c_int, c_i64, c_i64, c_i64, c_vp, c_vp],
    "quipamd_decode_attention_fused": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_i64, c_float, c_i64, c_vp],
    "quipamd_rope_inplace": [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_int, c_i64, c_int, c_int, c_int, c_i64, c_i64, c_vp],
    "quipamd_cholesky_lt": [c_vp, c_vp, c_i64, c_vp, c_vp],
    "quipamd_preproc_workspace_bytes": [c_i64, c_i64],
    "quipamd_preproc_rescale": [c_vp, c_vp, c_int, c_i64, c_i64, c_vp, c_vp, c_vp],
    "quipamd_preproc_trace_ridge": [c_vp, c_i64, c_float, c_vp, c_vp],
    "quipamd_gptq_qfnb_workspace_bytes": [c_i64, c_i64],
    "quipamd_gptq_qfnb_info_offset": [c_i64, c_i64],
    "quipamd_gptq_qfnb_debug": [c_int, c_i64, c_int],
    "quipamd_gptq_round_qfnb": [c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp],
    "quipamd_cholesky_config": [c_int, c_int],
    "quipamd_ldlq_config": [c_int],
    "quipamd_hessian_accum": [c_vp, c_int, c_i64, c_i64, c_i64, c_vp, c_vp],
    "quipamd_hessian_finish": [c_vp, c_double, c_vp, c_i64, c_vp],
    "quipamd_hessian_fast_workspace": [c_i64, c_i64],
    "quipamd_hessian_accum_fast": [c_vp, c_int, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp],
    "quipamd_probe_set": [c_vp],
    "quipamd_dequant_gemm_grouped_config": [c_int],
    "quipamd_decode_prefetch_next": [c_vp, c_vp, c_int],
}

_lib = None


class QuipAmdError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise QuipAmdError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python __graft_entry__.py` "
            "(needs hipcc); there is no CPU fallback for the quip_amd hot path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise QuipAmdError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.argtypes = argtypes
        fn.restype = (ctypes.c_char_p if name == "quipamd_last_error" else None if name in ("quipamd_vecquant_invalidate", "quipamd_cholesky_config", "quipamd_ldlq_config", "quipamd_gptq_qfnb_debug", "quipamd_ortho_blocked_config",
                                                                                                      "quipamd_dequant_gemm_grouped_config", "quipamd_decode_attention_config") else
                      c_i64 if name in ("quipamd_hessian_fast_workspace", "quipamd_vecquant_workspace_bytes", "quipamd_gptq_qfnb_workspace_bytes", "quipamd_gptq_qfnb_info_offset", "quipamd_preproc_workspace_bytes") else c_int)
    _lib = lib
    return lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.quipamd_last_error()
        raise QuipAmdError(f"{name} failed (code {rc}): {msg.decode() if msg else '?'}")
