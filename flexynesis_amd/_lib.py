"""ctypes binding of libfxhip.so (the C ABI declared in include/fxhip.h).

There is NO fallback: if the shared library is missing or a symbol is absent this module raises at
import, and every op raises ``RuntimeError(fx_last_error_string())`` on a non-zero return code.
"""
from __future__ import annotations

import ctypes as C
import os

# torch FIRST: PyTorch-ROCm ships its own libamdhip64.  libfxhip.so must bind to that already-loaded runtime --
# loaded before torch it would pull in /opt/rocm's copy, and the process would hold two HIP runtimes (launches then
# fail with "no ROCm-capable device is detected" and torch's device pointers mean nothing to the second runtime).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# FXHIP_LIB: A/B-test another build of the SAME ABI (e.g. a previous libfxhip.so); never a fallback
LIB_PATH = os.environ.get("FXHIP_LIB") or os.path.join(_HERE, "csrc", "libfxhip.so")

P = C.c_void_p
I = C.c_int
L = C.c_long
F = C.c_float
U64 = C.c_ulonglong

# name -> (restype, argtypes) ; mirrors include/fxhip.h one to one
PROTOTYPES = {
    "fx_last_error_string": (C.c_char_p, []),
    "fx_version": (I, []),
    "fx_hip_runtime_version": (I, []),
    "fx_source_hash": (C.c_char_p, []),
    "fx_gather_rows": (I, [P, P, P, I, I, L, L, P, L, P]),
    "fx_gemm_workspace_bytes": (L, [I, I, I]),
    "fx_gemm_f32": (I, [I, P, P, P, P, I, I, I, L, L, L, I, P, L, P]),
    "fx_linear_dw_adam_f32": (I, [P, P, P, P, P, I, I, I, L, L, L, P, P]),
    "fx_colsum": (I, [P, P, I, I, L, P]),
    "fx_gemm_splitk": (I, [I, I, I]),
    "fx_gemm_f32_slabs": (I, [I, P, P, P, I, I, I, L, L, P]),
    "fx_linear_fwd_bf16x3_splitk": (I, [I, I, I]),
    "fx_linear_fwd_bf16x3_splitk_ex": (I, [I, I, I, I]),
    "fx_linear_fwd_bf16x3_slabs_ex": (I, [P, L, P, P, P, I, I, I, L, L, I, P]),
    "fx_linear_fwd_bf16x3_slabs": (I, [P, L, P, P, P, I, I, I, L, L, P]),
    "fx_bn_act_fwd_slabs": (I, [P, P, P, I, L, P, P, P, P, P, P, P, P, I, I, L, L, I, I, I, F, U64, U64, P, P]),
    "fx_gram_hadamard_blocks": (I, [L]),
    "fx_gram_hadamard": (I, [P, P, I, P, I, L, P]),
    "fx_gather_split": (I, [P, P, P, P, P, P, P, I, I, L, L, L, L, P, L, P]),
    "fx_block_bwd_blocks": (I, [I]),
    "fx_block_bwd": (I, [P, P, P, P, P, P, I, P, P, P, P, P, P, P, P, P, P, P, L, P, P, I, I, L, L, I, I, F, I, P]),
    "fx_block_bwd_ex": (I, [P, P, P, P, P, P, I, P, P, P, P, P, P, P, P, P, P, P, L, P, P, I, I, L, L, I, I, F, I, P, P, L, I, P]),
    "fx_enc_tail_blocks": (I, [I]),
    "fx_enc_tail_fwd": (I, [P, I, I, I, I, I, F, P, P]),
    "fx_fusion_fwd": (I, [P, L, P, L, P, P, P, P, I, P, P, I, I, P]),
    "fx_fusion_fwd_pair": (I, [P, P, P, P, P, P, P, P, I, P, P, I, I, P]),
    "fx_block_bwd_group": (I, [P, I, I, I, I, F, P]),
    "fx_gather_split_group": (I, [P, I, P, I, P, L, P]),
    "fx_gram_kb_slices": (I, [I]),
    "fx_gram_kb_group": (I, [P, P, P, P, I, I, P]),
    "fx_reduce_group": (I, [P, P, P, P, P, P, I, P]),
    "fx_heads_fwd": (I, [P, I, P, L, I, I, I, F, P, P]),
    "fx_heads_bwd": (I, [P, I, P, L, P, L, I, I, I, F, P, P]),
    "fx_heads_step_scratch_floats": (L, [I, I, I]),
    "fx_heads_step": (I, [P, I, P, P, P, P, P, P, L, P, L, I, I, I, F, P, P, I, I, P, P, P, P, P, P]),
    "fx_split_bf16": (I, [P, P, P, I, I, L, L, P]),
    "fx_split_bf16_t": (I, [P, P, P, I, I, L, L, P]),
    "fx_linear_fwd_bf16x3_workspace_bytes": (L, [I, I, I]),
    "fx_linear_fwd_bf16x3": (I, [P, P, P, P, P, I, I, I, L, L, L, P, L, P]),
    "fx_linear_dw_adam_bf16x3": (I, [P, P, P, P, P, P, P, I, I, I, L, L, L, P, P]),
    "fx_linear_dw_adam_bf16x3_ex": (I, [P, P, P, P, P, P, P, I, I, I, L, L, L, P, I, I, I, P]),
    "fx_linear_fwd_bf16x3_ex": (I, [P, P, P, P, P, I, I, I, L, L, L, P, L, I, I, I, I, P]),
    "fx_linear_dw_adam_fwd_bf16x3_slabs": (I, [I, I]),
    "fx_linear_dw_adam_fwd_bf16x3_slabs_ex": (I, [I, I, I, I]),
    "fx_linear_dw_adam_fwd_bf16x3": (I, [P, P, P, P, P, P, P, I, I, I, L, L, L, P, P, P, L, I, P, L, I, P]),
    "fx_reduce_slabs": (I, [P, P, P, I, I, L, I, L, P]),
    "fx_reduce_slabs_par": (I, [P, P, P, I, I, L, I, L, P]),
    "fx_placement_probe": (I, [P, P, P, I, I, L, P]),
    "fx_placement_probe_oop": (I, [P, P, P, P, P, P, I, I, L, P]),
    "fx_linear_dw_adam_fwd_bf16x3_oop": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, L, L, L, P, P, P, L, I, P, L, I, P]),
    "fx_linear_bwd_x_bf16x3": (I, [P, P, P, P, I, I, I, L, L, L, P, L, P]),
    "fx_bn_act_fwd": (I, [P, P, P, P, P, P, P, P, P, P, I, I, L, L, I, I, I, F, U64, U64, P, P]),
    "fx_bn_act_bwd": (I, [P, P, P, P, P, P, P, P, P, P, I, I, L, L, L, L, I, I, F, I, P]),
    "fx_small_linear_fwd": (I, [P, P, P, P, I, I, I, L, L, P]),
    "fx_small_linear_bwd": (I, [P, P, P, P, P, P, I, I, I, L, L, L, I, P]),
    "fx_small_linear_bwd_group": (I, [P, I, P]),
    "fx_bn_eval_bwd": (I, [P, P, P, P, P, P, I, I, L, L, L, L, I, I, P]),
    "fx_sigmoid": (I, [P, P, L, P]),
    "fx_sigmoid_bwd": (I, [P, P, P, L, P]),
    "fx_softmax_rows": (I, [P, P, I, I, L, L, P]),
    "fx_reparam": (I, [P, P, P, P, P, L, U64, U64, P, P]),
    "fx_mul": (I, [P, P, P, L, P]),
    "fx_fill_normal": (I, [P, L, U64, U64, P, P]),
    "fx_randperm_scratch_bytes": (L, [L]),
    "fx_randperm": (I, [P, P, L, U64, U64, P, L, P]),
    "fx_triplet_sample": (I, [P, P, P, L, P, P, P, P, P, I, U64, U64, P, P]),
    "fx_lease_queue_create": (P, []),
    "fx_lease_queue_destroy": (None, [P]),
    "fx_lease_wrap": (P, [P, P, C.c_longlong, I, I, C.c_longlong]),
    "fx_lease_discard": (None, [P]),
    "fx_lease_drain": (I, [P, P, I]),
    "fx_lease_outstanding": (C.c_longlong, [P]),
    "fx_graph_begin": (I, [P, I]),
    "fx_graph_end": (I, [P, P, P]),
    "fx_graph_abort": (I, [P]),
    "fx_graph_launch": (I, [P, P]),
    "fx_graph_destroy": (I, [P]),
    "fx_graph_capturing": (I, [P]),
    "fx_mse_masked": (I, [P, P, P, P, I, L, L, P, F, P]),
    "fx_ce_masked": (I, [P, P, P, P, I, I, L, L, P, F, P]),
    "fx_cox_ph": (I, [P, P, P, P, P, I, L, L, P, F, P]),
    "fx_triplet": (I, [P, P, P, P, P, P, P, I, I, L, F, P, F, P]),
    "fx_mmd_workspace_floats": (L, [I, I]),
    "fx_mmd_rows": (I, [P, P, P, P, I, I, I, L, P, F, P]),
    "fx_mmd_rows_ex": (I, [P, P, P, P, I, I, I, L, P, F, I, P]),
    "fx_recon_blocks": (I, [L]),
    "fx_recon_sigmoid": (I, [P, P, P, P, P, L, P, F, P]),
    "fx_recon_sigmoid_slabs_blocks": (I, [I, I]),
    "fx_recon_sigmoid_slabs": (I, [P, P, P, P, P, I, L, P, P, I, I, L, P, F, P]),
    "fx_mmd_finalize": (I, [P, P, I, I, P, I, F, F, I, P]),
    "fx_total_loss": (I, [P, I, I, P, P, P, P, P]),
    "fx_step_begin": (I, [P, F, I, P]),
    "fx_adam_flat_clip": (I, [P, P, P, P, L, P, P, P, I, F, P]),
    "fx_fill": (I, [P, L, F, P]),
    "fx_scale_by": (I, [P, L, P, P]),
    "fx_stream_copy": (I, [P, P, L, P]),
    "fx_sumsq_blocks": (I, [L]),
    "fx_sumsq": (I, [P, P, L, P]),
    "fx_hadamard_sum": (I, [P, P, P, L, P]),
    "fx_clip_finalize": (I, [P, P, I, F, P]),
    "fx_adam_flat": (I, [P, P, P, P, L, P, P, P]),
    "fx_col_moments_chunks": (I, [I, I]),
    "fx_col_moments_workspace_bytes": (L, [I, I]),
    "fx_col_moments": (I, [P, I, L, I, I, P, P, I, P, P, P, P, P]),
    "fx_col_median": (I, [P, I, L, I, P, I, P, P]),
    "fx_row_moments": (I, [P, I, L, I, P, I, P, P, P]),
    "fx_ingest_transform": (I, [P, I, L, P, I, P, I, P, I, P, P, P, L, P]),
    "fx_gnn_row_blocks": (I, [L]),
    "fx_spmm_rows": (I, [P, P, P, P, P, I, I, I, L, P]),
    "fx_rowlin2": (I, [P, P, P, I, P, P, I, P, L, I, I, I, P]),
    "fx_rowlin_wgrad_workspace_bytes": (L, [L, I, I]),
    "fx_rowlin_wgrad": (I, [P, P, P, P, L, I, I, I, P, P]),
    "fx_bn_rows_workspace_bytes": (L, [L, I]),
    "fx_bn_rows_fwd": (I, [P, P, P, P, P, P, P, P, P, L, I, I, I, F, U64, U64, P, P, P]),
    "fx_bn_rows_bwd": (I, [P, P, P, P, P, P, P, P, P, L, I, I, F, U64, U64, P, P, P]),
}

# functions whose int return value is a size/count, not an error code
_QUERIES = {"fx_version", "fx_gnn_row_blocks", "fx_rowlin_wgrad_workspace_bytes", "fx_bn_rows_workspace_bytes", "fx_col_moments_chunks", "fx_col_moments_workspace_bytes", "fx_block_bwd_blocks", "fx_enc_tail_blocks", "fx_gram_kb_slices", "fx_gemm_splitk", "fx_linear_fwd_bf16x3_splitk", "fx_gram_hadamard_blocks", "fx_gemm_workspace_bytes", "fx_linear_fwd_bf16x3_workspace_bytes", "fx_mmd_workspace_floats", "fx_recon_blocks", "fx_recon_sigmoid_slabs_blocks", "fx_sumsq_blocks",
            "fx_last_error_string"}


class SmallLinearJob(C.Structure):
    """include/fxhip.h: fx_small_linear_job."""
    _fields_ = [("dx", C.c_void_p), ("gW", C.c_void_p), ("gb", C.c_void_p), ("dy", C.c_void_p), ("dy_mul", C.c_void_p),
                ("x", C.c_void_p), ("W", C.c_void_p), ("R", C.c_int), ("O", C.c_int), ("K", C.c_int), ("dx_accumulate", C.c_int),
                ("ldx", C.c_long), ("lddy", C.c_long), ("ldmul", C.c_long), ("lddx", C.c_long)]


class HeadDesc(C.Structure):
    """include/fxhip.h: fx_head_desc (host struct of device pointers)."""
    _fields_ = [(n, P) for n in ("W1", "b1", "gamma", "beta", "running_mean", "running_var", "W2", "b2", "y1", "a1",
                                 "save_mean", "save_invstd", "out", "mask", "dout", "gW1", "gb1", "ggamma", "gbeta",
                                 "gW2", "gb2")] + [("seed", U64), ("offset", U64), ("hidden", I), ("n_out", I)]


class BlockBwdDesc(C.Structure):
    """include/fxhip.h: fx_block_bwd_desc."""
    _fields_ = [("dE", P * 2), ("ldE", L * 2), ("W", P * 2), ("gW", P * 2), ("gb", P * 2), ("L", I * 2), ("n_up", I),
                ("x", P), ("out", P), ("gamma", P), ("save_mean", P), ("save_invstd", P), ("dgamma", P), ("dbeta", P),
                ("dbias", P), ("dy", P), ("dyT_hi", P), ("dyT_lo", P), ("ldt", L), ("gram_x", P), ("slots", P), ("C", I),
                ("ldx", L), ("ldo", L), ("accumulate", I)]


class GatherSplitDesc(C.Structure):
    """include/fxhip.h: fx_gather_split_desc."""
    _fields_ = [("x", P), ("hi", P), ("lo", P), ("hiT", P), ("loT", P), ("src", P), ("n_cols", I), ("ld_src", L), ("ldx", L),
                ("ldo", L), ("ldt", L)]


class EncTailDesc(C.Structure):
    """include/fxhip.h: fx_enc_tail_desc."""
    _fields_ = ([("slabs", P), ("slab_stride", L), ("lin_bias", P), ("x", P), ("out", P), ("gamma", P), ("beta", P),
                 ("running_mean", P), ("running_var", P), ("save_mean", P), ("save_invstd", P), ("mask", P),
                 ("W", P * 2), ("part", P * 2), ("seed", U64), ("offset", U64), ("n_slabs", I), ("H", I), ("n_up", I), ("L", I * 2)])


class FxError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m flexynesis_amd.csrc.build` "
            "(or __graft_entry__.build()).  flexynesis_amd has no CPU/PyTorch fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:         # header / library mismatch (an older libfxhip.so beside newer Python, FXHIP_LIB): fail loudly
            raise ImportError(f"{LIB_PATH} does not export {name}: ABI mismatch between this Python package and the library "
                              "(include/fxhip.h declares it) -- rebuild with `python -m flexynesis_amd.csrc.build --force`") from None
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


# The hipGraph lifetime rules (a graph destructor synchronises the device, fatal during another capture) and
# the fork shapes the tapes use (every branch forks from the capture stream) were established on these HIP runtimes.  Another
# runtime is not refused -- the library's kernels do not depend on it -- but it is said once, where a crash report would look.
VALIDATED_HIP_RUNTIMES = ((7, 0), (7, 2))


def hip_runtime_version():
    v = int(lib.fx_hip_runtime_version())
    return None if v < 0 else (v // 10_000_000, (v // 100_000) % 100, v % 100_000)


def _check_runtime():
    import warnings
    try:
        v = hip_runtime_version()
    except Exception:
        v = None
    if v is not None and (v[0], v[1]) not in VALIDATED_HIP_RUNTIMES:
        warnings.warn(f"flexynesis_amd: HIP runtime {v[0]}.{v[1]}.{v[2]} -- the hipGraph capture / release rules of the engine were "
                      f"validated on {', '.join('%d.%d' % x for x in VALIDATED_HIP_RUNTIMES)} (set FX_LEVEL1_GRAPHS=0 and "
                      "fit(use_graph=False) if captures misbehave)", RuntimeWarning, stacklevel=3)
    return v


def last_error() -> str:
    return (lib.fx_last_error_string() or b"").decode()


def check(rc: int, what: str = ""):
    if rc != 0:
        raise FxError(f"{what or 'libfxhip'} failed (rc={rc}): {last_error()}")


def call(name: str, *args):
    """Call an error-code-returning entry point and raise on failure."""
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise FxError(f"{name} failed (rc={rc}): {last_error()}")


def exported_symbols():
    return sorted(PROTOTYPES)
