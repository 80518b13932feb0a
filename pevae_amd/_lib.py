"""ctypes binding of libpevae_hip.so (the C ABI declared in include/pevae_hip.h).

There is NO CPU fallback: if the shared library is missing or fails to load, importing the kernels raises.
`python __graft_entry__.py` (or `make -C pevae_amd/csrc`) builds it in-tree with hipcc for gfx950.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PEVAE_HIP_LIB: load a diagnostic build of the same library instead (tools/ only; there is still no CPU path)
LIB_PATH = os.environ.get("PEVAE_HIP_LIB") or os.path.join(_HERE, "lib", "libpevae_hip.so")

vp, i32, i64, f32, sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
lng = C.c_long


class SumGroup(C.Structure):
    """struct pv_sum_group_desc of include/pevae_hip.h."""
    _fields_ = [("dst", vp * 8), ("src", (vp * 4) * 8), ("n", lng * 8), ("count", i32)]


class BlockCall(C.Structure):
    """struct pv_block_call of include/pevae_hip.h (one Block of a pv_block_fwd / pv_block_bwd launch)."""
    _P5 = vp * 5
    _fields_ = [("M", i32), ("Din", i32), ("mid", i32), ("Dout", i32), ("ln_eps", f32), ("accumulate", i32),
                ("W", vp * 5), ("Wt", vp * 5), ("bias", vp * 5), ("inv_sigma", vp * 5), ("u", vp * 5), ("v", vp * 5),
                ("gamma", vp), ("beta", vp),
                ("x", vp), ("ldx", lng), ("save", vp), ("mean", vp), ("rstd", vp), ("out", vp), ("ldo", lng),
                ("dout", vp), ("lddo", lng), ("dx", vp), ("lddx", lng), ("dY", vp), ("cpart", vp), ("ppart", vp),
                ("dW", vp * 5), ("db", vp * 5), ("dgamma", vp), ("dbeta", vp)]


BLOCK_GROUP_MAX = 3

# name -> (restype, argtypes); mirrors include/pevae_hip.h one-to-one (tests/test_abi.py checks that)
SIGNATURES = {
    "pv_last_error": (C.c_char_p, []),
    "pv_abi_version": (i32, []),
    "pv_device_info": (i32, [C.POINTER(i32), C.POINTER(i32), C.c_char_p, i32]),
    "pv_gemm_bf16_nt": (i32, [vp, i32, vp, i32, i32, i32, i32, vp, i32, vp, i32, vp, i32, i32, vp, vp, vp]),
    "pv_gemm_splitk_plan": (i32, [i32, i32, i32, i32, C.POINTER(i32)]),
    "pv_gemm_splitk_ws_bytes": (sz, [i32, i32, i32]),
    "pv_gemm_bf16_nt_ct": (i32, [vp, i32, vp, i32, i32, i32, i32, vp, vp, i32, vp]),
    "pv_gemm_bf16_nt_splitk": (i32, [vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, i32, i32, vp, vp, vp]),
    "pv_gemm_bf16_tn_splitk": (i32, [vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, i32, i32, vp]),
    "pv_vocab_nll_fwd_ws_bytes": (sz, [i32, i32]),
    "pv_vocab_nll_fwd": (i32, [vp, i32, vp, i32, vp, vp, i32, i32, i32, f32, i64, vp, vp, vp, vp, vp]),
    "pv_vocab_nll_bwd_prob": (i32, [vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, f32, vp, i32, vp, i32, vp, vp, vp, vp]),
    "pv_vocab_compact": (i32, [vp, i32, i64, vp, vp, vp]),
    "pv_vocab_gather": (i32, [vp, i32, vp, vp, i32, i32, vp, i32, vp, vp]),
    "pv_scatter_rows_f32": (i32, [vp, lng, vp, vp, i32, i32, vp, lng, vp]),
    "pv_vocab_fused_supported": (i32, [i32]),
    "pv_vocab_fused_fwd_ws_bytes": (sz, [i32, i32]),
    "pv_vocab_fused_fwd": (i32, [vp, i32, vp, i32, vp, vp, i32, i32, i32, f32, i64, vp, vp, vp, vp, vp, vp]),
    "pv_vocab_fused_dw_ws_bytes": (sz, [i32]),
    "pv_vocab_fused_dw": (i32, [vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, f32, vp, vp, i32, i32, vp, vp, vp]),
    "pv_vocab_fused_dw_wgs": (i32, [vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, f32, vp, vp, i32, i32, vp, i32, vp, i32, vp]),
    "pv_vocab_fused_dw_range": (i32, [vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, f32, i32, i32, vp, vp, i32, i32, vp, i32, vp,
                                      i32, vp]),
    "pv_vocab_gather32": (i32, [vp, i32, vp, vp, i32, i32, vp, i32, vp, vp, vp]),
    "pv_scatter_rows_scaled_f32": (i32, [vp, vp, f32, vp, vp, i32, i32, vp, lng, vp]),
    "pv_scatter_rows_scaled_drop_f32": (i32, [vp, vp, f32, vp, vp, i32, i32, vp, lng, i32, i32, lng, lng, f32, vp, vp, i32, vp]),
    "pv_vocab_argmax_ws_bytes": (sz, [i32, i32]),
    "pv_vocab_argmax": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, f32, i32, vp, vp, vp, vp]),
    "pv_vocab_topk": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, f32, i32, i32, vp, vp, vp, vp]),
    "pv_vocab_sample_ws_bytes": (sz, [i32, i32]),
    "pv_vocab_sample": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, f32, i32, vp, lng, vp, i32, vp, vp, vp, vp]),
    "pv_vocab_argmax_noalloc": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, f32, i32, vp, vp, vp]),
    "pv_vocab_argmax_noalloc_max_bytes": (sz, []),
    "pv_greedy_decode_ws_bytes": (sz, [i32, i32, i32, i32]),
    "pv_greedy_decode": (i32, [vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp]),
    "pv_greedy_decode_error": (i32, [vp, vp]),
    "pv_sample_decode_ws_bytes": (sz, [i32, i32, i32, i32]),
    "pv_sample_decode": (i32, [vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp, i32, vp, vp, vp, vp, vp]),
    "pv_vocab_logprob_dense": (i32, [vp, i32, vp, i32, vp, vp, i32, i32, i32, f32, vp, vp]),
    "pv_logsoftmax_bwd_bf16": (i32, [vp, vp, i32, i32, f32, vp, i32, vp, vp]),
    "pv_gru_geometry": (i32, [i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "pv_gru_sync_bytes": (sz, [i32]),
    "pv_rnn_sync_error": (i32, [C.POINTER(i32), i32]),
    "pv_rnn_sync_error_inject": (i32, [vp]),
    "pv_gru_fwd": (i32, [vp, lng, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, vp]),
    "pv_gru_bwd": (i32, [vp, lng, lng, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "pv_block_layout": (i32, [i32, i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "pv_block_fwd": (i32, [vp, i32, vp]),
    "pv_block_bwd": (i32, [vp, i32, vp]),
    "pv_block_bwd_phase": (i32, [vp, i32, i32, vp]),
    "pv_block_sync_error": (i32, [C.POINTER(i32)]),
    "pv_block_rows": (i32, [i32, i32]),
    "pv_dls_level_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, f32, f32, vp, vp, vp, vp, vp]),
    "pv_dls_level_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, vp, vp, vp, vp, vp]),
    "pv_dls_level_next_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, f32, f32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp]),
    "pv_dls_level_next_add_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, f32, f32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp]),
    "pv_dls_level_next_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp]),
    "pv_dls_inputs_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp]),
    "pv_dls_inputs_bwd": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp]),
    "pv_lstm_geometry": (i32, [i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "pv_lstm_fwd": (i32, [vp, lng, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "pv_lstm_bwd": (i32, [vp, lng, lng, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "pv_embedding_gather": (i32, [vp, lng, i32, vp, i32, i32, i32, lng, lng, vp, i32, vp, i32, vp, vp]),
    "pv_embedding_bwd_ws_bytes": (sz, [i32, i32]),
    "pv_embedding_bwd": (i32, [vp, i32, vp, i32, i32, i32, lng, lng, lng, i32, lng, i32, vp, vp, sz, vp]),
    "pv_embedding_sort": (i32, [vp, i32, i32, i32, lng, lng, lng, i32, vp, sz, vp]),
    "pv_embedding_bwd_sorted": (i32, [vp, i32, i32, lng, i32, lng, i32, vp, vp, sz, vp]),
    "pv_sparse_pack": (i32, [vp, lng, i32, vp, i32, vp, vp, vp]),
    "pv_sparse_merge": (i32, [vp, i32, lng, i32, i32, lng, vp, vp]),
    "pv_cast_pack_bf16": (i32, [vp, i32, i32, i32, i32, lng, lng, lng, vp, i32, vp]),
    "pv_transpose_to_bf16": (i32, [vp, i32, i32, i32, lng, vp, lng, i32, vp]),
    "pv_colsum_f32": (i32, [vp, i32, lng, i32, vp, i32, vp]),
    "pv_masked_sum_bwd": (i32, [vp, i64, vp, lng, vp, vp]),
    "pv_vocab_gscale": (i32, [vp, vp, vp, i64, i32, vp, vp]),
    "pv_masked_sum": (i32, [vp, vp, i64, lng, vp, vp]),
    "pv_sum_group": (i32, [vp, vp]),
    "pv_nll_head_fwd": (i32, [vp, i32, vp, vp]),
    "pv_nll_head_bwd": (i32, [vp, vp, vp, vp, i32, vp, vp]),
    "pv_pick_nll_fwd": (i32, [vp, vp, i64, i32, i32, vp, vp]),
    "pv_pick_nll_bwd": (i32, [vp, vp, i64, i32, i32, vp, vp]),
    "pv_reparam_kl_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp]),
    "pv_reparam_kl_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp]),
    "pv_gumbel_softmax_fwd": (i32, [vp, vp, i32, i32, f32, i32, vp, vp, vp, vp]),
    "pv_softmax_bwd_rows": (i32, [vp, vp, i32, i32, f32, vp, vp]),
    "pv_cat_rows_fwd": (i32, [vp, vp, lng, i32, i32, i32, vp, vp]),
    "pv_cat_rows_bwd": (i32, [vp, vp, lng, vp, i32, i32, i32, vp, vp, vp]),
    "pv_linear_f32": (i32, [vp, lng, lng, vp, lng, lng, vp, vp, lng, i32, i32, i32, f32, vp, i32, i32, lng, lng, lng, vp]),
    "pv_layernorm_fwd": (i32, [vp, vp, vp, vp, i32, i32, f32, vp, vp, vp, vp]),
    "pv_layernorm_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, i32, vp]),
    "pv_spectral_norm": (i32, [vp, vp, vp, i32, i32, f32, i32, vp, vp]),
    "pv_spectral_norm_multi": (i32, [vp, i32, i32, i32, f32, i32, i32, vp, vp]),
    "pv_spectral_norm_multi_ws_bytes": (sz, [i32, i32]),
    "pv_spectral_norm_reduce": (i32, [vp, vp, vp, i32, i32, vp, i32, vp, vp]),
    "pv_linear_f32_fused": (i32, [vp, lng, lng, vp, lng, lng, vp, vp, lng, i32, i32, i32, f32, vp, i32, vp, vp, vp, i32, vp, vp, vp]),
    "pv_spectral_norm_bwd": (i32, [vp, vp, vp, lng, i32, i32, vp, vp, vp, vp, vp, vp]),
    "pv_cons_loss_ws_floats": (sz, [vp, i32, i32]),
    "pv_cons_loss_fwd": (i32, [vp, vp, vp, i32, i32, f32, vp, vp, vp]),
    "pv_cons_loss_bwd": (i32, [vp, vp, i32, i32, vp, vp, vp, vp]),
    "pv_l2norm_rows_fwd": (i32, [vp, i32, i32, vp, vp, vp]),
    "pv_l2norm_rows_bwd": (i32, [vp, vp, vp, i32, i32, vp, vp]),
    "pv_infonce_fwd": (i32, [vp, i32, vp, vp, vp, vp]),
    "pv_infonce_bwd": (i32, [vp, vp, vp, vp, i32, vp, vp]),
    "pv_additive_scores_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "pv_additive_scores_bwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    "pv_softmax_rows": (i32, [vp, i32, i32, vp, vp]),
    "pv_softmax_rows_bwd": (i32, [vp, vp, i32, i32, vp, vp]),
    "pv_unary_f32": (i32, [vp, vp, lng, i32, vp, vp]),
    "pv_rng_state_bytes": (sz, []),
    "pv_rng_advance": (i32, [vp, vp]),
    "pv_dropout_f32": (i32, [vp, i32, i32, i32, lng, lng, lng, lng, lng, f32, vp, vp, i32, vp, lng, lng, vp]),
    "pv_dropout_pack_bf16": (i32, [vp, i32, i32, i32, i32, lng, lng, lng, lng, lng, f32, vp, vp, i32, vp, i32, vp]),
    "pv_optim_state_bytes": (sz, []),
    "pv_grad_guard": (i32, [vp, lng, vp, f32, f32, i32, f32, f32, i32, f32, vp]),
    "pv_adam_step": (i32, [vp, vp, vp, vp, lng, vp, f32, f32, f32, f32, vp]),
}

_lib = None


class PevaeHipError(RuntimeError):
    pass


def load():
    """dlopen the library (once) and attach prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "pevae_amd: %s is missing -- build the HIP kernels first (python -c 'import __graft_entry__ as g; "
            "g.build()' or make -C pevae_amd/csrc).  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.pv_abi_version() != 2:
        raise ImportError("pevae_amd: ABI version mismatch in %s" % LIB_PATH)
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().pv_last_error().decode("utf-8", "replace")
        raise PevaeHipError("%s failed (code %d): %s" % (what or "libpevae_hip call", rc, msg))


def call(name, *args):
    lib = load()
    check(getattr(lib, name)(*args), name)
