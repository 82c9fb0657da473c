"""ctypes binding of libstemgnn_hip.so (the C ABI declared in include/stemgnn.h).

There is no CPU or PyTorch fallback: if the HIP library has not been built the import
fails loudly, and every op raises if handed a non-CUDA tensor.
"""
import ctypes
import os
import re
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libstemgnn_hip.so")
HEADER_PATH = os.path.join(_PKG, "..", "include", "stemgnn.h")


class StemGnnLibraryError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the gfx950 HIP library has not been built. "
            "Run `python -m stem_gnn_amd.build` (or __graft_entry__.build()). "
            "stem_gnn_amd has no CPU fallback.")
    return ctypes.CDLL(LIB_PATH)


lib = _load()

P = c_void_p
I64 = c_int64
I32 = ctypes.c_int32


class GraphView(ctypes.Structure):
    """stemgnn_graph_view (include/stemgnn.h)."""
    _fields_ = [("num_nodes", I64), ("active_rows", I64), ("rowptr", P), ("src", P), ("eid", P), ("etype_slot", P),
                ("rowptr_t", P), ("dst_t", P), ("eid_t", P), ("etype_slot_t", P), ("inv_deg", P)]


class SageLayer(ctypes.Structure):
    """stemgnn_sage_layer (include/stemgnn.h)."""
    _fields_ = [("in_dim", I64), ("out_dim", I64), ("w_l", P), ("b_l", P), ("w_r", P), ("bn_weight", P), ("bn_bias", P),
                ("bn_running_mean", P), ("bn_running_var", P), ("bn_num_batches_tracked", P), ("bn_eps", c_float),
                ("bn_momentum", c_float), ("drop_seed", c_uint64), ("drop_offset", c_uint64), ("g_w_l", P), ("g_b_l", P),
                ("g_w_r", P), ("g_bn_weight", P), ("g_bn_bias", P)]


class VqParams(ctypes.Structure):
    """stemgnn_vq_params (include/stemgnn.h)."""
    _fields_ = [("dim", I64), ("heads", I64), ("code_dim", I64), ("codebook_size", I64), ("w_in", P), ("b_in", P),
                ("w_out", P), ("b_out", P), ("embed", P), ("commitment_weight", c_float), ("ortho_weight", c_float),
                ("ortho_ids", P), ("num_ortho_ids", I64), ("g_w_in", P), ("g_b_in", P), ("g_w_out", P), ("g_b_out", P),
                ("g_embed", P)]


class HeadsParams(ctypes.Structure):
    """stemgnn_heads_params (include/stemgnn.h)."""
    _fields_ = [("dim", I64), ("in_dim", I64), ("w_feat", P), ("b_feat", P), ("w_topo", P), ("b_topo", P), ("w_ts", P),
                ("b_ts", P), ("w_sem", P), ("b_sem", P), ("g_w_feat", P), ("g_b_feat", P), ("g_w_topo", P),
                ("g_b_topo", P), ("g_w_ts", P), ("g_b_ts", P), ("g_w_sem", P), ("g_b_sem", P)]


class EncoderCfg(ctypes.Structure):
    """stemgnn_encoder_cfg (include/stemgnn.h)."""
    _fields_ = [("num_layers", I32), ("use_bn", I32), ("training", I32), ("act", I32), ("negative_slope", c_float),
                ("dropout_p", c_float), ("out_rows", I64), ("feature_kind", I32)]

_SIGNATURES = {
    "stemgnn_abi_version": (c_int, []),
    "stemgnn_status_string": (c_char_p, [c_int]),
    "stemgnn_last_hip_error": (c_int, []),
    "stemgnn_csr_workspace_bytes": (c_size_t, [I64, I64]),
    "stemgnn_csr_build": (c_int, [P, I64, I64, c_int, P, P, P, P, P, c_size_t, P]),
    "stemgnn_graph_dropout_workspace_bytes": (c_size_t, [I64]),
    "stemgnn_graph_dropout_undirected": (c_int, [P, P, P, P, P, P, P, P, I64, I64, c_float, c_uint64, c_uint64, P,
                                                 P, P, P, P, P, P, P, P, P, c_size_t, P]),
    "stemgnn_graph_dropout_undirected_rows": (c_int, [P, P, P, P, P, P, P, P, I64, I64, I64, c_float, c_uint64, c_uint64,
                                                      P, P, P, P, P, P, P, P, P, P, c_size_t, P]),
    "stemgnn_sample_subset": (c_int, [I64, I64, c_uint64, c_uint64, P, P]),
    "stemgnn_mask_columns": (c_int, [P, I64, I64, c_float, c_uint64, c_uint64, P, P]),
    "stemgnn_negative_sample": (c_int, [P, P, P, P, I64, I64, c_uint64, c_uint64, P, P]),
    "stemgnn_negative_sample_into": (c_int, [P, P, P, P, I64, I64, c_uint64, c_uint64, P, I64, P]),
    "stemgnn_sample_edges": (c_int, [P, P, I64, I64, c_uint64, c_uint64, P, P, I64, P, P, P]),
    "stemgnn_edge_bce_loss": (c_int, [P, I64, I64, P, P, P]),
    "stemgnn_edge_dot_bwd_scaled": (c_int, [P, P, P, I64, I64, P, I64, P, P]),
    "stemgnn_sampler_init_map": (c_int, [P, I64, P]),
    "stemgnn_sampler_workspace_bytes": (c_size_t, [I64, I64, I64]),
    "stemgnn_sample_batch": (c_int, [P, P, P, I64, P, I64, P, I64, c_uint64, c_uint64, P, I64, I64, P, P, P, P, P, P, P,
                                     c_size_t, P]),
    "stemgnn_sampler_full_begin": (c_int, [P, I64, I64, P, P, P, P]),
    "stemgnn_sampler_full_hop_sizes": (c_int, [P, P, P, I32, I32, I64, P, P, P, P]),
    "stemgnn_sampler_full_hop_expand": (c_int, [P, P, P, P, I64, P, I32, I32, c_uint64, c_uint64, I64, I64, I64, P,
                                                P, P, P, P, P, P, P, P]),
    "stemgnn_sampler_full_finish_workspace_bytes": (c_size_t, [I64]),
    "stemgnn_sampler_full_finish": (c_int, [P, I32, P, P, P, P, P, P, P, P, I64, I64, P, P, P, P, P, P, P, P, P, P, P,
                                            c_size_t, P]),
    "stemgnn_sample_batch_views": (c_int, [P, P, P, I64, P, I64, P, I64, c_uint64, c_uint64, P, I64, I64, P, P, P, P, P, P,
                                           P, P, P, P, P, P, P, P, P, P, c_size_t, P]),
    "stemgnn_gather_i32": (c_int, [P, P, I64, P, P]),
    "stemgnn_group_by_key": (c_int, [P, I64, I64, P, P, P, c_size_t, P]),
    "stemgnn_sage_agg_fwd": (c_int, [P, I64, I64, P, P, P, P, P, P, I64, P, P]),
    "stemgnn_sage_agg_bwd": (c_int, [P, P, I64, I64, P, P, P, P, P, P, P, I64, P, P]),
    "stemgnn_sage_agg_bwd_acc": (c_int, [P, P, I64, I64, P, P, P, P, P, P, P, I64, P, P]),
    "stemgnn_vq_assign_bwd_fused": (c_int, [P, I64, P, P, c_float, P, P, P, P, I64, I64, I64, I64, P, P]),
    "stemgnn_vq_assign_lean": (c_int, [P, I64, I64, I64, P, P, I64, P, P, P, c_float, P, c_size_t, P]),
    "stemgnn_code_sqnorm": (c_int, [P, I64, I64, P, P]),
    "stemgnn_codes_project": (c_int, [P, P, P, I64, I64, I64, I64, P, P]),
    "stemgnn_code_segment_sums_workspace_bytes": (c_size_t, [I64, I64, I64, I64]),
    "stemgnn_code_segment_sums": (c_int, [P, I64, I64, P, I64, I64, P, P, c_size_t, P]),
    "stemgnn_segment_colsum": (c_int, [P, I64, I64, P, P]),
    "stemgnn_small_gemm": (c_int, [P, I64, I64, I64, P, I64, I64, I64, P, I64, I64, I64, I64, I64, I64, I64, P]),
    "stemgnn_vq_save_bytes": (c_size_t, [P, I64]),
    "stemgnn_vq_fwd": (c_int, [P, P, I64, c_int, P, P, P, P, c_size_t, P]),
    "stemgnn_vq_bwd_scratch_bytes": (c_size_t, [P, I64]),
    "stemgnn_vq_bwd": (c_int, [P, P, I64, P, P, P, P, P, c_size_t, P, c_size_t, P]),
    "stemgnn_heads_save_bytes": (c_size_t, [P, I64, I64, I64, I64]),
    "stemgnn_heads_fwd": (c_int, [P, P, P, P, I64, P, I64, P, P, P, I64, I64, c_uint64, c_uint64, c_uint64, c_uint64,
                                  P, P, P, P, P, P, P, c_size_t, P]),
    "stemgnn_heads_bwd_scratch_bytes": (c_size_t, [P, I64, I64, I64]),
    "stemgnn_heads_bwd": (c_int, [P, I64, P, P, P, I64, I64, P, P, P, P, P, c_size_t, I64, P, c_size_t, P]),
    "stemgnn_sage_agg_fwd_k": (c_int, [P, I32, I64, I64, P, P, P, P, P, P, I64, P, P]),
    "stemgnn_sage_agg_bwd_acc_k": (c_int, [P, P, I32, I64, I64, P, P, P, P, P, P, P, I64, P, P]),
    "stemgnn_linear_fwd_rows_k": (c_int, [P, P, I64, P, I32, P, I64, P, I64, I64, P, P, P, I64, I64, P]),
    "stemgnn_linear_bwd_weight_k": (c_int, [P, P, I32, I64, I64, I64, P, P, P, c_size_t, P]),
    "stemgnn_bn_act_drop_fwd_k": (c_int, [P, I64, I64, P, P, P, P, c_int, c_float, c_float, c_uint64, c_uint64, P, I32, P]),
    "stemgnn_mask_columns_k": (c_int, [P, I32, I64, I64, c_float, c_uint64, c_uint64, P, P]),
    "stemgnn_gather_rows_k": (c_int, [P, I32, I64, I64, P, I64, P, P, P]),
    "stemgnn_encoder_save_bytes": (c_size_t, [I64, I64, P, P]),
    "stemgnn_encoder_fwd": (c_int, [P, P, P, P, I64, P, P, P, P, c_size_t, P]),
    "stemgnn_encoder_bwd_scratch_bytes": (c_size_t, [I64, I64, P, P]),
    "stemgnn_encoder_bwd": (c_int, [P, P, P, P, I64, P, P, P, P, P, c_size_t, P, c_size_t, P]),
    "stemgnn_sage_agg_fwd_split": (c_int, [P, I64, I64, I64, P, P, P, P, P, P, I64, P, I32, I32, I32, I32, I64, I64,
                                           P, P, P, P, P, P, P]),
    "stemgnn_sage_agg_bwd_split": (c_int, [P, P, I64, I64, I64, P, P, P, P, P, P, P, I64, P, I32, I32, I32, I32, I64,
                                           I64, P, P, P, P, P, P, P]),
    "stemgnn_mean_agg_fwd": (c_int, [P, I64, I64, P, P, P, P]),
    "stemgnn_mean_agg_bwd": (c_int, [P, I64, I64, P, P, P, P, P]),
    "stemgnn_profile_k1": (c_int, [c_int]),
    "stemgnn_profile_k1_collect": (c_int, [P, P]),
    "stemgnn_profile_k1_collect_each": (c_int, [P, I64, P]),
    "stemgnn_inv_degree": (c_int, [P, I64, P, P]),
    "stemgnn_bn_workspace_bytes": (c_size_t, [I64, I64]),
    "stemgnn_bn_stats": (c_int, [P, I64, I64, c_float, P, P, P, P, c_float, P, c_size_t, P]),
    "stemgnn_bn_stats_from_partials": (c_int, [P, I64, I64, I64, c_float, P, P, P, P, c_float, P, P]),
    "stemgnn_linear_stats_partial_bytes": (c_size_t, [I64, I64]),
    "stemgnn_linear_stats_blocks": (I64, [I64, I64]),
    "stemgnn_linear_set_ws": (I32, [I32]),
    "stemgnn_linear_wsp_calls": (I64, []),
    "stemgnn_sample_edges2": (c_int, [P, P, I64, c_uint64, I64, c_uint64, P, P, I64, P, I64, c_uint64, P, P, I64, P, P]),
    "stemgnn_edge_concat_gather": (c_int, [P, I64, I64, P, I64, P, P, I64, P, P, P]),
    "stemgnn_edge_concat_bwd_add": (c_int, [P, I64, I64, P, I64, P, P, P, I64, P]),
    "stemgnn_edge_dot_bce_workspace_bytes": (c_size_t, [I64]),
    "stemgnn_edge_dot_bce": (c_int, [P, I64, I64, P, I64, I64, P, P, P, c_size_t, P]),
    "stemgnn_edge_det_workspace_bytes": (c_size_t, [I64, I64]),
    "stemgnn_edge_dot_bwd_det": (c_int, [P, P, P, I64, I64, P, I64, P, P, c_size_t, P]),
    "stemgnn_edge_concat_bwd_det": (c_int, [P, I64, I64, P, I64, P, P, c_size_t, P]),
    "stemgnn_set_deterministic": (c_int, [c_int]),
    "stemgnn_linear_few_rows": (c_int, [P, P, P, I64, I64, I64, P, I32, P]),
    "stemgnn_clip_grad_max_tensors": (I32, []),
    "stemgnn_clip_grad_workspace_bytes": (c_size_t, [I64, I32]),
    "stemgnn_clip_grad_norm": (c_int, [P, P, I32, c_float, P, P, c_size_t, P]),
    "stemgnn_weighted_sum": (c_int, [P, P, I32, P, P]),
    "stemgnn_weighted_sum_bwd": (c_int, [P, I32, P, P, P]),
    "stemgnn_grad_norm_coef": (c_int, [P, P, I32, c_float, P, P, c_size_t, P]),
    "stemgnn_adamw_step": (c_int, [P, P, P, P, P, I32, c_float, c_float, c_float, c_float, c_float, I64, P, P]),
    "stemgnn_linear_set_mode": (c_int, [c_int]),
    "stemgnn_linear_scratch_bytes": (c_size_t, [I64, I64, I64]),
    "stemgnn_vq_assign_scratch_bytes": (c_size_t, [I64, I64, I64, I64]),
    "stemgnn_linear_set_scratch": (c_int, [P, c_size_t, P]),
    "stemgnn_linear_set_bigtile": (c_int, [c_int]),
    "stemgnn_linear_set_pair": (c_int, [c_int]),
    "stemgnn_profile_bigtile": (c_int, [c_int]),
    "stemgnn_profile_bigtile_collect": (c_int, [P, P, P]),
    "stemgnn_linear_bigtile_calls": (I64, []),
    "stemgnn_linear_bigtile_fallbacks": (I64, []),
    "stemgnn_linear_bwd_data": (c_int, [P, P, I64, I64, I64, P, P]),
    "stemgnn_linear_fwd": (c_int, [P, P, I64, P, P, I64, P, I64, I64, P, P, P, I64, P]),
    "stemgnn_linear_fwd_rows": (c_int, [P, P, I64, P, P, I64, P, I64, I64, P, P, P, I64, I64, P]),
    "stemgnn_linear_bwd_weight_workspace_bytes": (c_size_t, [I64, I64, I64]),
    "stemgnn_linear_bwd_weight": (c_int, [P, P, I64, I64, I64, P, P, P, c_size_t, P]),
    "stemgnn_transpose": (c_int, [P, I64, I64, P, P]),
    "stemgnn_bn_act_drop_fwd": (c_int, [P, I64, I64, P, P, P, P, c_int, c_float, c_float, c_uint64, c_uint64, P, P]),
    "stemgnn_bn_act_drop_bwd": (c_int, [P, P, I64, I64, P, P, P, P, c_int, c_float, c_float, c_uint64, c_uint64,
                                        P, P, P, P, c_size_t, P]),
    "stemgnn_dropout_keep_mask": (c_int, [I64, c_float, c_uint64, c_uint64, P, P]),
    "stemgnn_vq_workspace_bytes": (c_size_t, [I64, I64, I64, I64]),
    "stemgnn_vq_assign_last_path": (c_int, []),
    "stemgnn_vq_assign_fwd": (c_int, [P, I64, I64, I64, P, I64, c_int, P, P, P, P, P, c_float, P, c_size_t, P]),
    "stemgnn_vq_assign_bwd": (c_int, [P, P, c_float, P, P, P, P, I64, I64, I64, I64, P, P]),
    "stemgnn_vq_ema_workspace_bytes": (c_size_t, [I64, I64, I64, I64]),
    "stemgnn_vq_ema_stats": (c_int, [P, P, P, I64, I64, I64, I64, P, P, P, c_size_t, P]),
    "stemgnn_loss_workspace_bytes": (c_size_t, [I64]),
    "stemgnn_mse_loss_fwd": (c_int, [P, P, I64, c_float, P, P, c_size_t, P]),
    "stemgnn_mse_loss_bwd": (c_int, [P, P, I64, c_float, P, P, P]),
    "stemgnn_cosine_loss_fwd": (c_int, [P, P, I64, I64, c_float, P, P, P, c_size_t, P]),
    "stemgnn_cosine_loss_bwd": (c_int, [P, P, I64, I64, c_float, P, P, P, P]),
    "stemgnn_ortho_loss_fwd": (c_int, [P, P, I64, I64, I64, I64, c_float, P, P, c_size_t, P]),
    "stemgnn_ortho_loss_bwd": (c_int, [P, P, I64, I64, I64, I64, c_float, P, P, P]),
    "stemgnn_edge_dot_fwd": (c_int, [P, I64, I64, P, I64, P, P]),
    "stemgnn_edge_dot_bwd": (c_int, [P, P, I64, I64, P, I64, P, P]),
    "stemgnn_edge_concat_fwd": (c_int, [P, I64, I64, P, I64, P, P]),
    "stemgnn_edge_concat_bwd": (c_int, [P, I64, I64, P, I64, P, P]),
    "stemgnn_gather_rows": (c_int, [P, I64, I64, P, I64, P, P]),
    "stemgnn_gather_rows_checked": (c_int, [P, I64, I64, P, I64, P, P, P]),
    "stemgnn_ema_lerp": (c_int, [P, P, I64, c_float, P]),
}


def declared_symbols():
    """Every function name include/stemgnn.h declares (used by the CPU export test)."""
    with open(HEADER_PATH) as f:
        text = f.read()
    return sorted(set(re.findall(r"\b(stemgnn_[a-z0-9_]+)\s*\(", text)))


for _name, (_res, _args) in _SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = header / library mismatch: fail at import
    _fn.restype = _res
    _fn.argtypes = _args

if lib.stemgnn_abi_version() != 1:
    raise ImportError("libstemgnn_hip.so ABI version mismatch; rebuild with python -m stem_gnn_amd.build")


def check(status: int, what: str = ""):
    if status != 0:
        msg = lib.stemgnn_status_string(status).decode()
        if status == -4:
            msg += f" (hipError_t {lib.stemgnn_last_hip_error()})"
        raise StemGnnLibraryError(f"{what or 'stemgnn call'} failed: {msg}")
