"""ctypes binding of liblpd_hip.so (C-ABI declared in include/lpd_hip.h).

torch is imported first on purpose: liblpd_hip.so needs libamdhip64.so.7, and the dynamic loader
then reuses the HIP runtime torch already loaded (same SONAME), so device pointers and stream
handles are shared between torch and these kernels.

There is NO fallback: if the library is missing this raises, and every op in ops.py raises on
non-CUDA tensors.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede CDLL, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LPD_HIP_LIB") or os.path.join(_HERE, "liblpd_hip.so")   # override: kernel-variant builds (tools/)

_c_int = ctypes.c_int
_c_ll = ctypes.c_longlong
_c_f = ctypes.c_float
_c_p = ctypes.c_void_p

# name -> argtypes (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    "lpd_version": [],
    "lpd_last_error": [],
    "lpd_stat_ws_bytes": [],
    "lpd_knn": [_c_p, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_int, _c_p],
    "lpd_gemm": [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                 _c_ll, _c_ll, _c_ll, _c_int, _c_p, _c_p, _c_p, _c_p, _c_int, _c_f, _c_int, _c_ll, _c_ll, _c_int, _c_int, _c_p],
"lpd_gemm_bf16x3": [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                 _c_ll, _c_ll, _c_ll, _c_int, _c_p, _c_p, _c_p, _c_p, _c_int, _c_f, _c_int, _c_ll, _c_ll, _c_int, _c_int, _c_p],
    "lpd_gemm_bf16x1": [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                 _c_ll, _c_ll, _c_ll, _c_int, _c_p, _c_p, _c_p, _c_p, _c_int, _c_f, _c_int, _c_ll, _c_ll, _c_int, _c_int, _c_p],
    "lpd_gemm_prep_b_bytes": [_c_int, _c_int],
    "lpd_gemm_prep_b": [_c_p, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p],
    "lpd_gemm_prep_b_batch": [_c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_ll, _c_p, _c_p],
    "lpd_gemm_x3t_applies": [_c_int, _c_int, _c_int, _c_int, _c_ll, _c_ll, _c_int],
    "lpd_gemm_x3t_rows_applies": [_c_int, _c_int, _c_int, _c_int, _c_ll, _c_ll],
    "lpd_gemm_x3t_rows": [_c_p, _c_ll, _c_p, _c_p, _c_ll, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_int, _c_f, _c_int, _c_ll, _c_ll, _c_ll,
                          _c_p],
    "lpd_gemm_x3t": [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_int, _c_f, _c_ll, _c_ll, _c_int, _c_int, _c_p],
    "lpd_gemm_x3w": [_c_p, _c_int, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_int, _c_f, _c_int,
                     _c_ll, _c_ll, _c_int, _c_int, _c_int, _c_p],
    "lpd_gemm_p8_applies": [_c_int, _c_int, _c_int, _c_int],
    "lpd_gemm_p8": [_c_p, _c_p, _c_ll, _c_int, _c_p, _c_p, _c_int, _c_ll, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p,
                    _c_int, _c_f, _c_int, _c_p],
    "lpd_gemm_p8_fused": [_c_p, _c_p, _c_ll, _c_int, _c_p, _c_p, _c_int, _c_ll, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p,
                          _c_int, _c_f, _c_p, _c_p, _c_ll, _c_p],
    "lpd_softmax_affine_parts": [_c_p, _c_int, _c_ll, _c_p, _c_int, _c_p, _c_p, _c_int, _c_p, _c_int, _c_p],
    "lpd_split_panels": [_c_p, _c_ll, _c_int, _c_p, _c_p, _c_ll, _c_int, _c_int, _c_int, _c_int, _c_p],
    "lpd_gemm_x3w_batched": [_c_p, _c_int, _c_int, _c_p, _c_ll, _c_int, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_int, _c_f, _c_int, _c_p],
    "lpd_gemm_x3w_bf16a": [_c_p, _c_int, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p],
    "lpd_gemm_x3w_act": [_c_p, _c_int, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_int, _c_f, _c_p, _c_int, _c_int, _c_int, _c_p],
    "lpd_gemm_x3w_stats": [_c_p, _c_int, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_int, _c_p, _c_p],
    "lpd_retrieval_topk": [_c_p, _c_p, _c_int, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_p],
    "lpd_f64_to_f32": [_c_p, _c_p, _c_ll, _c_p],
    "lpd_best_pos_bwd": [_c_p, _c_ll, _c_p, _c_ll, _c_ll, _c_p, _c_p, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p],
    "lpd_hard_negatives": [_c_p, _c_ll, _c_p, _c_ll, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p],
    "lpd_knn_pm_layout": [_c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p],
    "lpd_lpdnet_front": [_c_p, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_f, _c_p, _c_int, _c_int, _c_int, _c_p, _c_p],
    "lpd_knn_pm": [_c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_int, _c_p],
    "lpd_knn_workspace_floats": [_c_int, _c_int, _c_int, _c_int],
    "lpd_edge_gather_max": [_c_p, _c_int, _c_p, _c_int, _c_p, _c_p, _c_int, _c_p, _c_p, _c_int, _c_int, _c_int,
                            _c_int, _c_int, _c_f, _c_p],
    "lpd_edge_gather_max16": [_c_p, _c_int, _c_p, _c_int, _c_p, _c_p, _c_int, _c_p, _c_p, _c_int, _c_int, _c_int,
                              _c_int, _c_int, _c_f, _c_ll, _c_ll, _c_ll, _c_int, _c_p],
    "lpd_edge_gather_max16s": [_c_p, _c_int, _c_p, _c_int, _c_p, _c_p, _c_ll, _c_p, _c_p, _c_int, _c_int, _c_int,
                               _c_int, _c_int, _c_f, _c_ll, _c_ll, _c_ll, _c_int, _c_p],
    "lpd_edge_mlp_x1_bf16x3s": [_c_p, _c_int, _c_p, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_ll, _c_int, _c_int, _c_int, _c_int,
                                _c_f, _c_ll, _c_int, _c_p],
    "lpd_edge_mlp_bf16x3s": [_c_p, _c_int, _c_p, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_ll, _c_int, _c_int,
                             _c_int, _c_int, _c_int, _c_int, _c_f, _c_ll, _c_int, _c_p],
    "lpd_gemm_x3ts": [_c_p, _c_ll, _c_p, _c_p, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_int, _c_f, _c_ll, _c_ll, _c_int, _c_int, _c_p],
    "lpd_pack_idx16": [_c_p, _c_p, ctypes.c_longlong, _c_int, _c_p],
    "lpd_pack_idx16w": [_c_p, _c_p, ctypes.c_longlong, _c_int, _c_int, _c_p],
    "lpd_edge_gather_maxw": [_c_p, _c_int, _c_p, _c_int, _c_p, _c_p, _c_int, _c_p, _c_p, _c_int, _c_int, _c_int,
                             _c_int, _c_int, _c_f, _c_ll, _c_ll, _c_ll, _c_int, _c_p],
    "lpd_edge_mlp": [_c_p, _c_int, _c_p, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int,
                     _c_int, _c_int, _c_int, _c_int, _c_f, _c_ll, _c_int, _c_p],
    "lpd_edge_mlp_bf16x3": [_c_p, _c_int, _c_p, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int,
                     _c_int, _c_int, _c_int, _c_int, _c_f, _c_ll, _c_int, _c_p],
    "lpd_linear_smallk": [_c_p, _c_int, _c_p, _c_int, _c_int, _c_ll, _c_int, _c_p, _c_int, _c_int, _c_int, _c_int,
                          _c_p, _c_p, _c_p, _c_int, _c_f, _c_p],
    "lpd_transpose": [_c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_ll, _c_ll, _c_p],
    "lpd_softmax_affine": [_c_p, _c_p, _c_int, _c_int, _c_p, _c_p, _c_int, _c_p, _c_int, _c_p],
    "lpd_vlad_finalize": [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p],
    "lpd_colmax": [_c_p, _c_int, _c_p, _c_int, _c_int, _c_int, _c_p],
    "lpd_mul": [_c_p, _c_p, _c_p, _c_ll, _c_p],
    "lpd_gating": [_c_p, _c_int, _c_p, _c_int, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_p],
    "lpd_morton_sort": [_c_p, _c_p, _c_p, _c_int, _c_int, _c_p],
    "lpd_colstats": [_c_p, _c_ll, _c_ll, _c_int, _c_p, _c_p, _c_p, _c_p],
    "lpd_bn_finalize": [_c_p, _c_p, ctypes.c_double, _c_int, _c_p, _c_p, _c_p, _c_p, _c_f, _c_f, _c_p, _c_p, _c_p, _c_p, _c_p],
    "lpd_affine_act": [_c_p, _c_ll, _c_p, _c_ll, _c_ll, _c_int, _c_p, _c_p, _c_int, _c_f, _c_p],
    "lpd_affine_act2": [_c_p, _c_ll, _c_p, _c_ll, _c_p, _c_ll, _c_ll, _c_int, _c_p, _c_p, _c_int, _c_f, _c_p],
    "lpd_bn_act_bwd": [_c_p, _c_ll, _c_p, _c_ll, _c_p, _c_ll, _c_ll, _c_int, _c_p, _c_p, _c_p, _c_p, _c_int, _c_f, _c_int,
                       _c_p, _c_p, _c_p, _c_p],
    "lpd_bn_act_bwd_bf16": [_c_p, _c_ll, _c_p, _c_ll, _c_p, _c_ll, _c_ll, _c_int, _c_p, _c_p, _c_p, _c_p, _c_int, _c_f, _c_int,
                       _c_p, _c_p, _c_p, _c_p],
    "lpd_edge_build": [_c_p, _c_ll, _c_p, _c_ll, _c_p, _c_p, _c_ll, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_p],
    "lpd_group_max": [_c_p, _c_ll, _c_int, _c_p, _c_p, _c_int, _c_f, _c_p, _c_ll, _c_p, _c_ll, _c_int, _c_p],
    "lpd_group_max_sel": [_c_p, _c_ll, _c_int, _c_p, _c_p, _c_int, _c_f, _c_p, _c_ll, _c_p, _c_p, _c_ll, _c_ll, _c_int, _c_p],
    "lpd_edge_bn_bwd_sel": [_c_p, _c_ll, _c_p, _c_p, _c_p, _c_ll, _c_p, _c_int, _c_ll, _c_int, _c_p, _c_p, _c_p, _c_p, _c_int, _c_f,
                            _c_p, _c_p, _c_p, _c_p],
    "lpd_edge_bn_bwd_bf16_sel": [_c_p, _c_ll, _c_p, _c_p, _c_p, _c_ll, _c_p, _c_int, _c_ll, _c_int, _c_p, _c_p, _c_p, _c_p, _c_int, _c_f,
                                 _c_p, _c_p, _c_p, _c_p],
    "lpd_group_max_bwd": [_c_p, _c_ll, _c_p, _c_int, _c_p, _c_ll, _c_int, _c_int, _c_p],
    "lpd_edge_bn_bwd": [_c_p, _c_ll, _c_p, _c_p, _c_p, _c_p, _c_p, _c_ll, _c_int, _c_ll, _c_int, _c_p, _c_p, _c_p, _c_p, _c_int,
                        _c_f, _c_f, _c_p, _c_p, _c_p, _c_p],
    "lpd_edge_mlp_train_bwd": [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_p, _c_p, _c_int, _c_int, _c_p, _c_p,
                               _c_p, _c_p, _c_int, _c_int, _c_int, _c_f, _c_f, _c_p, _c_p, _c_p],
    "lpd_edge_dense_bwd_apply": [_c_p, _c_int, _c_p, _c_p, _c_p, _c_ll, _c_p, _c_ll, _c_p, _c_p, _c_p, _c_ll, _c_p, _c_ll, _c_ll, _c_int, _c_int,
                                 _c_p, _c_p, _c_p, _c_p, _c_p, _c_p],
    "lpd_edge_mlp_train": [_c_p, _c_int, _c_p, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_p, _c_int, _c_p, _c_p, _c_p,
                           _c_int, _c_int, _c_int, _c_int, _c_f, _c_p, _c_p],
    "lpd_group_sum": [_c_p, _c_int, _c_p, _c_ll, _c_ll, _c_int, _c_p],
    "lpd_scatter_add_rows": [_c_p, _c_p, _c_p, _c_ll, _c_ll, _c_int, _c_int, _c_int, _c_p],
    "lpd_graph_transpose": [_c_p, _c_ll, _c_int, _c_int, _c_p, _c_p, _c_p, _c_p],
    "lpd_gather_sum_rows": [_c_p, _c_p, _c_p, _c_p, _c_ll, _c_ll, _c_int, _c_int, _c_p],
    "lpd_dw_smallk": [_c_p, _c_ll, _c_p, _c_ll, _c_ll, _c_int, _c_int, _c_p, _c_p],
    "lpd_colmax_arg": [_c_p, _c_ll, _c_p, _c_p, _c_int, _c_int, _c_int, _c_p],
    "lpd_colmax_bwd": [_c_p, _c_p, _c_p, _c_ll, _c_int, _c_int, _c_int, _c_p],
    "lpd_cloud_outer": [_c_p, _c_ll, _c_p, _c_ll, _c_p, _c_int, _c_int, _c_int, _c_p],
    "lpd_softmax_bwd": [_c_p, _c_p, _c_p, _c_p, _c_ll, _c_int, _c_int, _c_p],
    "lpd_vlad_finalize_bwd": [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_p],
    "lpd_edge_split_fwd": [_c_p, _c_ll, _c_p, _c_ll, _c_p, _c_p, _c_p, _c_p, _c_p, _c_ll, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_p],
    "lpd_edge_split_bwd": [_c_p, _c_ll, _c_p, _c_p, _c_p, _c_p, _c_ll, _c_p, _c_ll, _c_p, _c_p, _c_p, _c_p, _c_ll, _c_p, _c_ll, _c_ll,
                           _c_int, _c_int, _c_p, _c_p, _c_p, _c_p, _c_int, _c_f, _c_int, _c_p, _c_p, _c_p, _c_p],
    "lpd_edge_split_fwd16_applies": [_c_int, _c_int, _c_int],
    "lpd_edge_split_fwd16": [_c_p, _c_ll, _c_p, _c_ll, _c_p, _c_p, _c_p, _c_p, _c_p, _c_ll, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_p],
    "lpd_edge_build_bf16": [_c_p, _c_ll, _c_p, _c_ll, _c_p, _c_p, _c_ll, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_p],
    "lpd_edge_act_max": [_c_p, _c_int, _c_p, _c_p, _c_int, _c_f, _c_p, _c_p, _c_ll, _c_p, _c_ll, _c_int, _c_p],
    "lpd_edge_act_max_bf16": [_c_p, _c_int, _c_p, _c_p, _c_int, _c_f, _c_p, _c_p, _c_ll, _c_p, _c_ll, _c_int, _c_p],
    "lpd_group_sel_stats_bf16": [_c_p, _c_int, _c_p, _c_p, _c_ll, _c_p, _c_ll, _c_int, _c_p, _c_p, _c_p, _c_p],
    "lpd_edge_bn_bwd_bf16": [_c_p, _c_ll, _c_p, _c_p, _c_p, _c_p, _c_p, _c_ll, _c_int, _c_ll, _c_int, _c_p, _c_p, _c_p, _c_p, _c_int,
                             _c_f, _c_f, _c_p, _c_p, _c_p, _c_p],
    "lpd_bn_sel_bwd_reduce": [_c_p, _c_ll, _c_p, _c_ll, _c_ll, _c_int, _c_p, _c_p, _c_p, _c_p, _c_int, _c_f, _c_p, _c_p, _c_p, _c_p, _c_p],
    "lpd_edge_dw_sel_bf16": [_c_p, _c_p, _c_p, _c_int, _c_ll, _c_p, _c_ll, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p],
    "lpd_gemm_bf16s_bnbwd": [_c_p, _c_p, _c_p, _c_int, _c_ll, _c_p, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p],
    "lpd_bn_sel_bwd_reduce_f32": [_c_p, _c_ll, _c_p, _c_ll, _c_ll, _c_int, _c_p, _c_p, _c_p, _c_p, _c_int, _c_f, _c_p, _c_p, _c_p, _c_p, _c_p],
    "lpd_edge_dw_sel_f32": [_c_p, _c_p, _c_p, _c_int, _c_ll, _c_p, _c_ll, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p],
    "lpd_gemm_f32s_bnbwd": [_c_p, _c_p, _c_p, _c_int, _c_ll, _c_p, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p],
    "lpd_gather_sum_rows_bf16": [_c_p, _c_p, _c_p, _c_p, _c_ll, _c_ll, _c_int, _c_int, _c_p],
    "lpd_gemm_bf16s": [_c_p, _c_p, _c_int, _c_int, _c_p, _c_ll, _c_int, _c_int, _c_p],
    "lpd_gemm_tn_bf16_ws_floats": [_c_ll, _c_int, _c_int],
    "lpd_edge_dw_sel_bf16_ws_bytes": [_c_ll],
    "lpd_gemm_tn_bf16": [_c_p, _c_p, _c_p, _c_p, _c_ll, _c_int, _c_int, _c_p],
    "lpd_gemm_tn_ws_floats": [_c_ll, _c_int, _c_int, _c_int],
    "lpd_gemm_tn": [_c_p, _c_ll, _c_p, _c_ll, _c_p, _c_p, _c_ll, _c_int, _c_int, _c_int, _c_ll, _c_ll, _c_int, _c_p],
    "lpd_gemm_tn_act": [_c_p, _c_ll, _c_p, _c_ll, _c_p, _c_p, _c_ll, _c_int, _c_int, _c_int, _c_ll, _c_ll, _c_int, _c_p, _c_p, _c_int, _c_f, _c_p],
    "lpd_gemm_tn_act_ws_floats": [_c_ll, _c_int, _c_int, _c_int, _c_int],
    "lpd_metric_loss": [_c_p, _c_ll, _c_p, _c_ll, _c_ll, _c_p, _c_ll, _c_ll, _c_p, _c_ll, _c_int, _c_int, _c_int,
                        _c_int, _c_f, _c_f, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p],
}
_RESTYPES = {"lpd_last_error": ctypes.c_char_p, "lpd_stat_ws_bytes": ctypes.c_longlong, "lpd_knn_workspace_floats": ctypes.c_longlong,
             "lpd_gemm_prep_b_bytes": ctypes.c_longlong, "lpd_gemm_tn_bf16_ws_floats": ctypes.c_longlong, "lpd_gemm_tn_ws_floats": ctypes.c_longlong,
             "lpd_gemm_tn_act_ws_floats": ctypes.c_longlong,
             "lpd_edge_dw_sel_bf16_ws_bytes": ctypes.c_longlong}

_lib = None


class LpdHipError(RuntimeError):
    pass


def load():
    """Load liblpd_hip.so; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LpdHipError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` at the repo root (needs hipcc). There is no CPU/PyTorch fallback for this path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, ctypes.c_int)
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().lpd_last_error()
        raise LpdHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
