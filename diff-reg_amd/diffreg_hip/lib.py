"""ctypes binding of libdiffreg_hip.so (C ABI declared in include/diffreg_hip.h).

The library is the product: there is NO CPU fallback.  Importing this module on a box where the
library was not built raises ImportError; calling an op with CPU tensors raises RuntimeError.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdiffreg_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError("libdiffreg_hip.so is missing (%s): run `python __graft_entry__.py build` or "
                      "`make -C diff-reg_amd/csrc`" % LIB_PATH)

_lib = ctypes.CDLL(LIB_PATH)

c_int, c_size_t, c_void_p, c_char_p, c_double, c_float = (ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p,
                                                          ctypes.c_char_p, ctypes.c_double, ctypes.c_float)

# name -> (restype, argtypes); kept in the order of include/diffreg_hip.h
SIGNATURES = {
    "dr_version": (c_int, []),
    "dr_strerror": (c_char_p, [c_int]),
    "dr_last_hip_error": (c_char_p, []),
    "dr_sinkhorn_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "dr_sinkhorn_f32": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                c_void_p, c_void_p, c_size_t, c_void_p]),
    "dr_sinkhorn_f16": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "dr_sinkhorn_f64": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                c_void_p, c_void_p, c_size_t, c_void_p]),
}


class LayerWeights(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("q_proj", "k_proj", "v_proj", "merge", "mlp0", "mlp2", "norm1_w", "norm1_b",
                                        "norm2_w", "norm2_b")]


class LoopConfig(ctypes.Structure):
    _fields_ = [("variant", c_int), ("C", c_int), ("H", c_int), ("n_layers", c_int), ("steps", c_int),
                ("sk_iters", c_int), ("voxel", c_float), ("origin", c_float * 3), ("sample_rate", c_float),
                ("max_condition_num", c_float), ("flags", c_int), ("h_alphas_cumprod", c_void_p),
                ("h_times", c_void_p)]


class LoopWeights(ctypes.Structure):
    _fields_ = [("layers", ctypes.POINTER(LayerWeights)), ("src_proj", c_void_p), ("bin_score", c_void_p),
                ("pe_freq", c_void_p), ("prepacked", c_void_p)]


class FusionLayerWeights(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("q_w", "q_b", "k_w", "k_b", "v_w", "v_b", "lin_w", "lin_b", "norm1_w", "norm1_b",
                                        "expand_w", "expand_b", "squeeze_w", "squeeze_b", "norm2_w", "norm2_b")]


class FusionWeights(ctypes.Structure):
    _fields_ = [("layers", ctypes.POINTER(FusionLayerWeights))] + \
               [(n, c_void_p) for n in ("img_emb_w", "img_emb_b", "pcd_emb_w", "pcd_emb_b", "img_in_w", "img_in_b", "dino_w",
                                        "dino_b", "all_w", "all_b", "pcd_in_w", "pcd_in_b", "out_w", "out_b", "src_proj",
                                        "bin_score", "prepacked")]


class PlanesLinear(ctypes.Structure):
    """dr_planes_linear (include/diffreg_hip.h)"""
    _fields_ = [("rows", c_int), ("C", c_int), ("nblk", c_int),
                ("a0", c_void_p), ("bound0", c_void_p), ("k0", c_int),
                ("a1", c_void_p), ("bound1", c_void_p), ("k1", c_int),
                ("packed", c_void_p), ("mode", c_int),
                ("out", c_void_p), ("ldo", c_int), ("blk_stride", c_int),
                ("cos_t", c_void_p), ("sin_t", c_void_p), ("rot_mask", c_int), ("rot_C", c_int), ("scale", c_float),
                ("out_image", c_void_p), ("out_image_k", c_int), ("out_k0", c_int), ("out_bound", c_void_p),
                ("relu", c_int),
                ("gamma", c_void_p), ("beta", c_void_p), ("resid", c_void_p), ("ldr", c_int), ("bound_resid", c_void_p),
                ("ln_bound", c_void_p), ("bias", c_void_p), ("bias_max", c_void_p), ("ln_postadd", c_int), ("weight_layout", c_int),
                ("split_workspace", c_void_p), ("split_workspace_bytes", c_size_t)]


class Loop2D3DConfig(ctypes.Structure):
    _fields_ = [("C", c_int), ("H", c_int), ("n_layers", c_int), ("img_dim", c_int), ("dino_dim", c_int), ("pcd_dim", c_int),
                ("steps", c_int), ("sk_iters", c_int), ("sample_rate", c_float), ("max_condition_num", c_float),
                ("flags", c_int), ("h_alphas_cumprod", c_void_p), ("h_times", c_void_p)]


class LoopTrace(ctypes.Structure):
    """dr_loop_trace: per-step records and (ABI 0.2.0) the teacher-forcing inputs / outputs of the parity tests"""
    _fields_ = [("x0", c_void_p), ("R_forwd", c_void_p), ("t_forwd", c_void_p), ("cond", c_void_p), ("feats_nopos", c_void_p),
                ("feats_pos", c_void_p), ("force_x", c_void_p), ("force_R", c_void_p), ("force_t", c_void_p), ("x_next", c_void_p),
                ("topk_idx", c_void_p), ("wconf", c_void_p)]


_P = ctypes.POINTER
SIGNATURES.update({
    "dr_init": (c_int, []),
    "dr_prof_enable": (None, [c_int]),
    "dr_prof_collect": (c_int, [c_void_p, c_void_p, c_void_p]),
    "dr_vol_pe_f32": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float,
                              c_void_p, c_void_p, c_void_p, c_void_p]),
    "dr_linear_f32": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                              c_float, c_void_p]),
    "dr_plane_image_bytes": (c_size_t, [c_int, c_int]),
    "dr_planes_from_f32": (c_int, [c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "dr_planes_from_f32_bounded": (c_int, [c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dr_attention_planes": (c_int, [c_int] * 5 + [c_void_p] * 10 + [c_void_p]),
    "dr_attention_planes_f16": (c_int, [c_int] * 5 + [c_void_p] * 10 + [c_void_p]),
    "dr_planes_to_f32": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "dr_plane_weight_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "dr_pack_weight_planes_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dr_plane_weight_bytes_wide": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "dr_plane_split_workspace_bytes": (c_size_t, [c_int]),
    "dr_plane_split_status": (c_int, [c_void_p, c_void_p, c_int]),
    "dr_pack_weight_planes_wide_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dr_ln_bound_f32": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dr_linear_planes_f32": (c_int, [ctypes.POINTER(PlanesLinear), c_void_p]),
    "dr_bias_max_f32": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dr_loop2d3d_prepack_bytes": (c_size_t, [ctypes.POINTER(Loop2D3DConfig)]),
    "dr_loop2d3d_prepack": (c_int, [ctypes.POINTER(Loop2D3DConfig), ctypes.POINTER(FusionWeights), c_void_p, c_size_t, c_void_p]),
    "dr_gemm_nt_batched_f32": (c_int, [c_int, c_int, c_int, c_int, c_void_p, ctypes.c_longlong, c_void_p, ctypes.c_longlong, c_void_p, ctypes.c_longlong,
                                       c_float, c_void_p]),
    "dr_linear_ex_f32": (c_int, [c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "dr_kpconv_gather_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                     c_void_p, c_int, c_void_p]),
    "dr_kpconv_gather_mode_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                          c_int, c_int, c_void_p, c_int, c_void_p]),
    "dr_kpconv_gather_backward_mode_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                                   c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "dr_col_stats_workspace_bytes": (c_size_t, [c_int, c_int]),
    "dr_col_stats_f32": (c_int, [c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dr_norm_apply_f32": (c_int, [c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float,
                                  c_int, c_void_p, c_int, c_void_p]),
    "dr_gather_pool_f32": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "dr_kpconv_gather_backward_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                              c_void_p, c_int, c_void_p, c_void_p]),
    "dr_norm_backward_workspace_bytes": (c_size_t, [c_int, c_int]),
    "dr_norm_backward_f32": (c_int, [c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                     c_void_p, c_float, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "dr_gather_pool_backward_f32": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "dr_attention_layer_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dr_attention_layer_f32": (c_int, [_P(LayerWeights), c_int, c_int, c_int, c_int, c_int] + [c_void_p] * 9 +
                               [c_void_p, c_size_t, c_void_p]),
    "dr_attention_layer_pe_f32": (c_int, [_P(LayerWeights), c_int, c_int, c_int, c_int, c_int] + [c_void_p] * 11 +
                                  [c_void_p, c_size_t, c_void_p]),
    "dr_attention_layer_train_saved_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dr_attention_layer_train_forward_f32": (c_int, [_P(LayerWeights), c_int, c_int, c_int, c_int, c_int] + [c_void_p] * 10 + [c_size_t, c_void_p]),
    "dr_attention_layer_backward_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "dr_attention_layer_backward_f32": (c_int, [_P(LayerWeights), c_int, c_int, c_int, c_int, c_int] + [c_void_p] * 12 + [_P(LayerWeights), c_void_p, c_size_t, c_void_p]),
    "dr_procrustes_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dr_procrustes_f32": (c_int, [c_int, c_int, c_int] + [c_void_p] * 5 + [c_int, c_float, c_float] + [c_void_p] * 7 + [c_void_p, c_size_t, c_void_p]),
    "dr_device_status": (c_int, [c_void_p, c_int]),
    "dr_denoise_loop_status": (c_int, [c_void_p, c_void_p, c_int]),
    "dr_procrustes_backward_f32": (c_int, [c_int, c_int, c_int, c_int] + [c_void_p] * 9),
    "dr_debug_sinkhorn_spin_limit": (None, [ctypes.c_uint]),
    "dr_debug_enable_env": (None, [c_int]),
    "dr_debug_launch_chain": (c_int, [c_int, c_int, c_int, c_void_p]),
    "dr_debug_gemm_config": (None, [c_int]),
    "dr_debug_attention_config": (None, [c_int]),
    "dr_debug_attention_split": (None, [c_int]),
    "dr_pnp_ransac_workspace_bytes": (c_size_t, [c_int, c_int]),
    "dr_pnp_ransac_f64": (c_int, [c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_double, ctypes.c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                  c_void_p]),
    "dr_patch_similarity_f32": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_void_p, c_void_p]),
    "dr_unique_pairs_workspace_bytes": (c_size_t, [c_int]),
    "dr_unique_pairs_i64": (c_int, [c_int, c_void_p, c_void_p, ctypes.c_longlong, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dr_corr_gather_f32": (c_int, [c_int, c_void_p, c_void_p, ctypes.c_longlong, c_int] + [c_void_p] * 14),
    "dr_train_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dr_match_matrix_f32": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dr_gt_noising_f64": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_double, c_double, c_void_p, c_void_p, c_void_p]),
    "dr_focal_loss_f32": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_int, c_void_p, c_void_p,
                                  c_void_p]),
    "dr_match_recall_f32": (c_int, [c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dr_motion_l1_f32": (c_int, [c_int, c_int] + [c_void_p] * 10),
    "dr_motion_l1_backward_f32": (c_int, [c_int, c_int] + [c_void_p] * 11),
    "dr_layernorm_f32": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "dr_layernorm_backward_workspace_bytes": (c_size_t, [c_int]),
    "dr_layernorm_backward_f32": (c_int, [c_int, c_int] + [c_void_p] * 9),
    "dr_attention_f32": (c_int, [c_int] * 5 + [c_void_p] * 3 + [c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    "dr_attention_backward_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dr_attention_backward_f32": (c_int, [c_int] * 5 + [c_void_p] * 5 + [c_int, c_void_p, c_void_p, c_float] + [c_void_p] * 3 + [c_void_p, c_size_t, c_void_p]),
    "dr_softmax_rows_f32": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dr_dual_softmax_f32": (c_int, [c_int, c_int, c_int, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dr_dual_softmax_backward_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dr_dual_softmax_backward_f32": (c_int, [c_int, c_int, c_int, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dr_softmax_backward_f32": (c_int, [c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    "dr_relu_backward_f32": (c_int, [ctypes.c_longlong, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dr_rotary_f32": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p]),
    "dr_focal_loss_backward_f32": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "dr_sinkhorn_backward_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dr_sinkhorn_backward_f32": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_size_t, c_void_p]),
    "dr_scatter_rows_f32": (c_int, [c_int, c_int, c_void_p, ctypes.c_int64, c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_void_p, c_void_p]),
    "dr_mutual_match_f64": (c_int, [c_int, c_int, c_int, c_void_p, c_double, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dr_mutual_match_f32": (c_int, [c_int, c_int, c_int, c_void_p, c_float, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dr_top1_union_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dr_top1_union_f64": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dr_top1_union_f32": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dr_loop_prepack_bytes": (c_size_t, [_P(LoopConfig)]),
    "dr_loop_prepack": (c_int, [_P(LoopConfig), _P(LoopWeights), c_void_p, c_size_t, c_void_p]),
    "dr_denoise_loop_workspace_bytes": (c_size_t, [_P(LoopConfig), c_int, c_int, c_int]),
    "dr_denoise_loop": (c_int, [_P(LoopConfig), _P(LoopWeights), c_int, c_int, c_int] + [c_void_p] * 14 +
                        [_P(LoopTrace), c_void_p, c_size_t, c_void_p]),
    "dr_denoise_loop_2d3d_workspace_bytes": (c_size_t, [_P(Loop2D3DConfig), c_int, c_int, c_int]),
    "dr_denoise_loop_2d3d": (c_int, [_P(Loop2D3DConfig), _P(FusionWeights), c_int, c_int, c_int] + [c_void_p] * 16 +
                             [_P(LoopTrace), c_void_p, c_size_t, c_void_p]),
    "dr_inlier_ratio_f32": (c_int, [c_int] * 4 + [c_void_p] * 7 + [c_float, c_void_p, c_void_p, c_void_p]),
    "dr_nrfmr_f32": (c_int, [c_int] * 4 + [c_void_p] * 9 + [c_int, c_void_p, c_void_p, c_float, c_float] + [c_void_p] * 4),
    "dr_ransac_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dr_ransac_corr_f64": (c_int, [c_int] * 4 + [c_void_p] * 4 + [c_double, c_int, ctypes.c_uint64] + [c_void_p] * 7 +
                           [c_size_t, c_void_p]),
    "dr_registration_recall_f64": (c_int, [c_int] + [c_void_p] * 5 + [c_double, c_void_p, c_void_p, c_void_p]),
    "dr_grid_subsample_workspace_bytes": (c_size_t, [c_int, c_int]),
    "dr_grid_subsample_f32": (c_int, [c_int, c_int, c_void_p, c_void_p, c_float] + [c_void_p] * 5 + [c_size_t, c_void_p]),
    "dr_radius_neighbors_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dr_radius_neighbors_f32": (c_int, [c_int, c_int, c_int] + [c_void_p] * 4 + [c_float, c_int] + [c_void_p] * 4 + [c_size_t, c_void_p]),
    "dr_mutual_topk_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dr_mutual_topk_select_f32": (c_int, [c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_float, c_int] + [c_void_p] * 4 +
                                  [ctypes.c_longlong, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dr_denoiser_match_f32": (c_int, [_P(LoopConfig), _P(LoopWeights), c_int, c_int, c_int] + [c_void_p] * 9 +
                              [c_void_p, c_size_t, c_void_p]),
})


def _bind(table):
    for name, (res, args) in table.items():
        fn = getattr(_lib, name)
        fn.restype = res
        fn.argtypes = args


_bind(SIGNATURES)
_INIT_DONE = False

ABI_VERSION = 202          # DR_ABI_VERSION of the include/diffreg_hip.h these signatures were written against
if _lib.dr_version() // 100 != ABI_VERSION // 100:
    raise ImportError("libdiffreg_hip.so is ABI %d, this binding is written against %d: rebuild (make -C diff-reg_amd/csrc)"
                      % (_lib.dr_version(), ABI_VERSION))


def ensure_init():
    """kernel attributes (dynamic LDS) -- needs a GPU, so it runs lazily before the first launch."""
    global _INIT_DONE
    if not _INIT_DONE:
        # tools/ only: the library ignores the DR_* tuning variables unless this one explicit switch is set
        if os.environ.get("DR_DIAGNOSTICS") == "1":
            _lib.dr_debug_enable_env(1)
        check(_lib.dr_init())
        _INIT_DONE = True


SK_OUT_CONF, SK_OUT_LOG, SK_MINSHIFT, SK_APPLY_MASK, SK_OUT_F32, SK_STRICT = 0x0, 0x1, 0x2, 0x4, 0x8, 0x10
SK_RAGGED = 0x20    # rows / columns outside the masks do not exist (a padded tile = its unpadded problem)


def raw():
    return _lib


def check(code):
    if code != 0:
        msg = _lib.dr_strerror(code).decode()
        if code == -2:
            msg += ": " + _lib.dr_last_hip_error().decode()
        raise RuntimeError("libdiffreg_hip: %s (%d)" % (msg, code))


def loop_status(workspace, clear=True):
    """dr_denoise_loop_status: waits for the current stream of the workspace's device and raises if the LAST loop call that ran on this
    workspace reported a device-side failure (DR_ETIMEOUT of a co-resident Sinkhorn launch).  The word belongs to the workspace, so
    concurrent engines / concurrent batches of one engine (one workspace each) are told apart."""
    st = torch.cuda.current_stream(workspace.device).cuda_stream
    check(_lib.dr_denoise_loop_status(ptr(workspace), c_void_p(st), 1 if clear else 0))


def device_status(device=None, clear=True):
    """Waits for the current stream of `device` and raises if a kernel reported a device-side failure since the last check
    (DR_ETIMEOUT: the single-launch Sinkhorn gave up waiting for a workgroup that was not resident; the outputs of that call are
    unspecified -- the flag is process-wide and sticky: the first reader with clear=True consumes it).
    Called wherever the host mirrors synchronise anyway (match counts)."""
    st = torch.cuda.current_stream(device).cuda_stream
    check(_lib.dr_device_status(c_void_p(st), 1 if clear else 0))


def ptr(t):
    """device pointer of a contiguous ROCm tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libdiffreg_hip ops need tensors on a ROCm device (got %s); there is no CPU path" % t.device)
    if not t.is_contiguous():
        raise RuntimeError("libdiffreg_hip ops need contiguous tensors")
    return c_void_p(t.data_ptr())


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_of(t):
    """the current HIP stream of the tensor's device as a void* (the raw-handle query where this torch build has it: a tenth of the cost of building
    a torch.cuda.Stream object, and every library call makes one)"""
    if _RAW_STREAM is not None:
        dev = t.device
        return c_void_p(_RAW_STREAM(dev.index if dev.index is not None else torch.cuda.current_device()))
    return c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def mask_u8(m):
    if m is None:
        return None
    if m.dtype == torch.bool:
        return m.contiguous().view(torch.uint8)
    return (m != 0).contiguous().view(torch.uint8)


# ---- plane images (two-plane fp16 operand images of the layer GEMMs) -------------------------------------------------
PL_F32, PL_PLANES, PL_LN = 0, 1, 2


def planes_from_f32(x):
    """x [rows, K] fp32 -> (image uint8, bound [rows])"""
    ensure_init()
    x = x.contiguous()
    rows, K = x.shape
    img = torch.zeros(_lib.dr_plane_image_bytes(rows, K), dtype=torch.uint8, device=x.device)
    bnd = torch.empty(rows, device=x.device)
    check(_lib.dr_planes_from_f32(rows, K, ptr(x), K, ptr(img), ptr(bnd), stream_of(x)))
    return img, bnd


def planes_from_f32_bounded(x, bound_in):
    ensure_init()
    x = x.contiguous()
    rows, K = x.shape
    img = torch.zeros(_lib.dr_plane_image_bytes(rows, K), dtype=torch.uint8, device=x.device)
    bnd = torch.empty(rows, device=x.device)
    check(_lib.dr_planes_from_f32_bounded(rows, K, ptr(x), K, ptr(bound_in.contiguous()), ptr(img), ptr(bnd), stream_of(x)))
    return img, bnd


def attention_planes(q, k, v, H, q_mask=None, k_mask=None, f16=False):
    """q [P,Lq,C], k, v [P,Lk,C] float32 (rotary already applied) -> softmax(q k^T / sqrt(d)) v per head, [P,Lq,C], through the
    plane-image attention kernel (images built here: head-padded columns, one k / v bound per segment)."""
    ensure_init()
    P, Lq, C = q.shape
    Lk = k.shape[1]
    d = C // H
    dp = (d + 15) // 16 * 16

    def pad(x):
        y = torch.zeros(x.shape[0] * x.shape[1], H * dp, device=x.device)
        xr = x.reshape(-1, H, d)
        y.view(-1, H, dp)[:, :, :d] = xr
        return y
    qi, qb = planes_from_f32(pad(q))
    kb_in = k.abs().amax((1, 2)).repeat_interleave(Lk)
    vb_in = v.abs().amax((1, 2)).repeat_interleave(Lk)
    ki, kb = planes_from_f32_bounded(pad(k), kb_in)
    vi, vb = planes_from_f32_bounded(pad(v), vb_in)
    oi = torch.zeros(_lib.dr_plane_image_bytes(P * Lq, H * dp), dtype=torch.uint8, device=q.device)
    ob = torch.zeros(P * Lq, device=q.device)
    check((_lib.dr_attention_planes_f16 if f16 else _lib.dr_attention_planes)(P, Lq, Lk, H, d, ptr(qi), ptr(qb), ptr(ki), ptr(kb), ptr(vi), ptr(vb), ptr(mask_u8(q_mask)), ptr(mask_u8(k_mask)),
                                   ptr(oi), ptr(ob), stream_of(q)))
    o = planes_to_f32(oi, ob, P * Lq, H * dp).view(P * Lq, H, dp)[:, :, :d]
    return o.reshape(P, Lq, C)


def planes_to_f32(img, bnd, rows, K):
    out = torch.empty(rows, K, device=img.device)
    check(_lib.dr_planes_to_f32(rows, K, ptr(img), ptr(bnd), ptr(out), K, stream_of(out)))
    return out


def pack_weight_planes(W, nblk, C, piece_len=None, piece_pad=None, wide=False):
    """W [nblk * C, K] -> packed image (uint8 tensor); wide: the wide-wave layout (dr_pack_weight_planes_wide_f32; pass wide=True to linear_planes too)"""
    ensure_init()
    W = W.contiguous()
    K = W.shape[1]
    piece_len = piece_len or K
    piece_pad = piece_pad or K
    nbytes = (_lib.dr_plane_weight_bytes_wide if wide else _lib.dr_plane_weight_bytes)(nblk, C, K, piece_len, piece_pad)
    if nbytes == 0:
        raise RuntimeError("unsupported plane weight shape")
    buf = torch.zeros(nbytes, dtype=torch.uint8, device=W.device)
    check((_lib.dr_pack_weight_planes_wide_f32 if wide else _lib.dr_pack_weight_planes_f32)(nblk, C, K, piece_len, piece_pad, ptr(W), ptr(buf), stream_of(W)))
    return buf


def ln_bound(gamma, beta):
    out = torch.empty(1, device=gamma.device)
    check(_lib.dr_ln_bound_f32(gamma.numel(), ptr(gamma), ptr(beta), ptr(out), stream_of(gamma)))
    return out


def linear_planes(rows, C, nblk, a0, b0, k0, packed, mode, *, a1=None, b1=None, k1=0, out=None, ldo=0, blk_stride=0, cos_t=None,
                  sin_t=None, rot_mask=0, rot_C=0, scale=1.0, out_image=None, out_image_k=0, out_k0=0, out_bound=None, relu=False,
                  gamma=None, beta=None, resid=None, ldr=0, bound_resid=None, lnb=None, bias=None, ln_postadd=False, wide=False, split_ws=None):
    """dr_linear_planes_f32; bias [nblk * C]: its per-block maxima are computed here (dr_bias_max_f32).  split_ws: a uint8 tensor of
    plane_split_workspace(C) bytes -- lets a launch that fills at most half the chip split its tiles' k range over two workgroups"""
    a = PlanesLinear()
    a.rows, a.C, a.nblk = rows, C, nblk
    dp = lambda t_: None if t_ is None else t_.data_ptr()
    a.a0, a.bound0, a.k0 = dp(a0), dp(b0), k0
    a.a1, a.bound1, a.k1 = dp(a1), dp(b1), k1
    a.packed, a.mode = dp(packed), mode
    a.out, a.ldo, a.blk_stride = dp(out), ldo, blk_stride
    a.cos_t, a.sin_t, a.rot_mask, a.rot_C, a.scale = dp(cos_t), dp(sin_t), rot_mask, rot_C, scale
    a.out_image, a.out_image_k, a.out_k0, a.out_bound = dp(out_image), out_image_k, out_k0, dp(out_bound)
    a.relu = 1 if relu else 0
    a.gamma, a.beta, a.resid, a.ldr, a.bound_resid, a.ln_bound = dp(gamma), dp(beta), dp(resid), ldr, dp(bound_resid), dp(lnb)
    bmax = None
    if bias is not None:
        bias = bias.contiguous().float()
        bmax = torch.empty(nblk, device=bias.device)
        check(_lib.dr_bias_max_f32(nblk, C, ptr(bias), ptr(bmax), stream_of(a0)))
    a.bias, a.bias_max, a.ln_postadd = dp(bias), dp(bmax), 1 if ln_postadd else 0
    a.weight_layout = 1 if wide else 0
    a.split_workspace, a.split_workspace_bytes = dp(split_ws), (split_ws.numel() if split_ws is not None else 0)
    check(_lib.dr_linear_planes_f32(ctypes.byref(a), stream_of(a0)))


def plane_split_workspace(C, device):
    """the exchange workspace of dr_planes_linear.split_workspace (uint8 tensor)"""
    return torch.empty(_lib.dr_plane_split_workspace_bytes(C), dtype=torch.uint8, device=device)


def plane_split_status(split_ws, clear=True):
    """raises (DR_ETIMEOUT) if a workgroup of the last split launch on this workspace never met its partner"""
    check(_lib.dr_plane_split_status(ptr(split_ws), stream_of(split_ws), 1 if clear else 0))


def scatter_rows(src, src_index, dst_index, dst, validate=True, status=None):
    """dst[dst_index[i]] = src[src_index[i]] (rows; split_feats of the reference).  Index tensors may live on the host (torch
    indexing accepts that); out-of-range indices raise IndexError like torch's indexed assignment (validate=True reads a 4-byte
    device flag, i.e. synchronises the stream; validate=False leaves such rows unwritten and the flag unread: pass `status`, a zeroed
    int32 device tensor, to share one flag between several calls and check it once with scatter_rows_check)."""
    ensure_init()
    src = src.contiguous().float()
    si = src_index.to(device=src.device, dtype=torch.int64).contiguous()
    di = dst_index.to(device=src.device, dtype=torch.int64).contiguous()
    assert dst.is_contiguous() and dst.dtype == torch.float32 and dst.shape[-1] == src.shape[-1] and dst.device == src.device
    assert si.numel() == di.numel()
    C = src.shape[-1]
    if status is None:
        status = torch.zeros(1, dtype=torch.int32, device=src.device)
    check(_lib.dr_scatter_rows_f32(si.numel(), C, ptr(src), src.numel() // C, ptr(si), ptr(di), ptr(dst), dst.numel() // C,
                                   ptr(status), stream_of(src)))
    if validate and int(status.item()):
        raise IndexError("scatter_rows: index out of range (src rows %d, dst rows %d)" % (src.numel() // C, dst.numel() // C))
    return dst


def scatter_rows_check(status):
    if int(status.item()):
        raise IndexError("scatter_rows: index out of range")


# ---- forward half of the training branch (csrc/train.hip) -------------------------------------------------------------------------
MATCH_SINKHORN, MATCH_DUAL_SOFTMAX = 0, 1


def _train_ws(P, N, M, dev):
    return torch.empty(_lib.dr_train_workspace_bytes(P, N, M), dtype=torch.uint8, device=dev)


def match_matrix(matches, P, N, M):
    """rows (b, i, j) of matches [K,3] int64 -> [P,N,M] float32 with 1 at the rows (match_2_conf_matrix, loss.py:316-320)"""
    ensure_init()
    matches = matches.to(torch.int64).contiguous()
    out = torch.empty(P, N, M, device=matches.device)
    check(_lib.dr_match_matrix_f32(P, N, M, matches.shape[0], ptr(matches), ptr(out), stream_of(out)))
    return out


def gt_noising(matrix_gt, randn, sqrt_ac, sqrt_one_minus_ac):
    """pipeline.py:209-214 -> matrix_gt_disturbed [P,N,M] float64"""
    ensure_init()
    matrix_gt, randn = matrix_gt.contiguous().float(), randn.contiguous().float()
    P, N, M = matrix_gt.shape
    out = torch.empty(P, N, M, dtype=torch.float64, device=matrix_gt.device)
    ws = _train_ws(P, N, M, out.device)
    check(_lib.dr_gt_noising_f64(P, N, M, ptr(matrix_gt), ptr(randn), float(sqrt_ac), float(sqrt_one_minus_ac), ptr(out), ptr(ws), stream_of(out)))
    return out


def focal_loss(conf, conf_gt, weight=None, alpha=0.25, gamma=2.0, pos_w=1.0, neg_w=1.0, match_type="sinkhorn"):
    """compute_correspondence_loss (loss.py:273-314) -> 0-d float32 tensor"""
    ensure_init()
    conf, conf_gt = conf.contiguous().float(), conf_gt.contiguous().float()
    weight = weight.contiguous().float() if weight is not None else None
    P, N, M = conf.shape
    loss = torch.empty((), device=conf.device)
    ws = _train_ws(P, N, M, conf.device)
    mt = {"sinkhorn": MATCH_SINKHORN, "dual_softmax": MATCH_DUAL_SOFTMAX}[match_type]
    check(_lib.dr_focal_loss_f32(P, N, M, ptr(conf), ptr(conf_gt), ptr(weight), alpha, gamma, pos_w, neg_w, mt, ptr(loss), ptr(ws), stream_of(conf)))
    return loss


def match_recall(conf_gt, match_pred):
    """compute_match_recall (loss.py:323-345) -> (recall, precision) 0-d float32 tensors"""
    ensure_init()
    conf_gt = conf_gt.contiguous().float()
    match_pred = match_pred.to(torch.int64).contiguous()
    P, N, M = conf_gt.shape
    out = torch.empty(2, device=conf_gt.device)
    ws = _train_ws(P, N, M, conf_gt.device)
    check(_lib.dr_match_recall_f32(P, N, M, ptr(conf_gt), match_pred.shape[0], ptr(match_pred), ptr(out), ptr(ws), stream_of(conf_gt)))
    return out[0], out[1]


def motion_l1(s_pcd, R_pred, t_pred, R_gt, t_gt, overlap_mask, flow=None):
    """the L1 motion term of ge_coarse_loss (loss.py:108-128) -> 0-d float32 tensor"""
    ensure_init()
    P, N, _ = s_pcd.shape
    f = lambda x: x.contiguous().float()
    s_pcd, R_pred, t_pred, R_gt, t_gt = f(s_pcd), f(R_pred), f(t_pred.reshape(P, 3)), f(R_gt), f(t_gt.reshape(P, 3))
    flow = f(flow) if flow is not None else None
    loss = torch.empty((), device=s_pcd.device)
    ws = _train_ws(P, N, 1, s_pcd.device)
    om = mask_u8(overlap_mask)
    check(_lib.dr_motion_l1_f32(P, N, ptr(s_pcd), ptr(flow), ptr(R_pred), ptr(t_pred), ptr(R_gt), ptr(t_gt), ptr(om), ptr(loss),
                                ptr(ws), stream_of(s_pcd)))
    return loss


def motion_l1_backward(s_pcd, R_pred, t_pred, R_gt, t_gt, overlap_mask, flow=None):
    """d loss / d (R_pred, t_pred) of motion_l1 -> ([P,3,3], [P,3,1])"""
    ensure_init()
    P, N, _ = s_pcd.shape
    f = lambda x: x.contiguous().float()
    s_pcd, R_pred, t_pred, R_gt, t_gt = f(s_pcd), f(R_pred), f(t_pred.reshape(P, 3)), f(R_gt), f(t_gt.reshape(P, 3))
    flow = f(flow) if flow is not None else None
    gR, gt = torch.empty(P, 3, 3, device=s_pcd.device), torch.empty(P, 3, device=s_pcd.device)
    ws = _train_ws(P, N, 1, s_pcd.device)
    om = mask_u8(overlap_mask)
    check(_lib.dr_motion_l1_backward_f32(P, N, ptr(s_pcd), ptr(flow), ptr(R_pred), ptr(t_pred), ptr(R_gt), ptr(t_gt), ptr(om), ptr(gR), ptr(gt), ptr(ws),
                                         stream_of(s_pcd)))
    return gR, gt.view(P, 3, 1)


def layernorm(x, gamma, beta, eps=1e-5):
    """nn.LayerNorm over the last dim -> (y, mean_rstd [rows,2])"""
    ensure_init()
    x = x.contiguous().float()
    rows, C = x.reshape(-1, x.shape[-1]).shape
    y = torch.empty_like(x)
    st = torch.empty(rows, 2, device=x.device)
    check(_lib.dr_layernorm_f32(rows, C, ptr(x), ptr(gamma.contiguous()), ptr(beta.contiguous()), eps, ptr(y), ptr(st), stream_of(x)))
    return y, st


def layernorm_backward(x, gamma, mean_rstd, grad_y):
    ensure_init()
    x, grad_y = x.contiguous().float(), grad_y.contiguous().float()
    rows, C = x.reshape(-1, x.shape[-1]).shape
    gx = torch.empty_like(x)
    gg, gb = torch.empty(C, device=x.device), torch.empty(C, device=x.device)
    ws = torch.empty(_lib.dr_layernorm_backward_workspace_bytes(C), dtype=torch.uint8, device=x.device)
    check(_lib.dr_layernorm_backward_f32(rows, C, ptr(x), ptr(gamma.contiguous()), ptr(mean_rstd), ptr(grad_y), ptr(gx), ptr(gg), ptr(gb), ptr(ws), stream_of(x)))
    return gx, gg, gb


def attention(q, k, v, H, q_mask=None, k_mask=None):
    """fused softmax(q k^T / sqrt(d)) v per head: q [B,L,C], k / v [B,S,C] (token layout, head h in columns h d ..) -> [B,L,C]"""
    ensure_init()
    B, L, C = q.shape
    S, d = k.shape[1], C // H
    q, k, v = q.contiguous().float(), k.contiguous().float(), v.contiguous().float()
    out = torch.empty_like(q)
    check(_lib.dr_attention_f32(B, H, L, S, d, ptr(q), ptr(k), ptr(v), C, ptr(mask_u8(q_mask)), ptr(mask_u8(k_mask)), 1.0 / d ** 0.5, ptr(out), stream_of(q)))
    return out


def attention_backward(q, k, v, o, grad_o, H, q_mask=None, k_mask=None):
    """backward of attention(): -> (grad_q, grad_k, grad_v), fused (dr_attention_backward_f32)"""
    ensure_init()
    B, L, C = q.shape
    S, d = k.shape[1], C // H
    q, k, v, o, g = (t_.contiguous().float() for t_ in (q, k, v, o, grad_o))
    gq, gk, gv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    wsb = _lib.dr_attention_backward_workspace_bytes(B, H, L)
    ws = torch.empty(wsb, dtype=torch.uint8, device=q.device)
    check(_lib.dr_attention_backward_f32(B, H, L, S, d, ptr(q), ptr(k), ptr(v), ptr(o), ptr(g), C, ptr(mask_u8(q_mask)), ptr(mask_u8(k_mask)), 1.0 / d ** 0.5,
                                         ptr(gq), ptr(gk), ptr(gv), ptr(ws), wsb, stream_of(q)))
    return gq, gk, gv


def softmax_rows(scores, scale, q_mask=None, k_mask=None):
    """scores [B,H,L,S] -> softmax over S of scale * scores with the layer's key mask"""
    ensure_init()
    scores = scores.contiguous().float()
    B, H, L, S = scores.shape
    P = torch.empty_like(scores)
    qm, km = mask_u8(q_mask), mask_u8(k_mask)
    check(_lib.dr_softmax_rows_f32(B, H, L, S, ptr(scores), float(scale), ptr(qm), ptr(km), ptr(P), stream_of(scores)))
    return P


def softmax_backward(P, grad_P, scale):
    ensure_init()
    P, grad_P = P.contiguous(), grad_P.contiguous().float()
    out = torch.empty_like(P)
    check(_lib.dr_softmax_backward_f32(P.numel() // P.shape[-1], P.shape[-1], ptr(P), ptr(grad_P), float(scale), ptr(out), stream_of(P)))
    return out


def relu_backward(y, grad_y):
    ensure_init()
    y, grad_y = y.contiguous(), grad_y.contiguous().float()
    out = torch.empty_like(y)
    check(_lib.dr_relu_backward_f32(y.numel(), ptr(y), ptr(grad_y), ptr(out), stream_of(y)))
    return out


def rotary(x, cos_t, sin_t, inverse=False, scale=1.0):
    """embed_rotary on rows [rows, C] with half tables [rows, C/2] (inverse = its transpose)"""
    ensure_init()
    x = x.contiguous().float()
    out = torch.empty_like(x)
    rows, C = x.reshape(-1, x.shape[-1]).shape
    check(_lib.dr_rotary_f32(rows, C, ptr(x), ptr(cos_t.contiguous()), ptr(sin_t.contiguous()), 1 if inverse else 0, float(scale), ptr(out), stream_of(x)))
    return out


def focal_loss_backward(conf, conf_gt, alpha=0.25, gamma=2.0, pos_w=1.0, neg_w=1.0):
    """d loss / d conf of the sinkhorn-form focal loss (loss.py:311-314) -> [P,N,M] float32"""
    ensure_init()
    conf, conf_gt = conf.contiguous().float(), conf_gt.contiguous().float()
    P, N, M = conf.shape
    g = torch.empty_like(conf)
    ws = _train_ws(P, N, M, conf.device)
    check(_lib.dr_focal_loss_backward_f32(P, N, M, ptr(conf), ptr(conf_gt), alpha, gamma, pos_w, neg_w, ptr(g), ptr(ws), stream_of(conf)))
    return g


def sinkhorn_backward(scores, bin_score, iters, src_mask, tgt_mask, grad_conf):
    """backward of conf = exp(log_optimal_transport(scores, bin_score, iters, masks))[:, :-1, :-1] -> (grad_scores [P,N,M], grad_bin_score 0-d)"""
    ensure_init()
    scores, grad_conf = scores.contiguous().float(), grad_conf.contiguous().float()
    P, N, M = scores.shape
    gs = torch.empty_like(scores)
    ga = torch.empty(P, device=scores.device)
    wsb = _lib.dr_sinkhorn_backward_workspace_bytes(P, N, M, int(iters))
    ws = torch.empty(wsb, dtype=torch.uint8, device=scores.device)
    sm, tm = mask_u8(src_mask), mask_u8(tgt_mask)
    bs = torch.as_tensor(bin_score).detach().reshape(1).float().to(scores.device).contiguous()
    check(_lib.dr_sinkhorn_backward_f32(P, N, M, ptr(scores), ptr(sm), ptr(tm), ptr(bs), int(iters), ptr(grad_conf), ptr(gs), ptr(ga), ptr(ws), wsb,
                                        stream_of(scores)))
    return gs, ga.sum()


def mutual_match(conf, thr=0.0, mutual=True, cap=None, want_mask=False):
    """Matching.get_match on device: conf [P,N,M] float32 / float64 -> (matches [P,cap,3] int64, mconf [P,cap], count [P] int32,
    mask [P,N,M] uint8 or None); rows beyond count[p] are undefined, count > cap means the list was truncated."""
    ensure_init()
    conf = conf.contiguous()
    P, N, M = conf.shape
    cap = cap or (N + M)
    dev = conf.device
    matches = torch.zeros(P, cap, 3, dtype=torch.int64, device=dev)
    mconf = torch.zeros(P, cap, dtype=conf.dtype, device=dev)
    count = torch.zeros(P, dtype=torch.int32, device=dev)
    mask = torch.zeros(P, N, M, dtype=torch.uint8, device=dev) if want_mask else None
    fn = _lib.dr_mutual_match_f64 if conf.dtype == torch.float64 else _lib.dr_mutual_match_f32
    check(fn(P, N, M, ptr(conf), float(thr), 1 if mutual else 0, cap, ptr(matches), ptr(mconf), ptr(count), ptr(mask), stream_of(conf)))
    return matches, mconf, count, mask


def sinkhorn_f16(scores, bin_score, iters, src_mask=None, tgt_mask=None, *, minshift=False, apply_mask=False, ragged=False):
    """dr_sinkhorn_f16 (opt-in): [B,N,M] float16 scores -> float16 conf; tiles up to 256 x 256"""
    ensure_init()
    assert scores.dim() == 3 and scores.dtype == torch.float16
    scores = scores.contiguous()
    B, N, M = scores.shape
    flags = (SK_MINSHIFT if minshift else 0) | (SK_APPLY_MASK if apply_mask else 0) | (SK_RAGGED if ragged else 0)
    out = torch.empty(B, N, M, dtype=torch.float16, device=scores.device)
    bs = bin_score.detach().to(device=scores.device, dtype=torch.float32).reshape(1).contiguous()
    sm, tm = mask_u8(src_mask), mask_u8(tgt_mask)
    check(_lib.dr_sinkhorn_f16(B, N, M, ptr(scores), ptr(sm), ptr(tm), ptr(bs), int(iters), flags, ptr(out), stream_of(scores)))
    return out


def sinkhorn(scores, bin_score, iters, src_mask=None, tgt_mask=None, *, minshift=False, apply_mask=False,
             log_output=False, out_f32=False, strict=False, ragged=False, out=None):
    """Batched Sinkhorn with dustbins on [B,N,M] fp32/fp64 scores.

    Returns conf [B,N,M] (= exp(log_optimal_transport(...))[:, :-1, :-1]) or, with log_output, the full
    [B,N+1,M+1] log assignment (3D/models/matching.py:61-93)."""
    assert scores.dim() == 3
    scores = scores.contiguous()
    B, N, M = scores.shape
    f64 = scores.dtype == torch.float64
    if not f64 and scores.dtype != torch.float32:
        raise RuntimeError("sinkhorn: fp32 or fp64 scores only")
    flags = (SK_OUT_LOG if log_output else 0) | (SK_MINSHIFT if minshift else 0) | \
            (SK_APPLY_MASK if apply_mask else 0) | (SK_OUT_F32 if (f64 and out_f32) else 0) | (SK_STRICT if strict else 0) | \
            (SK_RAGGED if ragged else 0)
    odt = torch.float32 if (not f64 or out_f32) else torch.float64
    shape = (B, N + 1, M + 1) if log_output else (B, N, M)
    if out is None:
        out = torch.empty(shape, dtype=odt, device=scores.device)
    else:
        assert out.shape == shape and out.dtype == odt and out.is_contiguous()
    bs = bin_score.detach().to(device=scores.device, dtype=torch.float32).reshape(1).contiguous()
    wsb = _lib.dr_sinkhorn_workspace_bytes(B, N, M, 8 if f64 else 4, flags)
    ws = torch.empty(wsb, dtype=torch.uint8, device=scores.device) if wsb else None
    sm, tm = mask_u8(src_mask), mask_u8(tgt_mask)
    fn = _lib.dr_sinkhorn_f64 if f64 else _lib.dr_sinkhorn_f32
    check(fn(B, N, M, ptr(scores), ptr(sm), ptr(tm), ptr(bs), int(iters), flags, ptr(out), ptr(ws), wsb,
             stream_of(scores)))
    return out


# ------------------------------------------------------------------------------------------------
# thin wrappers of the remaining ops (all tensors on the ROCm device, float32 unless noted)
# ------------------------------------------------------------------------------------------------
def pe_freq(C, device):
    """div_term of VolumetricPositionEncoding.forward (position_encoding.py:58), computed by torch so
    that the table is the reference's own."""
    import math
    d = C // 3
    # evaluated on the host like the oracle/reference CPU run, then moved (a device exp may differ by an ulp)
    return torch.exp(torch.arange(0, d, 2, dtype=torch.float) * (-math.log(10000.0) / d)).to(device).contiguous()


def vol_pe(xyz, C, origin, voxel, R=None, t=None, rows_per_pair=None):
    """xyz [rows,3] -> (cos, sin) tables [rows, C/2] of the (optionally warped) points."""
    ensure_init()
    xyz = xyz.contiguous().float()
    rows = xyz.shape[0]
    cos = torch.empty(rows, C // 2, device=xyz.device)
    sin = torch.empty_like(cos)
    fr = pe_freq(C, xyz.device)
    if R is not None:
        R = R.contiguous().float()
        t = t.contiguous().float()
    check(_lib.dr_vol_pe_f32(rows, rows_per_pair or max(rows, 1), C, ptr(xyz), ptr(R), ptr(t), float(origin[0]),
                             float(origin[1]), float(origin[2]), float(voxel), ptr(fr), ptr(cos), ptr(sin), stream_of(xyz)))
    return cos, sin


def linear(x, W, epilogue=0, cos=None, sin=None, rot_C=0, scale=1.0):
    ensure_init()
    x = x.contiguous()
    W = W.contiguous()
    out = torch.empty(x.shape[0], W.shape[0], device=x.device)
    check(_lib.dr_linear_f32(x.shape[0], W.shape[0], x.shape[1], ptr(x), ptr(W), ptr(out), epilogue, ptr(cos), ptr(sin),
                             rot_C, float(scale), stream_of(x)))
    return out


def bmm_nt(a, b, scale=1.0):
    """a [..., R, K] @ b[..., N, K]^T over the leading (batch) dims, one launch (dr_gemm_nt_batched_f32; K is zero-padded to a multiple of 4)"""
    ensure_init()
    lead = a.shape[:-2]
    R, K = a.shape[-2:]
    N = b.shape[-2]
    a, b = a.reshape(-1, R, K).contiguous().float(), b.reshape(-1, N, K).contiguous().float()
    pad = (-K) % 4
    if pad:
        a, b = torch.nn.functional.pad(a, (0, pad)), torch.nn.functional.pad(b, (0, pad))
    nb = a.shape[0]
    out = torch.empty(nb, R, N, device=a.device)
    check(_lib.dr_gemm_nt_batched_f32(nb, R, N, K + pad, ptr(a), R * (K + pad), ptr(b), N * (K + pad), ptr(out), R * N, float(scale), stream_of(a)))
    return out.view(*lead, R, N)


def linear_ex(x, W, bias=None, epilogue=0, scale=1.0, K=None):
    """x [rows, lda >= K] @ W[ncols, K]^T (+ bias) -> [rows, ncols]"""
    ensure_init()
    K = W.shape[1] if K is None else K
    out = torch.empty(x.shape[0], W.shape[0], device=x.device)
    check(_lib.dr_linear_ex_f32(x.shape[0], W.shape[0], K, ptr(x), x.stride(0), ptr(W), ptr(bias), ptr(out), out.stride(0), epilogue,
                                float(scale), stream_of(x)))
    return out


KP_INFLUENCE = {"constant": 0, "linear": 1, "gaussian": 2}


def kpconv_gather(q_pts, s_pts, neighb_inds, x, kernel_points, extent, influence="linear", aggregation="sum"):
    """-> weighted features [Nq, ceil4(K*Cin)] (see dr_kpconv_gather_mode_f32; KP_influence / aggregation_mode of blocks.py:304-326)"""
    ensure_init()
    Nq, H = neighb_inds.shape
    K, Cin = kernel_points.shape[0], x.shape[1]
    ld = (K * Cin + 3) // 4 * 4
    out = torch.empty(Nq, ld, device=x.device)
    check(_lib.dr_kpconv_gather_mode_f32(Nq, s_pts.shape[0], H, Cin, K, ptr(q_pts.contiguous()), ptr(s_pts.contiguous()),
                                         ptr(neighb_inds.contiguous()), ptr(x.contiguous()), ptr(kernel_points.contiguous()), float(extent),
                                         KP_INFLUENCE[influence], {"sum": 0, "closest": 1}[aggregation], ptr(out), ld, stream_of(x)))
    return out


def col_stats(x):
    """per-column mean and 1/sqrt(var + 1e-5) over the rows of x [N, C] (InstanceNorm1d of BatchNormBlock)"""
    ensure_init()
    N, C = x.shape
    mean, rstd = torch.empty(C, device=x.device), torch.empty(C, device=x.device)
    wsb = _lib.dr_col_stats_workspace_bytes(N, C)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
    check(_lib.dr_col_stats_f32(N, C, ptr(x), x.stride(0), ptr(mean), ptr(rstd), ptr(ws), wsb, stream_of(x)))
    return mean, rstd


def norm_apply(a, stats_a, b=None, stats_b=None, slope=0.1, activate=True):
    """act( norm(a) + [norm(b) | b | 0] )"""
    ensure_init()
    N, C = a.shape
    out = torch.empty(N, C, device=a.device)
    mb, rb = stats_b if stats_b is not None else (None, None)
    check(_lib.dr_norm_apply_f32(N, C, ptr(a), a.stride(0), ptr(stats_a[0]), ptr(stats_a[1]), ptr(b), b.stride(0) if b is not None else 0,
                                 ptr(mb), ptr(rb), float(slope), 1 if activate else 0, ptr(out), C, stream_of(a)))
    return out


def gather_pool(x, inds, first_only=False):
    """max_pool / closest_pool over index lists [n2, H] (int64; indices >= len(x) read a zero row)"""
    ensure_init()
    n2, H = inds.shape
    out = torch.empty(n2, x.shape[1], device=x.device)
    inds = inds.contiguous()
    check(_lib.dr_gather_pool_f32(n2, H, H, x.shape[1], ptr(x.contiguous()), x.shape[0], ptr(inds), 1 if first_only else 0, ptr(out),
                                  stream_of(x)))
    return out


def kpconv_gather_backward(q_pts, s_pts, neighb_inds, x, kernel_points, extent, grad_weighted, influence="linear", aggregation="sum"):
    """d loss / d x [Ns, Cin] of kpconv_gather (grad_weighted [Nq, ceil4(K*Cin)])"""
    ensure_init()
    Nq, H = neighb_inds.shape
    K, Cin = kernel_points.shape[0], x.shape[1]
    gw = grad_weighted.contiguous()
    gx = torch.empty_like(x)
    check(_lib.dr_kpconv_gather_backward_mode_f32(Nq, s_pts.shape[0], H, Cin, K, ptr(q_pts.contiguous()), ptr(s_pts.contiguous()),
                                                  ptr(neighb_inds.contiguous()), ptr(x.contiguous()), ptr(kernel_points.contiguous()), float(extent),
                                                  KP_INFLUENCE[influence], {"sum": 0, "closest": 1}[aggregation], ptr(gw), gw.shape[1], ptr(gx),
                                                  stream_of(x)))
    return gx


def norm_backward(grad_out, out, a, stats_a, b=None, stats_b=None, slope=0.1, activate=True):
    """backward of norm_apply -> (grad_a, grad_b or None)"""
    ensure_init()
    N, C = a.shape
    g = grad_out.contiguous()
    ga = torch.empty(N, C, device=a.device)
    gb = torch.empty(N, C, device=a.device) if b is not None else None
    mb, rb = stats_b if stats_b is not None else (None, None)
    wsb = _lib.dr_norm_backward_workspace_bytes(N, C)
    ws = torch.empty(wsb, dtype=torch.uint8, device=a.device)
    check(_lib.dr_norm_backward_f32(N, C, ptr(g), C, ptr(out), out.stride(0) if out is not None else 0, ptr(a), a.stride(0), ptr(stats_a[0]), ptr(stats_a[1]),
                                    ptr(b), b.stride(0) if b is not None else 0, ptr(mb), ptr(rb), float(slope), 1 if activate else 0, ptr(ga), C,
                                    ptr(gb), C, ptr(ws), wsb, stream_of(a)))
    return ga, gb


def gather_pool_backward(x, inds, grad_out, first_only=False):
    """backward of gather_pool -> grad_x (shape of x)"""
    ensure_init()
    n2, H = inds.shape
    gx = torch.empty_like(x)
    check(_lib.dr_gather_pool_backward_f32(n2, H, H, x.shape[1], ptr(x.contiguous()), x.shape[0], ptr(inds.contiguous()), 1 if first_only else 0,
                                           ptr(grad_out.contiguous()), ptr(gx), stream_of(x)))
    return gx


_LAYER_KEYS = ("q_proj.weight", "k_proj.weight", "v_proj.weight", "merge.weight", "mlp.0.weight", "mlp.2.weight",
               "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias")


def layer_weights(tensors):
    """tensors: the 10 tensors of one GeometryAttentionLayer in _LAYER_KEYS order (kept alive by the caller)."""
    lw = LayerWeights()
    for (name, _), tns in zip(LayerWeights._fields_, tensors):
        assert tns.is_cuda and tns.is_contiguous() and tns.dtype == torch.float32
        setattr(lw, name, tns.data_ptr())
    return lw


def attention_layer(tensors, C, H, x, y, cos_x=None, sin_x=None, cos_y=None, sin_y=None, x_mask=None, y_mask=None, xq=None, yk=None):
    """x [P,Lx,C] attends y [P,Ly,C] (GeometryAttentionLayer.forward).  Rotary tables given: pe_type 'rotary'.  No tables: no position code inside
    the layer (entangled form), or -- with xq = x + pe_x, yk = y + pe_y -- pe_type 'sinusoidal' (transformero.py:50-57)."""
    ensure_init()
    P, Lx, _ = x.shape
    Ly = y.shape[1]
    lw = layer_weights(tensors)
    out = torch.empty_like(x)
    wsb = _lib.dr_attention_layer_workspace_bytes(P, Lx, Ly, C)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
    xm, ym = mask_u8(x_mask), mask_u8(y_mask)
    cont = lambda a: None if a is None else a.contiguous().float()
    x, y, xq, yk, cos_x, sin_x, cos_y, sin_y = (cont(a) for a in (x, y, xq, yk, cos_x, sin_x, cos_y, sin_y))
    check(_lib.dr_attention_layer_pe_f32(ctypes.byref(lw), C, H, P, Lx, Ly, ptr(x), ptr(y), ptr(xq), ptr(yk), ptr(cos_x), ptr(sin_x), ptr(cos_y),
                                         ptr(sin_y), ptr(xm), ptr(ym), ptr(out), ptr(ws), wsb, stream_of(x)))
    return out


def attention_layer_train_forward(tensors, C, H, x, y, cos_x, sin_x, cos_y, sin_y, x_mask=None, y_mask=None):
    """GeometryAttentionLayer.forward keeping what the backward needs: x [B,L,C], y [B,S,C] -> (out [B,L,C], saved: an opaque uint8 tensor)"""
    ensure_init()
    B, L, _ = x.shape
    S = y.shape[1]
    lw = layer_weights(tensors)
    out = torch.empty(B, L, C, device=x.device)
    nb = _lib.dr_attention_layer_train_saved_bytes(B, L, S, C)
    saved = torch.empty(nb, dtype=torch.uint8, device=x.device)
    xm, ym = mask_u8(x_mask), mask_u8(y_mask)
    check(_lib.dr_attention_layer_train_forward_f32(ctypes.byref(lw), C, H, B, L, S, ptr(x), ptr(y), ptr(cos_x), ptr(sin_x), ptr(cos_y), ptr(sin_y),
                                                    ptr(xm), ptr(ym), ptr(out), ptr(saved), nb, stream_of(x)))
    return out, saved


def attention_layer_backward(tensors, C, H, x, y, cos_x, sin_x, cos_y, sin_y, x_mask, y_mask, saved, grad_out):
    """-> grad_x [B,L,C], grad_y [B,S,C], the ten parameter gradients in the order of lib._LAYER_KEYS"""
    ensure_init()
    B, L, _ = x.shape
    S = y.shape[1]
    lw = layer_weights(tensors)
    grads = [torch.empty_like(t_) for t_ in tensors]
    gw = layer_weights(grads)
    gx, gy = torch.empty(B, L, C, device=x.device), torch.empty(B, S, C, device=x.device)
    wsb = _lib.dr_attention_layer_backward_workspace_bytes(B, H, L, S, C)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
    xm, ym = mask_u8(x_mask), mask_u8(y_mask)
    check(_lib.dr_attention_layer_backward_f32(ctypes.byref(lw), C, H, B, L, S, ptr(x), ptr(y), ptr(cos_x), ptr(sin_x), ptr(cos_y), ptr(sin_y),
                                               ptr(xm), ptr(ym), ptr(saved), ptr(grad_out), ptr(gx), ptr(gy), ctypes.byref(gw), ptr(ws), wsb,
                                               stream_of(x)))
    return gx, gy, grads


def dual_softmax(sim, temperature, src_mask=None, tgt_mask=None):
    """sim [P,N,M] float32 -> conf = softmax_dim1(sim / T | source rows) * softmax_dim2(sim / T | target columns) (matching.py:193-205)"""
    ensure_init()
    sim = sim.contiguous().float()
    P, N, M = sim.shape
    conf = torch.empty_like(sim)
    stats = torch.empty(2 * P * M, dtype=torch.float32, device=sim.device)
    sm, tm = mask_u8(src_mask), mask_u8(tgt_mask)
    check(_lib.dr_dual_softmax_f32(P, N, M, ptr(sim), float(temperature), ptr(sm), ptr(tm), ptr(conf), ptr(stats), stream_of(sim)))
    return conf


def dual_softmax_backward(sim, temperature, src_mask, tgt_mask, grad_conf):
    """d loss / d sim of conf = dual_softmax(sim, temperature, masks) given d loss / d conf (dr_dual_softmax_backward_f32)"""
    ensure_init()
    sim, grad_conf = sim.contiguous().float(), grad_conf.contiguous().float()
    P, N, M = sim.shape
    gs = torch.empty_like(sim)
    nb = _lib.dr_dual_softmax_backward_workspace_bytes(P, N, M)
    ws = torch.empty(nb, dtype=torch.uint8, device=sim.device)
    sm, tm = mask_u8(src_mask), mask_u8(tgt_mask)
    check(_lib.dr_dual_softmax_backward_f32(P, N, M, ptr(sim), float(temperature), ptr(sm), ptr(tm), ptr(grad_conf), ptr(gs), ptr(ws), nb, stream_of(sim)))
    return gs


def procrustes(conf, src_pcd, tgt_pcd, src_mask, tgt_mask, sample_rate, max_condition_num, use_mask_len=False,
               want_topk=False):
    """SoftProcrustesLayer.forward on device.  conf [P,N,M] float32 -> R,t,R_forwd,t_forwd,cond(f64),ok(bool)[,idx]."""
    ensure_init()
    if conf.dtype != torch.float32:
        raise RuntimeError("procrustes: float32 conf expected (the reference raises on float64, quirk Q3)")
    conf = conf.contiguous()
    P, N, M = conf.shape
    dev = conf.device
    R = torch.empty(P, 3, 3, device=dev); t = torch.empty(P, 3, 1, device=dev)
    Rf = torch.empty(P, 3, 3, device=dev); tf = torch.empty(P, 3, 1, device=dev)
    cond = torch.empty(P, dtype=torch.float64, device=dev)
    ok = torch.empty(P, dtype=torch.int32, device=dev)
    K = int(float(torch.tensor(float(max(N, M)), dtype=torch.float32) * sample_rate))
    idx = torch.zeros(P, K, dtype=torch.int32, device=dev) if want_topk else None     # (a pair's own K may be smaller -- 4D: the tail stays 0)
    sm, tm = mask_u8(src_mask), mask_u8(tgt_mask)
    wsb = _lib.dr_procrustes_workspace_bytes(P, N, M)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev) if wsb else None
    check(_lib.dr_procrustes_f32(P, N, M, ptr(conf), ptr(src_pcd.contiguous().float()), ptr(tgt_pcd.contiguous().float()),
                                 ptr(sm), ptr(tm), 1 if use_mask_len else 0, float(sample_rate), float(max_condition_num),
                                 ptr(R), ptr(t), ptr(Rf), ptr(tf), ptr(cond), ptr(ok), ptr(idx), ptr(ws), wsb, stream_of(conf)))
    res = (R, t, Rf, tf, cond, ok.bool())
    return res + (idx,) if want_topk else res


def procrustes_backward(conf, src_pcd, tgt_pcd, topk_idx, grad_R, grad_t, k_count=None):
    """d loss / d conf [P,N,M] of the weighted Procrustes fit on the entries `topk_idx` [P,K] (int32) selected by procrustes(..., want_topk=True)"""
    ensure_init()
    conf = conf.contiguous().float()
    P, N, M = conf.shape
    K = topk_idx.shape[1]
    g = torch.empty(P, N, M, device=conf.device)
    kc = k_count.to(device=conf.device, dtype=torch.int32).contiguous() if k_count is not None else None
    check(_lib.dr_procrustes_backward_f32(P, N, M, K, ptr(conf), ptr(src_pcd.contiguous().float()), ptr(tgt_pcd.contiguous().float()),
                                          ptr(topk_idx.contiguous()), ptr(kc), ptr(grad_R.contiguous().float().reshape(P, 9)),
                                          ptr(grad_t.contiguous().float().reshape(P, 3)), ptr(g), stream_of(conf)))
    return g


def top1_union(conf):
    """conf [P,N,M] (f32/f64) -> list of int64 [K_p,3] match tensors (one host sync for the counts)."""
    ensure_init()
    conf = conf.contiguous()
    P, N, M = conf.shape
    out = torch.empty(P, N + M, 3, dtype=torch.int64, device=conf.device)
    cnt = torch.empty(P, dtype=torch.int32, device=conf.device)
    f64 = conf.dtype == torch.float64
    fn = _lib.dr_top1_union_f64 if f64 else _lib.dr_top1_union_f32
    wsb = _lib.dr_top1_union_workspace_bytes(P, N, M, 8 if f64 else 4)
    ws = torch.empty(wsb, dtype=torch.uint8, device=conf.device) if wsb else None
    check(fn(P, N, M, ptr(conf), ptr(out), ptr(cnt), ptr(ws), wsb, stream_of(conf)))
    counts = cnt.cpu().tolist()
    return [out[p, :counts[p]] for p in range(P)]


# ------------------------------------------------------------------------------------------------
# evaluation harness.  matches int64 [P,cap,3], count int32 [P] (the loop's own output layout)
# ------------------------------------------------------------------------------------------------
def _f32c(x, shape):
    x = x.to(torch.float32).reshape(shape).contiguous()
    return x


def inlier_ratio(matches, count, s_pcd, t_pcd, rot, trn, inlier_thr, s2t_flow=None):
    """-> (ir [P] float32, n_inlier [P] int32)   (MatchMotionLoss.compute_inlier_ratio, 3D/models/loss.py:383-410)"""
    matches = matches.contiguous()
    P, cap, _ = matches.shape
    N, M = s_pcd.shape[1], t_pcd.shape[1]
    dev = matches.device
    ir = torch.empty(P, device=dev)
    n_inl = torch.empty(P, dtype=torch.int32, device=dev)
    flow = None if s2t_flow is None else _f32c(s2t_flow, (P, N, 3))
    check(_lib.dr_inlier_ratio_f32(P, cap, N, M, ptr(matches), ptr(count.contiguous()), ptr(_f32c(s_pcd, (P, N, 3))),
                                   ptr(_f32c(t_pcd, (P, M, 3))), ptr(_f32c(rot, (P, 9))), ptr(_f32c(trn, (P, 3))), ptr(flow),
                                   float(inlier_thr), ptr(ir), ptr(n_inl), stream_of(matches)))
    return ir, n_inl


def nrfmr(matches, count, s_pcd, t_pcd, raw_pcd, raw_flow, raw_offsets, metric_index, q_offsets, max_q, rot, trn,
          knn_radius=0.1, recall_thr=0.04, want_blended=False):
    """-> (nrfmr [P] float32, n_recalled [P] int32[, blended [sum Q,3]])   (compute_nrfmr, 3D/lib/tester.py:150-210)"""
    matches = matches.contiguous()
    P, cap, _ = matches.shape
    N, M = s_pcd.shape[1], t_pcd.shape[1]
    dev = matches.device
    out = torch.empty(P, device=dev)
    n_rec = torch.empty(P, dtype=torch.int32, device=dev)
    bl = torch.empty(metric_index.shape[0], 3, device=dev) if want_blended else None
    check(_lib.dr_nrfmr_f32(P, cap, N, M, ptr(matches), ptr(count.contiguous()), ptr(_f32c(s_pcd, (P, N, 3))),
                            ptr(_f32c(t_pcd, (P, M, 3))), ptr(_f32c(raw_pcd, (-1, 3))), ptr(_f32c(raw_flow, (-1, 3))),
                            ptr(raw_offsets.contiguous()), ptr(metric_index.contiguous()), ptr(q_offsets.contiguous()), int(max_q),
                            ptr(_f32c(rot, (P, 9))), ptr(_f32c(trn, (P, 3))), float(knn_radius), float(recall_thr), ptr(out),
                            ptr(n_rec), ptr(bl), stream_of(matches)))
    return (out, n_rec, bl) if want_blended else (out, n_rec)


def ransac_corr(matches, count, s_pcd, t_pcd, distance_thr=0.05, iters=50000, seed=0, pair_ids=None):
    """-> dict(rot [P,3,3], trn [P,3,1] float64, fitness [P], inlier_rmse [P], best_iter [P] int32)
    (ransac_regist_coarse / Open3D correspondence RANSAC, 3D/models/loss.py:13-24, 347-379)"""
    matches = matches.contiguous()
    P, cap, _ = matches.shape
    N, M = s_pcd.shape[1], t_pcd.shape[1]
    dev = matches.device
    f64 = dict(dtype=torch.float64, device=dev)
    rot, trn = torch.empty(P, 3, 3, **f64), torch.empty(P, 3, 1, **f64)
    fit, rmse = torch.empty(P, **f64), torch.empty(P, **f64)
    bi = torch.empty(P, dtype=torch.int32, device=dev)
    wsb = _lib.dr_ransac_workspace_bytes(P, cap, int(iters))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    ids = None if pair_ids is None else pair_ids.to(device=dev, dtype=torch.int64).contiguous()
    check(_lib.dr_ransac_corr_f64(P, cap, N, M, ptr(matches), ptr(count.contiguous()), ptr(_f32c(s_pcd, (P, N, 3))),
                                  ptr(_f32c(t_pcd, (P, M, 3))), float(distance_thr), int(iters), int(seed), ptr(ids), ptr(rot),
                                  ptr(trn), ptr(fit), ptr(rmse), ptr(bi), ptr(ws), wsb, stream_of(matches)))
    return dict(rot=rot, trn=trn, fitness=fit, inlier_rmse=rmse, best_iter=bi)


def registration_recall(rot_est, trn_est, rot_gt, trn_gt, info, thr=0.2):
    """-> (err [P] float64, success [P] int32)   (compute_registration_recall, 3D/models/loss.py:27-44, 415-448)"""
    P = rot_est.shape[0]
    dev = rot_est.device
    err = torch.empty(P, dtype=torch.float64, device=dev)
    ok = torch.empty(P, dtype=torch.int32, device=dev)
    d = lambda x, s: x.to(torch.float64).reshape(s).contiguous()
    check(_lib.dr_registration_recall_f64(P, ptr(d(rot_est, (P, 9))), ptr(d(trn_est, (P, 3))), ptr(_f32c(rot_gt, (P, 9))),
                                          ptr(_f32c(trn_gt, (P, 3))), ptr(d(info, (P, 36))), float(thr), ptr(err), ptr(ok),
                                          stream_of(rot_est)))
    return err, ok


# ------------------------------------------------------------------------------------------------
# collate-time ops: stacked clouds [n,3] float32 + lengths int32 [nb], all on the device
# ------------------------------------------------------------------------------------------------
def grid_subsample(points, lengths, dl):
    """-> (sub_points [n,3] buffer, sub_lengths [nb] int32, total [1] int32, status [1] int32), asynchronous; the first
    total[0] rows of the buffer are valid (cpp_subsampling.subsample_batch, grid_subsampling.cpp:4-211)"""
    points = points.to(torch.float32).contiguous()
    lengths = lengths.to(device=points.device, dtype=torch.int32).contiguous()
    n, nb = points.shape[0], lengths.shape[0]
    dev = points.device
    out = torch.empty(max(n, 1), 3, device=dev)
    ol = torch.empty(nb, dtype=torch.int32, device=dev)
    tot = torch.empty(1, dtype=torch.int32, device=dev)
    status = torch.empty(1, dtype=torch.int32, device=dev)
    wsb = _lib.dr_grid_subsample_workspace_bytes(n, nb)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    check(_lib.dr_grid_subsample_f32(n, nb, ptr(points), ptr(lengths), float(dl), ptr(out), ptr(ol), ptr(tot), ptr(status), ptr(ws),
                                     wsb, stream_of(points)))
    return out, ol, tot, status


def radius_neighbors(queries, supports, q_lengths, s_lengths, radius, limit):
    """-> (neighbors int64 [nq, limit], max_count [1] int32, status [1] int32), asynchronous
    (cpp_neighbors.batch_query + [:, :limit], neighbors.cpp:210-333)"""
    queries = queries.to(torch.float32).contiguous()
    supports = supports.to(torch.float32).contiguous()
    dev = queries.device
    ql = q_lengths.to(device=dev, dtype=torch.int32).contiguous()
    sl = s_lengths.to(device=dev, dtype=torch.int32).contiguous()
    nq, ns, nb = queries.shape[0], supports.shape[0], ql.shape[0]
    out = torch.empty(nq, limit, dtype=torch.int64, device=dev)
    mc = torch.empty(1, dtype=torch.int32, device=dev)
    status = torch.empty(1, dtype=torch.int32, device=dev)
    wsb = _lib.dr_radius_neighbors_workspace_bytes(nq, ns, nb)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    check(_lib.dr_radius_neighbors_f32(nq, ns, nb, ptr(queries), ptr(supports), ptr(ql), ptr(sl), float(radius), int(limit), ptr(out),
                                       ptr(mc), ptr(status), ptr(ws), wsb, stream_of(queries)))
    return out, mc, status


def batch_mutual_topk_select(score_mat, k, row_masks=None, col_masks=None, largest=True, threshold=None, mutual=True):
    """-> (batch_indices, row_indices, col_indices, scores) like vision3d.ops.batch_mutual_topk_select(..., reduce_result=True)
    (Diff-Reg-2d3d/vision3d/ops/mutual_topk_select.py:63-134); a 2-D score_mat is one batch element (mutual_topk_select :7-60,
    returns (row_indices, col_indices, scores)).  One host sync for the number of correspondences."""
    two_d = score_mat.dim() == 2
    s = (score_mat[None] if two_d else score_mat).to(torch.float32).contiguous()
    B, N, M = s.shape
    if k > N or k > M:
        raise RuntimeError("selected index k out of range")      # what torch.topk says in the reference (mutual_topk_select.py:27-28)
    dev = s.device
    cap = B * (min(N, M) if mutual else (N + M)) * k
    idx = torch.empty(max(cap, 1), 3, dtype=torch.int64, device=dev)
    sc = torch.empty(max(cap, 1), device=dev)
    tot = torch.empty(1, dtype=torch.int32, device=dev)
    wsb = _lib.dr_mutual_topk_workspace_bytes(B, N, M)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    check(_lib.dr_mutual_topk_select_f32(B, N, M, ptr(s), int(k), 1 if largest else 0, 0 if threshold is None else 1,
                                         0.0 if threshold is None else float(threshold), 1 if mutual else 0, ptr(mask_u8(row_masks)),
                                         ptr(mask_u8(col_masks)), ptr(idx), ptr(sc), cap, ptr(tot), ptr(ws), wsb, stream_of(s)))
    n = int(tot.item())
    assert n <= cap
    idx, sc = idx[:n], sc[:n]
    if two_d:
        return idx[:, 1], idx[:, 2], sc
    return idx[:, 0], idx[:, 1], idx[:, 2], sc


def pnp_ransac(points, pixels, intrinsics, num_iterations=50000, distance_tolerance=8.0, seed=0, transposed=True):
    """PnP-RANSAC on device tensors -> dict(transform [4,4] float64 device, n_inlier, best_iter) (None with fewer than 4 correspondences)"""
    ensure_init()
    n = points.shape[0]
    if n < 4:
        return None
    points, pixels = points.contiguous().float(), pixels.contiguous().float()
    K = (ctypes.c_double * 9)(*[float(x) for x in torch.as_tensor(intrinsics, dtype=torch.float64).reshape(-1).tolist()])
    T = torch.empty(4, 4, dtype=torch.float64, device=points.device)
    ni = torch.zeros(1, dtype=torch.int32, device=points.device)
    bi = torch.zeros(1, dtype=torch.int32, device=points.device)
    wsb = _lib.dr_pnp_ransac_workspace_bytes(n, int(num_iterations))
    ws = torch.empty(wsb, dtype=torch.uint8, device=points.device)
    check(_lib.dr_pnp_ransac_f64(n, ptr(points), ptr(pixels), 1 if transposed else 0, K, int(num_iterations), float(distance_tolerance), int(seed), ptr(T),
                                 ptr(ni), ptr(bi), ptr(ws), wsb, stream_of(points)))
    return dict(transform=T, n_inlier=ni, best_iter=bi)


def patch_similarity(img_feats, img_knn_indices, pcd_feats, pcd_knn_indices):
    """[P,Ki,Kc] similarity of the image / point patches of P node correspondences (EXP/model.py:726-738); an index equal to
    pcd_feats.shape[0] addresses the zero row the reference appends"""
    ensure_init()
    img_feats, pcd_feats = img_feats.contiguous().float(), pcd_feats.contiguous().float()
    ii, pj = img_knn_indices.to(torch.int64).contiguous(), pcd_knn_indices.to(torch.int64).contiguous()
    P, Ki = ii.shape
    Kc = pj.shape[1]
    out = torch.empty(P, Ki, Kc, device=img_feats.device)
    check(_lib.dr_patch_similarity_f32(P, Ki, Kc, img_feats.shape[1], ptr(img_feats), ptr(ii), ptr(pcd_feats), ptr(pj), pcd_feats.shape[0], ptr(out),
                                       stream_of(out)))
    return out


def unique_pairs(first, second, multiplier):
    """sorted distinct first * multiplier + second (torch.unique of EXP/model.py:760-761) -> (keys [n] int64, count [1] int32) without a host sync"""
    ensure_init()
    a, b = first.to(torch.int64).contiguous(), second.to(torch.int64).contiguous()
    n = a.numel()
    keys = torch.empty(max(n, 1), dtype=torch.int64, device=a.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=a.device)
    wsb = _lib.dr_unique_pairs_workspace_bytes(n)
    ws = torch.empty(wsb, dtype=torch.uint8, device=a.device)
    check(_lib.dr_unique_pairs_i64(n, ptr(a), ptr(b), int(multiplier), ptr(keys), ptr(cnt), ptr(ws), wsb, stream_of(a)))
    return keys, cnt


def corr_gather(keys, count, num_points_f, img_points_f, img_pixels_f, pcd_points_f, pcd_pixels_f, img_feats_f, pcd_feats_f):
    """EXP/model.py:761-774 -> dict with the reference's output names (one host read of the count to size the views)"""
    ensure_init()
    f = lambda x, w: x.contiguous().float().view(-1, w)
    ip, ix, pp, px = f(img_points_f, 3), f(img_pixels_f, 2), f(pcd_points_f, 3), f(pcd_pixels_f, 2)
    fi, fp = img_feats_f.contiguous().float(), pcd_feats_f.contiguous().float()
    cap = keys.numel()
    dev = keys.device
    o = dict(img_corr_indices=torch.empty(cap, dtype=torch.int64, device=dev), pcd_corr_indices=torch.empty(cap, dtype=torch.int64, device=dev),
             img_corr_points=torch.empty(cap, 3, device=dev), img_corr_pixels=torch.empty(cap, 2, device=dev),
             pcd_corr_points=torch.empty(cap, 3, device=dev), pcd_corr_pixels=torch.empty(cap, 2, device=dev), corr_scores=torch.empty(cap, device=dev))
    check(_lib.dr_corr_gather_f32(cap, ptr(count), ptr(keys), int(num_points_f), fi.shape[1], ptr(ip), ptr(ix), ptr(pp), ptr(px), ptr(fi), ptr(fp),
                                  ptr(o["img_corr_indices"]), ptr(o["pcd_corr_indices"]), ptr(o["img_corr_points"]), ptr(o["img_corr_pixels"]),
                                  ptr(o["pcd_corr_points"]), ptr(o["pcd_corr_pixels"]), ptr(o["corr_scores"]), stream_of(keys)))
    n = int(count.item())
    return {k: v[:n] for k, v in o.items()}


PROF_KINDS = ("gemm", "attention", "layernorm", "position_code", "sinkhorn", "procrustes", "state", "gemm_split")


def prof_enable(on=True):
    _lib.dr_prof_enable(1 if on else 0)


def prof_collect():
    """-> {family: (calls, total_ms, work)}; synchronises the device."""
    n = len(PROF_KINDS)
    calls = (ctypes.c_int * n)(); ms = (ctypes.c_double * n)(); work = (ctypes.c_double * n)()
    check(_lib.dr_prof_collect(calls, ms, work))
    return {k: (calls[i], ms[i], work[i]) for i, k in enumerate(PROF_KINDS)}
