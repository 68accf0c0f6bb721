"""ctypes declarations for the CLIP-forward part of the C ABI (include/centerclip_hip.h)."""
import ctypes as c

CC_MAX_LAYERS = 32
EPI = {"f16": 0, "f16_gelu": 1, "f32_resid": 2, "f32": 4}


class BlockWeights(c.Structure):
    """struct cc_block_weights"""
    _fields_ = [(n, c.c_void_p) for n in (
        "ln_1_weight", "ln_1_bias", "in_proj_weight_f16", "in_proj_bias", "out_proj_weight_f16", "out_proj_bias",
        "ln_2_weight", "ln_2_bias", "c_fc_weight_f16", "c_fc_bias", "c_proj_weight_f16", "c_proj_bias",
        "in_proj_ln_weight_f16", "in_proj_ln_c1", "in_proj_ln_c2", "c_fc_ln_weight_f16", "c_fc_ln_c1", "c_fc_ln_c2")]


class VitModel(c.Structure):
    """struct cc_vit_model"""
    _fields_ = [("layers", c.c_int32), ("width", c.c_int32), ("heads", c.c_int32), ("patch", c.c_int32),
                ("resolution", c.c_int32), ("embed_dim", c.c_int32),
                ("conv1_weight_f16", c.c_void_p), ("class_embedding", c.c_void_p),
                ("positional_embedding", c.c_void_p), ("ln_pre_weight", c.c_void_p), ("ln_pre_bias", c.c_void_p),
                ("ln_post_weight", c.c_void_p), ("ln_post_bias", c.c_void_p), ("proj", c.c_void_p),
                ("blocks", c.POINTER(BlockWeights)),
                ("cluster_frames", c.c_int32 * CC_MAX_LAYERS), ("cluster_tokens", c.c_int32 * CC_MAX_LAYERS),
                ("cluster_metric", c.c_int32), ("cluster_norm_p", c.c_float), ("cluster_threshold", c.c_float),
                ("cluster_iter_limit", c.c_int32), ("cluster_split_size", c.c_int32), ("cluster_pre_norm", c.c_int32),
                ("cluster_variants", c.c_void_p), ("row_policy", c.c_int32), ("conv2_weight_f16", c.c_void_p)]


class Frames(c.Structure):
    """cc_frames (include/centerclip_hip.h): format 0 = fp32 CHW (normalised), 1 = uint8 CHW, 2 = uint8 HWC."""
    _fields_ = [("data", c.c_void_p), ("format", c.c_int32), ("mean", c.c_float * 3), ("std", c.c_float * 3)]


class TextModel(c.Structure):
    """struct cc_text_model"""
    _fields_ = [("layers", c.c_int32), ("width", c.c_int32), ("heads", c.c_int32), ("context_length", c.c_int32),
                ("vocab_size", c.c_int32), ("embed_dim", c.c_int32),
                ("token_embedding", c.c_void_p), ("positional_embedding", c.c_void_p),
                ("ln_final_weight", c.c_void_p), ("ln_final_bias", c.c_void_p), ("text_projection", c.c_void_p),
                ("blocks", c.POINTER(BlockWeights)), ("row_policy", c.c_int32)]


ROWS_ALL_TEXT, ROWS_ALL_LAST_BLOCK = 1, 2          # CC_ROWS_* (include/centerclip_hip.h)


def declare(lib):
    vp, i32, i64, f32, sz = c.c_void_p, c.c_int32, c.c_int64, c.c_float, c.c_size_t
    lib.cc_linear_f16.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.cc_layernorm_f32.argtypes = [vp, i64, vp, vp, vp, i64, i32, i32, f32, i32, vp]
    lib.cc_attention_f16.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
    lib.cc_fold_layernorm_linear_f32.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, vp, vp]
    lib.cc_fold_layernorm_linear_f32.restype = c.c_int
    lib.cc_row_stats_f16.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    lib.cc_linear_ln_f16.argtypes = [vp, vp, vp, vp, vp, i32, f32, vp, i32, i32, i32, i32, i32, vp]
    lib.cc_linear_resid_stats_f16.argtypes = [vp, vp, vp, vp, vp, vp, c.POINTER(i32), vp, vp, i32, vp, i32, i32, i32, i32, vp]
    lib.cc_inproj_attention_f16.argtypes = [vp, vp, vp, vp, vp, i32, f32, vp, i32, i32, i32, i32, vp, vp, vp, vp]
    lib.cc_inproj_attention_f16.restype = c.c_int
    lib.cc_linear_tile_for.argtypes = [i32, i32, i32, i32]
    lib.cc_linear_tile_for.restype = c.c_int
    lib.cc_linear_resid_stats_slots.argtypes = [i32, i32, i32, i32]
    lib.cc_linear_resid_stats_slots.restype = c.c_int
    lib.cc_attention_strided_f16.argtypes = [vp, vp, i32, i32, i32, i32, i32, i64, i64, vp]
    lib.cc_attention_strided_f16.restype = c.c_int
    lib.cc_text_encode_hidden.argtypes = [c.POINTER(TextModel), vp, i32, i32, vp, vp, vp, sz, vp]
    lib.cc_text_encode_hidden.restype = c.c_int
    lib.cc_head_project_f32.argtypes = [vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, vp]
    lib.cc_head_project_f32.restype = c.c_int
    lib.cc_contrastive_loss_f32.argtypes = [vp, i32, i64, i64, vp, vp, sz, vp]
    lib.cc_contrastive_loss_f32.restype = c.c_int
    lib.cc_contrastive_grad_workspace_bytes.argtypes = [i32, i32, i32]
    lib.cc_contrastive_grad_workspace_bytes.restype = sz
    lib.cc_contrastive_loss_grad_f32.argtypes = [vp, vp, vp, i64, i64, i32, i32, i32, f32, f32, vp, vp, vp, vp, vp, sz, vp]
    lib.cc_contrastive_loss_grad_f32.restype = c.c_int
    lib.cc_contrastive_loss_grad_dev_f32.argtypes = [vp, vp, vp, i64, i64, i32, i32, i32, f32, vp, f32, vp, vp, vp, vp, vp, sz, vp]
    lib.cc_contrastive_loss_grad_dev_f32.restype = c.c_int
    lib.cc_normalize_rows_f32.argtypes = [vp, vp, i32, i32, vp]
    lib.cc_normalize_rows_f32.restype = c.c_int
    lib.cc_loose_similarity_grouped_f32.argtypes = [vp, vp, vp, i32, i64, i64, i64, i64, i32, i32, i32, i32, f32, vp, i32,
                                                    vp, vp, sz, vp]
    lib.cc_loose_similarity_grouped_f32.restype = c.c_int
    for name in ("cc_row_stats_f16", "cc_linear_ln_f16", "cc_linear_resid_stats_f16"):
        getattr(lib, name).restype = c.c_int
    lib.cc_rank_counts_f32.argtypes = [vp, i32, i32, i64, i64, i32, vp, vp]
    lib.cc_rank_counts_f32.restype = c.c_int
    lib.cc_rank_counts_cols_f32.argtypes = [vp, i32, i32, i64, i64, vp, vp, vp]
    lib.cc_rank_counts_cols_f32.restype = c.c_int
    lib.cc_rank_counts_ref_f32.argtypes = [vp, i32, i32, i64, i64, vp, vp, vp]
    lib.cc_rank_counts_ref_f32.restype = c.c_int
    lib.cc_group_max_rows_f32.argtypes = [vp, i32, i32, i64, vp, i32, vp, vp]
    lib.cc_group_max_rows_f32.restype = c.c_int
    lib.cc_vit_workspace_bytes.restype = sz
    lib.cc_vit_workspace_bytes.argtypes = [c.POINTER(VitModel), i32, i32]
    lib.cc_vit_forced_medoids_count.restype = i64
    lib.cc_vit_forced_medoids_count.argtypes = [c.POINTER(VitModel), i32]
    lib.cc_vit_encode.argtypes = [c.POINTER(VitModel), vp, i32, i32, vp, vp, vp, vp, vp, sz, vp]
    lib.cc_vit_encode_frames.argtypes = [c.POINTER(VitModel), c.POINTER(Frames), i32, i32, vp, vp, vp, vp, vp, sz, vp]
    lib.cc_vit_encode_frames.restype = c.c_int
    lib.cc_clip_encode_frames.argtypes = [c.POINTER(VitModel), c.POINTER(Frames), i32, i32, vp, vp, vp,
                                          c.POINTER(TextModel), vp, i32, i32, vp, vp, sz, vp]
    lib.cc_clip_encode_frames.restype = c.c_int
    lib.cc_token_gather_f32.argtypes = [vp, i64, i64, i32, i32, i32, i32, i32, i32, vp, vp, i64, i64, vp]
    lib.cc_token_gather_f32.restype = c.c_int
    lib.cc_text_workspace_bytes.restype = sz
    lib.cc_text_workspace_bytes.argtypes = [c.POINTER(TextModel), i32, i32]
    lib.cc_text_encode.argtypes = [c.POINTER(TextModel), vp, i32, i32, vp, vp, sz, vp]
    lib.cc_clip_workspace_bytes.restype = sz
    lib.cc_clip_workspace_bytes.argtypes = [c.POINTER(VitModel), i32, i32, c.POINTER(TextModel), i32, i32]
    lib.cc_clip_encode.argtypes = [c.POINTER(VitModel), vp, i32, i32, vp, vp, c.POINTER(TextModel), vp, i32, i32, vp, vp, sz, vp]
    lib.cc_clip_encode.restype = c.c_int
    lib.cc_similarity_workspace_bytes.restype = sz
    lib.cc_layernorm_backward_workspace_bytes.argtypes = [i32, i32]
    lib.cc_layernorm_backward_workspace_bytes.restype = sz
    lib.cc_layernorm_backward_f32.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp, vp, sz, vp]
    lib.cc_layernorm_backward_f32.restype = c.c_int
    lib.cc_quick_gelu_f16.argtypes = [vp, vp, i64, vp]
    lib.cc_quick_gelu_f16.restype = c.c_int
    lib.cc_quick_gelu_backward_f16.argtypes = [vp, vp, vp, i64, vp, vp]
    lib.cc_quick_gelu_backward_f16.restype = c.c_int
    lib.cc_attention_backward_f16.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, sz, vp]
    lib.cc_attention_backward_workspace_bytes.argtypes = [i32, i32, i32]
    lib.cc_attention_backward_workspace_bytes.restype = sz
    lib.cc_attention_backward_f16.restype = c.c_int
    lib.cc_column_sums_workspace_bytes.argtypes = [i32, i32]
    lib.cc_column_sums_workspace_bytes.restype = sz
    lib.cc_column_sums_f32.argtypes = [vp, i32, i32, vp, vp, sz, vp]
    lib.cc_column_sums_f32.restype = c.c_int
    lib.cc_cast_scaled_f16.argtypes = [vp, vp, i64, vp, vp, vp]
    lib.cc_cast_scaled_f16.restype = c.c_int
    lib.cc_unscale_f32.argtypes = [vp, i64, vp, vp, vp]
    lib.cc_unscale_f32.restype = c.c_int
    lib.cc_linear_unscaled_f16.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp]
    lib.cc_linear_unscaled_f16.restype = c.c_int
    lib.cc_cast_transpose_f16.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, sz, vp]
    lib.cc_cast_transpose_colsum_workspace_bytes.argtypes = [i32, i32]
    lib.cc_cast_transpose_colsum_workspace_bytes.restype = sz
    lib.cc_cast_transpose_f16.restype = c.c_int
    lib.cc_linear_resid_f16.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.cc_linear_resid_f16.restype = c.c_int
    lib.cc_wgrad_tn_workspace_bytes.restype = sz
    lib.cc_wgrad_tn_workspace_bytes.argtypes = [i32, i32, i32]
    lib.cc_wgrad_tn_f16.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, i32, vp, vp, sz, vp]
    lib.cc_wgrad_tn_f16.restype = c.c_int
    lib.cc_bertadam_workspace_bytes.argtypes = []
    lib.cc_bertadam_workspace_bytes.restype = sz
    lib.cc_bertadam_multi_f32.argtypes = [vp, i32, f32, f32, f32, f32, vp]
    lib.cc_bertadam_multi_f32.restype = c.c_int
    lib.cc_bertadam_norm_blocks.argtypes = [i64]
    lib.cc_bertadam_norm_blocks.restype = i32
    lib.cc_bertadam_step_blocks.argtypes = [i64]
    lib.cc_bertadam_step_blocks.restype = i32
    lib.cc_bertadam_multi_large_f32.argtypes = [vp, i32, i32, i32, f32, f32, f32, f32, vp, sz, vp]
    lib.cc_bertadam_multi_large_f32.restype = c.c_int
    lib.cc_bertadam_step_f32.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, f32, vp, vp, sz, vp]
    lib.cc_bertadam_step_f32.restype = c.c_int
    lib.cc_similarity_plane_row_bytes.argtypes = [i32]
    lib.cc_similarity_plane_row_bytes.restype = sz
    lib.cc_similarity_padded_rows.argtypes = [i32]
    lib.cc_similarity_padded_rows.restype = i32
    lib.cc_normalize_rows_planes_f32.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    lib.cc_normalize_rows_planes_f32.restype = c.c_int
    lib.cc_video_pool_normalize_planes_f32.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
    lib.cc_video_pool_normalize_planes_f32.restype = c.c_int
    lib.cc_scaled_dot_planes_f32.argtypes = [vp, vp, i32, i32, i32, i32, f32, vp, i32, vp]
    lib.cc_scaled_dot_planes_f32.restype = c.c_int
    lib.cc_scaled_dot_planes_products_f32.argtypes = [vp, vp, i32, i32, i32, i32, f32, i32, vp, i32, vp]
    lib.cc_scaled_dot_planes_products_f32.restype = c.c_int
    lib.cc_similarity_workspace_bytes.argtypes = [i32, i32, i32]
    lib.cc_video_pool_normalize_f32.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    lib.cc_loose_similarity_f32.argtypes = [vp, vp, vp, i32, i32, i32, i32, f32, vp, i32, vp, vp, sz, vp]
    lib.cc_loose_similarity_strided_f32.argtypes = [vp, vp, vp, c.c_int64, c.c_int64, i32, i32, i32, i32, f32, vp, i32, vp,
                                                    vp, sz, vp]
    lib.cc_loose_similarity_strided_f32.restype = c.c_int
    lib.cc_scaled_dot_nt_f32.argtypes = [vp, vp, i32, i32, i32, f32, vp, i32, vp, sz, vp]
    for name in ("cc_linear_f16", "cc_layernorm_f32", "cc_attention_f16", "cc_vit_encode", "cc_text_encode",
                 "cc_video_pool_normalize_f32", "cc_loose_similarity_f32", "cc_scaled_dot_nt_f32"):
        getattr(lib, name).restype = c.c_int
