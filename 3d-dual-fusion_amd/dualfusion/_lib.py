"""ctypes binding of libdf3d_hip.so (the C ABI declared in include/df3d_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or cannot be loaded
every op raises.  (The CPU restatement in oracle/ is test infrastructure and is never
imported from here.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libdf3d_hip.so")

c_int = ctypes.c_int
c_uint = ctypes.c_uint
c_longlong = ctypes.c_longlong
c_size_t = ctypes.c_size_t
c_float = ctypes.c_float
c_void_p = ctypes.c_void_p
c_ulonglong = ctypes.c_ulonglong
c_char_p = ctypes.c_char_p

# name -> (restype, argtypes); mirrors include/df3d_hip.h one to one
SIGNATURES = {
    "df3d_version": (c_int, []),
    "df3d_last_error": (c_char_p, []),
    "df3d_device_count": (c_int, []),
    "df3d_device_arch": (c_int, [c_char_p, c_int]),
    "df3d_split_overflow": (c_int, [c_int, c_char_p, c_int]),
    "df3d_split_overflow_collect": (c_int, [c_void_p, c_int, c_void_p]),
    "df3d_split_overflow_units": (c_int, [c_char_p, c_int]),
    "df3d_hard_voxelize_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "df3d_hard_voxelize": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "df3d_hard_voxelize_batched": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                           c_void_p]),
    "df3d_grid_bytes": (c_size_t, [c_int, c_void_p]),
    "df3d_grid_build": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "df3d_subm_neighbors": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    "df3d_conv_out_indices": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_void_p, c_void_p]),
    "df3d_conv_neighbors": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p]),
    "df3d_conv_transpose_out_indices": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_void_p, c_void_p]),
    "df3d_conv_transpose_neighbors": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_void_p]),
    "df3d_nbr_to_pairs_workspace_bytes": (c_size_t, [c_int, c_int]),
    "df3d_nbr_to_pairs": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "df3d_pairs_to_nbr": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "df3d_sparse_conv_fused": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "df3d_nbr_row_lists_bytes": (c_size_t, [c_int, c_int]),
    "df3d_nbr_row_lists": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "df3d_sparse_conv_fused_lists": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "df3d_sparse_conv_grouped": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int,
                                         c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "df3d_conv_tile_count": (c_int, [c_int, c_int, c_int, c_int]),
    "df3d_conv_tiles_workspace_bytes": (c_size_t, [c_int]),
    "df3d_conv_tiles": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "df3d_sparse_conv_fused_tiled": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                             c_void_p]),
    "df3d_timing_begin": (c_int, []),
    "df3d_timing_end": (c_int, []),
    "df3d_timing_get": (c_int, [c_int, c_void_p, c_void_p]),
    "df3d_sparse_to_dense": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "df3d_ms_deform_attn_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "df3d_ms_deform_attn_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                             c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "df3d_ms_deform_attn_backward_binned_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "df3d_ms_deform_attn_backward_binned_slab_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "df3d_ms_deform_attn_backward_binned": (c_int, [c_void_p] * 6 + [c_int] * 7 + [c_void_p] * 4 + [c_size_t, c_void_p, c_void_p]),
    "df3d_furthest_point_sample": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "df3d_pe_gather_add": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int,
                                   c_void_p, c_void_p]),
    "df3d_group_attention": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "df3d_group_attention_split": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "df3d_ball_query": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_int, c_void_p, c_void_p]),
    "df3d_group_points": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "df3d_gather_points": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "df3d_project_voxels": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p]),
    "df3d_scatter_to_image": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                      c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "df3d_assemble_queries": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "df3d_scatter_winner": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "df3d_gate_scatter_rows": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                       c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "df3d_gate_scatter": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                  c_void_p, c_void_p, c_int, c_void_p]),
    "df3d_gate_finish": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "df3d_query_pixel_rows_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "df3d_query_pixel_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                      c_void_p, c_size_t, c_void_p]),
    "df3d_assemble_queries2_compact": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "df3d_imgproj_split_compact": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p]),
    "df3d_gate_rows": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "df3d_gate_finish_bias": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "df3d_fusion_writeback_split": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                            c_void_p, c_void_p, c_void_p]),
    "df3d_sparse_to_dense_rows_split": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "df3d_query_slots": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "df3d_assemble_queries2": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "df3d_assemble_queries2_slots": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_void_p]),
    "df3d_conv_packed_weight_bytes": (c_size_t, [c_int, c_int, c_int]),
    "df3d_conv_pack_weights": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "df3d_bn_rows_supported": (c_int, [c_int]),
    "df3d_bn_rows_scratch_doubles": (c_int, [c_int]),
    "df3d_bn_rows_forward": (c_int, [c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_float, c_float, c_int, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "df3d_bn_rows_backward": (c_int, [c_void_p, c_void_p, c_longlong, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p]),
    "df3d_conv_pack_weights_groups": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "df3d_split_rows_scaled": (c_int, [c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "df3d_split_rows": (c_int, [c_void_p, c_longlong, c_int, c_void_p, c_void_p]),
    "df3d_sparse_conv_split": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                       c_void_p]),
    "df3d_ffn_packed_bytes": (c_size_t, [c_int, c_int]),
    "df3d_ffn_pack": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "df3d_ffn_fused": (c_int, [c_void_p, c_longlong, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_float, c_void_p, c_void_p]),
    "df3d_ffn_set_precision": (c_int, [c_int]),
    "df3d_ffn_fused_jobs": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "df3d_timing_count_pairs": (c_int, [c_int]),
    "df3d_timing_filter": (c_int, [c_int, c_int, c_int]),
    "df3d_timing_sample": (c_int, [c_int]),
    "df3d_timing_get2": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "df3d_sparse_to_dense_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "df3d_conv2d_neighbors": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "df3d_rows_to_bf16": (c_int, [c_void_p, ctypes.c_longlong, c_int, c_void_p, c_void_p]),
    "df3d_rows_from_bf16": (c_int, [c_void_p, ctypes.c_longlong, c_int, c_void_p, c_void_p]),
    "df3d_conv_packed_weight_bytes_bf16": (c_size_t, [c_int, c_int, c_int]),
    "df3d_conv_pack_weights_bf16": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "df3d_sparse_conv_bf16": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "df3d_invert_neighbors": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "df3d_chanfirst_dot": (c_int, [c_void_p, c_void_p, c_int, c_int, c_longlong, c_void_p, c_void_p]),
    "df3d_sparse_conv_grad_filters_scaled": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p,
                                                    c_void_p, c_void_p]),
    "df3d_sparse_conv_grad_filters_bf16": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "df3d_rows_grad_weights_scaled": (c_int, [c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "df3d_pow2_scale_floats": (c_int, []),
    "df3d_rows_pow2_scale": (c_int, [c_void_p, c_longlong, c_void_p, c_void_p]),
    "df3d_rows_grad_weights": (c_int, [c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p, c_void_p]),
    "df3d_sparse_conv_grad_filters": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p,
                                              c_void_p]),
    "df3d_conv_rows_split": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int,
                                     c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "df3d_conv_packed_weight_bytes3": (c_size_t, [c_int, c_int, c_int]),
    "df3d_conv_pack_weights3": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "df3d_split_rows3": (c_int, [c_void_p, ctypes.c_longlong, c_int, c_void_p, c_void_p]),
    "df3d_conv_rows_split3": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                      c_void_p]),
    "df3d_boxes_bev_pairwise": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "df3d_nms_bev_workspace_bytes": (c_size_t, [c_int, c_int]),
    "df3d_nms_bev": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_int, c_void_p, c_void_p, c_void_p,
                             c_size_t, c_void_p]),
    "df3d_centerhead_predict_workspace_bytes": (c_size_t, [c_int, c_void_p]),
    "df3d_centerhead_predict": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_size_t, c_void_p]),
    "df3d_head_final_conv": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_int, c_void_p]),
    "df3d_head_final_packed_bytes": (c_size_t, [c_int]),
    "df3d_head_final_pack": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "df3d_head_final_conv_packed": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_int, c_void_p]),
    "df3d_head_final_conv_backward": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_void_p]),
    "df3d_centerhead_loss_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "df3d_centerhead_loss": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_float,
                                     c_void_p, c_void_p, c_size_t, c_void_p]),
    "df3d_centerhead_loss_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_float,
                                     c_void_p, c_void_p, c_size_t, c_void_p]),
    "df3d_heatmap_proposals_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "df3d_heatmap_proposals": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_uint, c_int, c_void_p, c_int,
                                       c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_size_t, c_void_p]),
    "df3d_transfusion_decode": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p]),
    "df3d_furthest_point_sample_with_dist": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "df3d_group_points_grad": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "df3d_gather_points_grad": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "df3d_rows_linear_packed_bytes": (c_size_t, [c_int, c_int]),
    "df3d_rows_linear": (c_int, [c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int,
                                 c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p]),
    "df3d_voxel_image_sample": (c_int, [c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                        c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p]),
    "df3d_backbone_inputs_ready": (c_int, [c_void_p]),
    "df3d_lt_layer_packed_bytes": (c_longlong, []),
    "df3d_lt_layer_vector_floats": (c_int, []),
    "df3d_lt_layer": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p]),
    "df3d_lt_layer_pe_packed_bytes": (c_longlong, []),
    "df3d_lt_layer_pe_vector_floats": (c_int, []),
    "df3d_lt_layer_gather": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p]),
    "df3d_lt_layer_scatter": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "df3d_boxes_overlap_bev_xyxyr": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "df3d_tf_match_cost": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "df3d_draw_heatmap_gaussian": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                           c_void_p, c_void_p]),
    "df3d_gaussian_focal_loss_workspace_bytes": (c_size_t, [c_longlong]),
    "df3d_gaussian_focal_loss": (c_int, [c_void_p, c_longlong, c_longlong, c_longlong, c_void_p, c_int, c_int, c_int, c_float,
                                         c_float, c_float, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "df3d_tf_query_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                   c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "df3d_sparse_maxpool": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "df3d_sparse_maxpool_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "df3d_dynamic_voxelize": (c_int, [c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "df3d_topk_keys_workspace_bytes": (c_size_t, [c_int, c_longlong, c_int]),
    "df3d_topk_keys": (c_int, [c_void_p, c_int, c_longlong, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "df3d_cross_attention_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "df3d_cross_attention_train": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                           c_float, c_float, c_ulonglong, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "df3d_cross_attention_backward": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                              c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_ulonglong,
                                              c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "df3d_cross_attention": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                     c_float, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "df3d_imgproj_packed_bytes": (c_size_t, [c_int, c_int]),
    "df3d_imgproj_pack": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "df3d_imgproj_split": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "df3d_value_fold_gemm": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_int,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "df3d_value_fold_gemm_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_int,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "df3d_backbone_run": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                  c_size_t, c_void_p, c_void_p, c_void_p]),
    "df3d_backbone_geometry": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p,
                                       c_void_p, c_void_p, c_void_p]),
    "df3d_backbone_convs": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "df3d_backbone_convs_range": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "df3d_backbone_geometry_wait": (c_int, [c_void_p, c_void_p]),
    "df3d_frame_head_image_wait": (c_int, [c_void_p, c_void_p]),
    "df3d_head_worker_create": (c_void_p, [c_int]),
    "df3d_head_worker_destroy": (c_int, [c_void_p]),
    "df3d_frame_head_submit": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "df3d_frame_head_wait": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "df3d_backbone_release": (c_int, [c_void_p]),
    "df3d_actr_prep": (c_int, [c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_void_p, c_void_p, c_void_p]),
    "df3d_relu_dropout_bf16": (c_int, [c_void_p, c_longlong, c_float, c_ulonglong, c_void_p]),
    "df3d_relu_dropout_backward_bf16": (c_int, [c_void_p, c_void_p, c_longlong, c_float, c_void_p, c_void_p]),
    "df3d_relu_dropout": (c_int, [c_void_p, c_longlong, c_float, c_ulonglong, c_void_p]),
    "df3d_relu_dropout_backward": (c_int, [c_void_p, c_void_p, c_longlong, c_float, c_void_p, c_void_p]),
    "df3d_dropout_add_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_ulonglong, c_longlong, c_int,
                                           c_void_p, c_void_p, c_void_p, c_void_p]),
    "df3d_dropout_add_layernorm_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_ulonglong, c_longlong, c_int,
                                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "df3d_add_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_longlong, c_int, c_void_p,
                                   c_void_p]),
    "df3d_add_layernorm_split": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_longlong, c_int, c_void_p,
                                   c_void_p, c_void_p]),
    "df3d_bigate_sum": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_int,
                                c_void_p, c_void_p, c_void_p]),
    "df3d_ms_deform_attn_fused": (c_int, [c_void_p, c_longlong, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_longlong,
                                          c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "df3d_ms_deform_attn_fused_bf16": (c_int, [c_void_p, c_longlong, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_longlong,
                                               c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "df3d_rows_groupnorm": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p,
                                    c_void_p, c_void_p]),
    "df3d_scaled_moments": (c_int, [c_void_p, c_longlong, c_longlong, c_void_p, c_int, c_int, c_int, c_void_p,
                                    c_void_p]),
    "df3d_groupnorm_fold": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_int,
                                    c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "df3d_fusion_writeback": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                      c_void_p, c_void_p]),
}

_lib = None


DF3D_OK, DF3D_EINVAL, DF3D_ENOMEM, DF3D_EHIP = 0, -1, -2, -3      # include/df3d_hip.h


class Df3dError(RuntimeError):
    pass


class StreamArg(c_void_p):
    """What ops._stream() passes as the stream of a launch: lets a recording launch tape (dualfusion/tape.py) tell launches
    from queries and find the argument to rewrite."""


_recorder = None        # dualfusion/tape.py: stands in for the library while a launch tape records


def load():
    """Load (once) and return the ctypes handle.  Raises loudly when the library is absent."""
    global _lib
    if _recorder is not None:
        return _recorder
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Df3dError(
            "libdf3d_hip.so not found at %s: build it with `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback on the product path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().df3d_last_error()
        raise Df3dError("%s failed (rc=%d): %s" % (what or "df3d call", rc, msg.decode() if msg else "?"))


def int3(v):
    """host int[3] argument"""
    arr = (c_int * 3)(*[int(x) for x in v])
    return ctypes.cast(arr, c_void_p), arr  # keep `arr` alive in the caller


def float_arr(v):
    arr = (c_float * len(v))(*[float(x) for x in v])
    return ctypes.cast(arr, c_void_p), arr
