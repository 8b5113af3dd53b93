/*
 * df3d_hip.h -- C ABI of libdf3d_hip.so: the MI355X (gfx950) implementation of the
 * 3D-Dual-Fusion data-parallel hot path (point->voxel scatter + mean VFE, sparse-conv
 * rulebooks, fused sparse convolution, dense scatter, multi-scale deformable attention,
 * point ops of the 3D local self-attention, camera-fusion gather/scatter).
 *
 * Drop-in boundary: the reference binds its native ops through pybind11
 * torch extensions (`voxel_layer`, `sparse_conv_ext`, `MultiScaleDeformableAttention`,
 * `furthest_point_sample_ext`, `ball_query_ext`, `group_points_ext`, `gather_points_ext`).
 * Every entry point below names the reference binding it replaces (file:line under
 * /root/reference: TF/ = TransFusion, CP/ = CenterPoint).  Signatures are plain C:
 * device pointers, sizes and a hipStream_t passed as void*; no torch types.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in _host;
 *   - tensors are dense, row-major, contiguous (the reference asserts the same,
 *     ms_deform_attn_cuda.cu:28-38);
 *   - every function returns 0 on success, a negative DF3D_E* code on error;
 *     df3d_last_error() gives the message (the reference raises RuntimeError);
 *   - kernels are enqueued on `stream` (the reference uses the current torch stream,
 *     torch_utils.h:23-27); nothing synchronises unless stated;
 *   - scratch memory is caller-provided: ask df3d_*_workspace_bytes() first.
 */
#ifndef DF3D_HIP_H_
#define DF3D_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DF3D_OK 0
#define DF3D_EINVAL (-1)   /* bad argument / unsupported shape */
#define DF3D_ENOMEM (-2)   /* workspace too small              */
#define DF3D_EHIP (-3)     /* a HIP runtime call failed        */

#define DF3D_MAX_KVOL 32   /* kernel volume supported by the rulebook/conv kernels (3x3x3 = 27) */

int df3d_version(void);
const char *df3d_last_error(void);
/* number of devices visible / name of device 0 ("gfx950..."), for loud failure on a wrong box */
int df3d_device_count(void);
int df3d_device_arch(char *buf, int buflen);
/* Range check of the fp16 operand splits (round 5; csrc/common.h): the matrix-core kernels of the fp32 configurations
 * take their operands as fp16 hi + lo pairs of the fp32 values scaled by 2^5 (activations: |x| < 2047) or 2^7 (filters:
 * |w| < 511).  A value outside that range (or a NaN / inf) raises a sticky flag on the device instead of passing
 * silently.  Returns 1 if any kernel has raised it since the last reset, 0 if not, < 0 on error; `where` (may be NULL)
 * receives the comma-separated source units that raised it.  SYNCHRONISES the device.  The reference has no counterpart
 * (its fp32 GEMMs have fp32's range, spconv_ops.h:302,338); data beyond the range belongs on the three-part mode. */
int df3d_split_overflow(int reset, char *where, int where_len);
/* round 6 -- the same poll without a host wait: one small launch on `stream` ORs every unit's flag into *out (device word; bit
 * i & 31 = unit i of df3d_split_overflow_units) and clears them when `reset`; the caller reads the word with a device -> host
 * copy it makes anyway.  df3d_split_overflow_units: the units' names, comma separated, -> their number. */
int df3d_split_overflow_collect(uint32_t *out, int reset, void *stream);
int df3d_split_overflow_units(char *buf, int buflen);

/* ------------------------------------------------------------------------------------
 * Voxelisation + fused mean VFE.
 * Replaces voxel_layer.hard_voxelize (TF/mmdet3d/ops/voxel/src/voxelization.h:51-69,
 * bound at voxelization.cpp:7-11, called by TF/mmdet3d/ops/voxel/voxelize.py:46-57) and the
 * numba kernel CP/det3d/ops/point_cloud/point_cloud_ops.py:7-55; mean VFE =
 * CP/det3d/models/readers/voxel_encoder.py:17-24 / TF/.../voxel_encoder.py:27-44.
 *   points [P,C] f32.  Outputs (caller allocates max_voxels rows):
 *   voxels [max_voxels,max_points,C] f32 (may be NULL), coors [max_voxels,3] i32 (z,y,x),
 *   num_points_per_voxel [max_voxels] i32, mean [max_voxels,C] f32 (may be NULL),
 *   voxel_num [1] i32 on the device (first-appearance voxel order, bit-exact with the
 *   reference CPU implementation).  break_at_cap=1: C++ semantics (the scan stops at
 *   the first point that would open voxel #max_voxels, voxelization_cpu.cpp:78);
 *   0: numba semantics (later points still join existing voxels).
 * ---------------------------------------------------------------------------------- */
size_t df3d_hard_voxelize_workspace_bytes(int num_points, int max_points, int max_voxels);
int df3d_hard_voxelize(const float *points, int num_points, int num_features,
                       const float *voxel_size_host, const float *coors_range_host,
                       int max_points, int max_voxels, int break_at_cap,
                       float *voxels, int32_t *coors, int32_t *num_points_per_voxel,
                       float *mean, int32_t *voxel_num,
                       void *workspace, size_t workspace_bytes, void *stream);
/* same, writing coors4 [max_voxels, 4] rows (batch_index, z, y, x): the layout the sparse tensors take
 * (the reference prepends the batch column on the host, CP/det3d/torchie/parallel/collate.py) */
int df3d_hard_voxelize_batched(const float *points, int num_points, int num_features,
                               const float *voxel_size_host, const float *coors_range_host,
                               int max_points, int max_voxels, int break_at_cap, int batch_index,
                               float *voxels, int32_t *coors4, int32_t *num_points_per_voxel,
                               float *mean, int32_t *voxel_num,
                               void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * Rulebook.  Replaces sparse_conv_ext.get_indice_pairs_3d
 * (TF/mmdet3d/ops/spconv/src/all.cc:24, spconv_ops.h:27-141, kernels indice.cu.h:24-203).
 *
 * The reference fills a dense int32 grid of B*Z*Y*X cells per call (340 MB at 0.075 m).
 * Here an occupancy *directory* (1 bit per cell + a 32-bit popcount prefix per 64 cells,
 * i.e. 12 bytes per 64 cells) answers "row index of voxel (b,z,y,x)" with two reads, and
 * doubles as the sorted output list of a strided convolution.
 *
 * df3d_grid_build: directory over `indices` [n,4] (b,z,y,x) i32 living in a
 *   [batch, shape] grid.  `perm` ([n] i32, may be NULL) receives rank->row; pass NULL when
 *   the rows are already sorted by flat index (outputs of df3d_conv_out_indices are).
 * df3d_subm_neighbors: nbr [K,n] i32, nbr[k][o] = input row feeding output o through kernel
 *   offset k (row-major over kz,ky,kx as geometry.h:69-73), -1 if absent.  stride 1,
 *   pad = ksize/2 (spconv_ops.h:76-79).
 * df3d_conv_out_indices: active outputs of a strided sparse conv, sorted by flat index
 *   (what the reference's GPU path yields through torch::_unique, spconv_ops.h:119-137);
 *   builds the OUTPUT grid directory in out_grid; writes out_indices [cap,4] and the count
 *   (device i32).  cap >= min(n*K, batch*vol_out) rows is always enough.
 * df3d_conv_neighbors: nbr [K,n_out] for the strided conv (lookup in the INPUT directory).
 * df3d_nbr_to_pairs: reference-format rulebook from a nbr table: indice_pairs [K,2,n_in]
 *   (-1 padded), indice_num [K]; pairs of one offset are emitted in output-row order.
 * ---------------------------------------------------------------------------------- */
size_t df3d_grid_bytes(int batch, const int *shape_host);
int df3d_grid_build(const int32_t *indices, int n, int batch, const int *shape_host,
                    void *grid, size_t grid_bytes, int32_t *perm, void *stream);
int df3d_subm_neighbors(const void *grid, const int32_t *perm, const int32_t *indices, int n,
                        int batch, const int *shape_host, const int *ksize_host,
                        const int *dilation_host, int32_t *nbr, void *stream);
int df3d_conv_out_indices(const int32_t *indices, int n, int batch, const int *in_shape_host,
                          const int *out_shape_host, const int *ksize_host, const int *stride_host,
                          const int *padding_host, const int *dilation_host,
                          void *out_grid, size_t out_grid_bytes,
                          int32_t *out_indices, int out_cap, int32_t *num_out, void *stream);
int df3d_conv_neighbors(const void *in_grid, const int32_t *in_perm, const int32_t *out_indices,
                        int n_out, int batch, const int *in_shape_host, const int *ksize_host,
                        const int *stride_host, const int *padding_host, const int *dilation_host,
                        int32_t *nbr, void *stream);
/* Transposed convolution (TF/mmdet3d/ops/spconv/conv.py:262-455 SparseConvTranspose2d/3d, get_indice_pairs(transpose=True):
 * ops.py:72-94 -> spconv_ops.h:27-141 -> geometry.h:88-142,194-245): input voxel `in` writes output cell
 * in*stride - pad + c*dil through kernel index c; out_shape = (in-1)*stride - 2*pad + k + output_padding.  Same
 * contracts as df3d_conv_out_indices / df3d_conv_neighbors (outputs sorted by flat index, nbr [K, n_out]). */
int df3d_conv_transpose_out_indices(const int32_t *indices, int n, int batch, const int *in_shape, const int *out_shape,
                                    const int *ksize, const int *stride, const int *padding, const int *dilation,
                                    void *out_grid, size_t out_grid_bytes, int32_t *out_indices, int out_cap,
                                    int32_t *num_out, void *stream);
int df3d_conv_transpose_neighbors(const void *in_grid, const int32_t *in_perm, const int32_t *out_indices, int n_out,
                                  int batch, const int *in_shape, const int *ksize, const int *stride,
                                  const int *padding, const int *dilation, int32_t *nbr, void *stream);
size_t df3d_nbr_to_pairs_workspace_bytes(int kvol, int n_out);
int df3d_nbr_to_pairs(const int32_t *nbr, int kvol, int n_out, int n_in,
                      int32_t *indice_pairs, int32_t *indice_num,
                      void *workspace, size_t workspace_bytes, void *stream);
/* inverse helper for callers that hold a reference-format rulebook: nbr [K,n_out] from pairs */
int df3d_pairs_to_nbr(const int32_t *indice_pairs, const int32_t *indice_num_host, int kvol,
                      int n_in, int n_out, int32_t *nbr, void *stream);

/* ------------------------------------------------------------------------------------
 * Sparse convolution, fused.  Replaces sparse_conv_ext.indice_conv_fp32 and
 * fused_indice_conv_fp32 (TF/.../src/all.cc:32,38; indiceConv<T> spconv_ops.h:260-361:
 * 27 x {gather kernel, GEMM, scatter-add kernel} + a D2H sync per conv;
 * fusedIndiceConvBatchNorm fused_spconv_ops.h:28-132).
 * One output-stationary implicit-GEMM kernel per conv:
 *   out[o,:] = act( (sum_k features[nbr[k][o],:] @ filters[k] + bias) * scale + shift + residual )
 * features [n_in,cin] f32, filters [K,cin,cout] f32 (= spconv weight [kD,kH,kW,Cin,Cout],
 * conv.py:98-99), bias/scale/shift [cout] or NULL, residual [n_out,cout] or NULL, relu 0/1.
 * ---------------------------------------------------------------------------------- */
int df3d_sparse_conv_fused(const float *features, int n_in, int cin,
                           const float *filters, int kvol, int cout,
                           const int32_t *nbr, int n_out,
                           const float *bias, const float *scale, const float *shift,
                           const float *residual, int relu, float *out, void *stream);

/* Pair-balanced work split for the compute-bound layers.  The occupancy of a LiDAR sweep varies ~1.4x
 * between equal-row tiles and the slowest workgroup sets the kernel time, so the pair-compacted kernel
 * accepts row ranges that hold equal numbers of rulebook pairs.  They depend only on the neighbour table
 * and are cached with it by the host side (one prefix scan per rulebook, not per conv).
 *   df3d_conv_tile_count: recommended number of ranges for a (cin, cout) layer, 0 = not applicable.
 *   df3d_conv_tiles: tile_rows [ntiles+1] i32, tile_rows[i]..tile_rows[i+1] = rows of range i.
 *   df3d_sparse_conv_fused_tiled: as df3d_sparse_conv_fused, with the ranges (NULL/0 = equal rows). */
int df3d_conv_tile_count(int n_out, int cin, int cout, int kvol);
size_t df3d_conv_tiles_workspace_bytes(int n_out);
int df3d_conv_tiles(const int32_t *nbr, int kvol, int n_out, int ntiles, int32_t *tile_rows,
                    void *workspace, size_t workspace_bytes, void *stream);
int df3d_sparse_conv_fused_tiled(const float *features, int n_in, int cin,
                                 const float *filters, int kvol, int cout,
                                 const int32_t *nbr, int n_out,
                                 const float *bias, const float *scale, const float *shift,
                                 const float *residual, int relu, float *out,
                                 const int32_t *tile_rows, int ntiles, void *stream);

/* round 4 -- ROW LISTS of a neighbour table for the small-channel layers (C_in <= 16, C_out 16 | 32: conv_input and the conv1
 * stage of the backbones, whose tables are ~90 % empty): the present entries of every output row, offsets ascending, packed
 * (offset << 26 | input row), rows back to back.  Same role as the reference's compacted indice_pairs[K][2][N] + indice_num[K]
 * (spconv_ops.h:27-141), transposed to output-row order so that the accumulation order stays the table's.
 *   df3d_nbr_row_lists_bytes(kvol, n_out)   bytes of the blob (offsets, entries at full capacity, scratch)
 *   df3d_nbr_row_lists                      builds it (count, scan, fill: three launches, no host round trip); n_in < 2^26
 *   df3d_sparse_conv_fused_lists            df3d_sparse_conv_fused that takes the blob (NULL, or a shape the small-channel
 *                                           kernel does not serve: exactly df3d_sparse_conv_fused).  Bit-identical results.
 *                                           out_split (optional, out_channels % 8 == 0): also the df3d_split_rows form of the
 *                                           result, from the same launch (what a following matrix-core layer reads). */
size_t df3d_nbr_row_lists_bytes(int kvol, int n_out);
int df3d_nbr_row_lists(const int32_t *nbr, int kvol, int n_out, int n_in, void *lists, size_t lists_bytes, void *stream);
int df3d_sparse_conv_fused_lists(const float *features, int n_in, int cin, const float *filters, int kvol, int cout,
                                 const int32_t *nbr, const void *lists, int n_out, const float *bias, const float *scale,
                                 const float *shift, const float *residual, int relu, float *out, void *out_split,
                                 void *stream);

/* Per-launch timing of the sparse-conv kernels with HIP events recorded on the launching stream, adjacent
 * to the launch (measurement aid for bench.py's roofline leg; off by default, not thread-safe).
 * begin: reset + enable.  end: disable, returns the number of records.  get: shape4 = (cin, cout, kvol,
 * n_out) of record i and its duration in ms (synchronises on the record's stop event). */
int df3d_timing_begin(void);
int df3d_timing_end(void);
int df3d_timing_get(int i, int *shape4_host, float *ms_host);
/* df3d_timing_count_pairs(1): metadata mode for a pass OUTSIDE any timed region -- every recorded launch also
 * counts its valid (output row, offset) pairs R (one extra kernel + a D2H read per launch), which is what the
 * algorithmic bytes / flops of SURVEY.md section 8(d) are stated in.  df3d_timing_get2 returns R (-1 when it was
 * not counted) and whether the split-precision kernel served the launch. */
int df3d_timing_count_pairs(int on);
/* Only launches of this (cin, cout, kernel volume) are timed (0 = any): event pairs around EVERY conv launch cost the
 * timed region ~6 %; the roofline needs the dominant kernel only. */
int df3d_timing_filter(int cin, int cout, int kvol);
/* events around every `every`-th launch that passes the filter only (1 = all; reset by the call) */
int df3d_timing_sample(int every);
int df3d_timing_get2(int i, int *shape4, float *ms, long long *pairs, int *split);

/* SparseConvTensor.dense() (TF/mmdet3d/ops/spconv/structure.py:5-18,55-64): zero-fill +
 * scatter + permute fused; out [B, C, D, H, W] f32 (the backbones view it as [B, C*D, H, W]). */
/* `groups` convolutions over ONE neighbour table in one launch of the exact-fp32 MFMA kernel, operands read and written
 * in place as COLUMN SLICES of wider rows: group g convolves features[:, g * group_in : g * group_in + cin] (row stride
 * ld_in floats) with filters[g] ([groups, kvol, cin, cout]) and bias / scale / shift[g * cout : (g + 1) * cout] into
 * out[:, g * group_out : g * group_out + cout] (row stride ld_out floats).  group_in = 0: every group reads the same
 * columns.  This is the detection heads' exact-fp32 mode (CP/det3d/models/bbox_heads/center_head.py:66-110: the first
 * convs of all branches read the 64 shared channels, each final conv reads its branch's 64 channels); epilogue as
 * df3d_sparse_conv_fused without the residual.  cin in {16 .. 512} (power of two), cout in {16, 32, 64, 128, 256}. */
int df3d_sparse_conv_grouped(const float *features, int n_in, int cin, int ld_in, int group_in, const float *filters, int kvol,
                             int cout, int groups, const int32_t *nbr, int n_out, const float *bias, const float *scale,
                             const float *shift, int relu, float *out, int ld_out, int group_out, void *stream);
int df3d_sparse_to_dense(const float *features, const int32_t *indices, int n, int channels,
                         int batch, const int *shape_host, float *out, void *stream);

/* Dense BEV neck on the sparse-convolution kernels (CP/det3d/models/necks/rpn.py,
 * dense 3x3 / strided / transposed conv2d + BatchNorm + ReLU on [B,256,180,180]).
 *   df3d_sparse_to_dense_rows: SparseConvTensor.dense().view(N, C*D, H, W) emitted channels-last as rows
 *       [(b,y,x)][c*D + d] fp32 -- the layout the conv kernels gather from (no NCHW volume, no permute);
 *   df3d_conv2d_neighbors: neighbour table nbr[kh*kw][B*Ho*Wo] of a dense Conv2d (cross-correlation, tap k =
 *       ky*kw + kx) or, with transposed != 0, of a ConvTranspose2d with kernel == stride (tap k = (oy%s)*s + ox%s).
 * A dense layer is then df3d_sparse_conv_split / df3d_sparse_conv_fused on those rows with filters
 * [kh*kw][Cin][Cout] (Conv2d weight.permute(2,3,1,0); ConvTranspose2d weight.permute(2,3,0,1)). */
int df3d_sparse_to_dense_rows(const float *features, const int32_t *indices, int n, int channels, int batch,
                              const int *shape_host, float *out_rows, void *stream);
/* the same rows as split rows (bf16 hi | lo per 8 columns, the operand format of df3d_sparse_conv_split): what the BEV neck's
 * first convolution reads -- no fp32 copy, no df3d_split_rows pass (channels * shape[0] must be a multiple of 8) */
int df3d_sparse_to_dense_rows_split(const float *features, const int32_t *indices, int n, int channels, int batch,
                                    const int *shape_host, void *out_split, void *stream);
int df3d_conv2d_neighbors(int batch, int H, int W, int kh, int kw, int stride, int pad, int transposed,
                          int32_t *nbr, void *stream);

/* bf16 variant of the fused sparse convolution (BASELINE configs[2]/[3]: "bf16, fp32 accumulate"): feature rows are
 * [N][C] bf16 (C % 8 == 0), the packed filter bank holds bf16 weights, one MFMA product per operand pair, fp32
 * accumulation and fp32 epilogue (bias, folded BN, residual -- read as bf16 rows --, ReLU), output as bf16 rows
 * (out_bf16) and / or fp32 rows (out); either may be NULL.  Half the gather / store bytes and a third of the matrix
 * instructions of df3d_sparse_conv_split.  Served (cin, cout): 32->32/64, 64->64/128, 128->128/256, 256->128/256.
 *   df3d_rows_to_bf16 / df3d_rows_from_bf16: fp32 rows <-> bf16 rows (round to nearest even). */
int df3d_rows_to_bf16(const float *features, long long n, int c, void *rows_bf16, void *stream);
int df3d_rows_from_bf16(const void *rows_bf16, long long n, int c, float *features, void *stream);
size_t df3d_conv_packed_weight_bytes_bf16(int kvol, int cin, int cout);
int df3d_conv_pack_weights_bf16(const float *filters, int kvol, int cin, int cout, void *packed, void *stream);
int df3d_sparse_conv_bf16(const void *features_bf16, int n_in, int cin, const void *packed_filters, int kvol, int cout,
                          const int32_t *nbr, int n_out, const float *bias, const float *scale, const float *shift,
                          const void *residual_bf16, int relu, float *out, void *out_bf16, void *stream);

/* Sparse convolution backward.  Replace sparse_conv_ext.indice_conv_backward_fp32
 * (TF/mmdet3d/ops/spconv/src/all.cc:21-51, include/spconv/spconv_ops.h:363-456):
 *   input gradient  = df3d_sparse_conv_fused(grad_out, filters^T per offset [K][Cout][Cin], inverse table, n_in) -- the
 *       forward kernel, output-stationary in the input rows (no scatter-add).  df3d_invert_neighbors builds the
 *       inverse table inv[k][i] = o of nbr[k][o] = i ([K][n_in], -1 = none); a submanifold convolution needs none:
 *       inv[k] = nbr[K-1-k].
 *   filter gradient = df3d_sparse_conv_grad_filters: grad_filters[k][ci][co] = sum_o features[nbr[k][o]][ci] *
 *       grad_out[o][co] ([K][Cin][Cout], zero-filled here; fp32 MFMA partial tiles are added with fp32 atomics, so
 *       the summation order varies between runs like the reference's cuBLAS split-K).  Channel counts that are multiples
 *       of 4 run on the LDS-staged pair-compacted kernels (any Cout) -- from 64 channels on both sides on the 16-bit matrix
 *       cores with both operands split into three bf16 parts (six products, fp32 accumulate: fp32-grade, ~1e-6 of scale;
 *       round 5), below that with exact fp32 products; anything else on the direct kernel (Cout <= 128).
 *   df3d_rows_grad_weights: the same contraction without a table -- grad_weights[ci][co] = sum_r x[r][ci] * grad_out[r][co]
 *       ([Cin][Cout], zero-filled here): the weight gradient of a linear layer over rows (torch.nn.Linear.weight.grad =
 *       this with the roles of x and grad_out exchanged; the adapter's query / value / feed-forward projections,
 *       CP/det3d/models/fusion/actr_transformer.py:388-424), where the library GEMM runs a [Cin x rows] . [rows x Cout]
 *       product with 32 k - 240 k rows at a fifth of its rate. */
int df3d_invert_neighbors(const int32_t *nbr, int kvol, int n_out, int n_in, int32_t *inv, void *stream);
int df3d_sparse_conv_grad_filters(const float *features, int n_in, int cin, const float *grad_out, int n_out, int cout,
                                  const int32_t *nbr, int kvol, float *grad_filters, void *stream);
int df3d_rows_grad_weights(const float *x, const float *grad_out, long long n, int cin, int cout, float *grad_weights,
                           void *stream);
/* Two-part forms of the two gradients above (fp16 pairs, three products instead of six): an ACTIVATION operand carries the fixed
 * scale of every split kernel, a GRADIENT operand the power-of-two block scale of its tensor -- `grad_scale` / `x_scale` /
 * `g_scale` point at it on the device (scale[0] of df3d_split_rows_scaled or df3d_rows_pow2_scale; NULL = an activation operand).
 * df3d_rows_pow2_scale: scale [df3d_pow2_scale_floats()] <- (2^k with max |x| * 2^k in [512, 1024), max |x|, workspace). */
int df3d_sparse_conv_grad_filters_scaled(const float *features, int n_in, int cin, const float *grad_out, int n_out, int cout,
                                         const int32_t *nbr, int kvol, const float *grad_scale, float *grad_filters, void *stream);
int df3d_rows_grad_weights_scaled(const float *x, const float *grad_out, long long n, int cin, int cout, const float *x_scale,
                                  const float *g_scale, float *grad_weights, void *stream);
/* round 6 -- the filter gradient of bf16 mixed-precision training (BASELINE configs[2] / [3]): both operands rounded to one bf16
 * part where they are staged, one matrix-core product per pair block, fp32 accumulate (what the reference's fp16-AMP runs give
 * `indice_conv_backward`'s filtersGrad, TF/mmdet3d/ops/spconv/include/spconv/spconv_ops.h:363-456); layers under 64 channels
 * take df3d_sparse_conv_grad_filters. */
int df3d_sparse_conv_grad_filters_bf16(const float *features, int n_in, int cin, const float *grad_out, int n_out, int cout,
                                       const int32_t *nbr, int kvol, float *grad_filters, void *stream);
int df3d_rows_pow2_scale(const float *x, long long n_elems, float *scale, void *stream);
int df3d_pow2_scale_floats(void);
/* out[n][c] = sum_s x[n][c][s] * g[n][s] over channel-first maps x [nmaps][channels][S], g [nmaps][S]: the weight gradient of a
 * one-output 1 x 1 convolution (the image gate's `reduced_dim3`, CP/det3d/models/fusion/point_to_image_projection.py:34-61) is the
 * sum of out over the maps. */
int df3d_chanfirst_dot(const float *x, const float *g, int nmaps, int channels, long long S, float *out, void *stream);

/* BatchNorm with batch statistics over channels-last rows [n][c] (training rows).  Replace
 * torch.nn.BatchNorm1d / BatchNorm2d in train() mode behind the convolutions of the sparse backbone, the BEV neck and the head
 * (CP/det3d/models/backbones/scn.py:51-118, necks/rpn.py:22-163, bbox_heads/center_head.py:66-110), optionally with the ReLU
 * that follows them.  forward: y = relu?((x - mean) * rstd * weight + bias) with the batch mean / biased variance of the
 * columns; running_mean / running_var (may be NULL) updated in place with `momentum` and the unbiased variance; `sums`
 * (df3d_bn_rows_scratch_doubles(c) doubles) is scratch, `saved` [4][c] floats (mean, rstd, scale, shift) is what backward needs.  backward: dx, dweight
 * [c], dbias [c] (either may be NULL) from x, dy and `saved`; with relu != 0 the mask is recomputed from x.
 * Channel counts: multiples of 4 that divide 256 or are multiples of 256 (df3d_bn_rows_supported). */
int df3d_bn_rows_supported(int c);
int df3d_bn_rows_scratch_doubles(int c);
int df3d_bn_rows_forward(const float *x, long long n, int c, const float *weight, const float *bias, float eps,
                         float momentum, int relu, float *running_mean, float *running_var, double *sums, float *saved,
                         float *y, void *stream);
int df3d_bn_rows_backward(const float *x, const float *dy, long long n, int c, const float *saved, int relu, double *sums,
                          float *dx, float *dweight, float *dbias, void *stream);

/* Grouped / multi-head convolution over pixel (or voxel) rows on the split-precision kernel -- the detection head's
 * stacks (CP/det3d/models/bbox_heads/center_head.py:66-110: per task and per head Conv2d 3x3 64 -> 64 + BN + ReLU
 * + Conv2d 3x3 64 -> classes) as ONE launch per depth instead of one cuDNN call per head.
 *   Group g reads the input columns [g * in_group_stride, + cin) of the split rows `in_split` ([n_in][in_channels],
 *   df3d_split_rows layout; in_group_stride = 0: every group reads the same columns), uses the g-th of `groups`
 *   consecutive packed filters (each df3d_conv_pack_weights(kvol, cin, cout)) and writes the output columns
 *   [g * cout, + cout) of `out` ([n_out][out_channels] f32, and of `out_split` when given).  With out_cols
 *   ([groups][2] i32 on the device: first output column, number of valid columns <= 32; needs cout = 32) only the
 *   valid columns are stored, anywhere in the row -- the heads' 1..3-channel outputs next to each other.
 *   `out` may be NULL when only the split rows are wanted (an intermediate layer).
 *   bias / scale / shift are indexed g * cout + c.  Served (cin, cout): those of df3d_conv_packed_weight_bytes
 *   != 0 on the output-stationary kernel (64->64, 64->32, 512->64, 128->128, ...). */
int df3d_conv_rows_split(const void *in_split, int n_in, int in_channels, int cin, int in_group_stride,
                         const void *packed_filters, int kvol, int cout, int groups, const int32_t *nbr, int n_out,
                         const float *bias, const float *scale, const float *shift, int relu, float *out,
                         int out_channels, const int32_t *out_cols, void *out_split, void *stream);

/* ------------------------------------------------------------------------------------
 * "split3" precision (round 4): the same convolutions with operands in THREE bf16 parts -- hi + mid + lo is the fp32 value
 * exactly -- and six matrix-core products per operand pair (every product of parts i, j with i + j <= 2; what is dropped is
 * <= 2^-24 of a product), fp32 accumulate.  Results are fp32-grade (the reference's own precision: spconv_ops.h:260-361
 * runs torch::mm in fp32) at 1/6 of the bf16 matrix rate = 2.6x the fp32 matrix rate of gfx950.
 *   rows     [n][C/8][hi 8 x bf16 | mid | lo]   48 B per 8 channels (df3d_split_rows3, or a convolution's out_split3)
 *   filters  df3d_conv_pack_weights3 (groups filter banks [groups][kvol][cin][cout] back to back; groups = 1 for one bank)
 * df3d_conv_rows_split3 = df3d_conv_rows_split on these formats, plus the optional fp32 `residual` rows of
 * df3d_sparse_conv_split (one group, dense output rows): it serves the sparse backbone, the BEV neck and the head alike. */
size_t df3d_conv_packed_weight_bytes3(int kvol, int cin, int cout);
int df3d_conv_pack_weights3(const float *filters, int groups, int kvol, int cin, int cout, void *packed, void *stream);
int df3d_split_rows3(const float *features, long long n, int c, void *split3, void *stream);
int df3d_conv_rows_split3(const void *in_split3, int n_in, int in_channels, int cin, int in_group_stride, const void *packed3,
                          int kvol, int cout, int groups, const int32_t *nbr, int n_out, const float *bias, const float *scale,
                          const float *shift, const float *residual, int relu, float *out, int out_channels,
                          const int32_t *out_cols, void *out_split3, void *stream);

/* ------------------------------------------------------------------------------------
 * Detection tail: rotated BEV overlap / IoU / NMS.  Replace the pybind module
 * `iou3d_nms_cuda` (CP/det3d/ops/iou3d_nms/src/iou3d_nms_api.cpp:11-17): boxes_overlap_bev_gpu / boxes_iou_bev_gpu
 * (iou3d_nms.cpp:38-85) and nms_gpu / nms_normal_gpu (iou3d_nms.cpp:88-188), plus the numba circle NMS
 * (CP/det3d/core/utils/circle_nms_jit.py:4-27).  Boxes are rows [x, y, z, dx, dy, dz, heading] f32.
 *   df3d_boxes_bev_pairwise: out[na][nb] = overlap area (mode 0) or rotated IoU (mode 1).
 *   df3d_nms_bev: greedy NMS of `lists` independent lists boxes[lists][cap][7], each already sorted by descending
 *       score, counts[lists] valid boxes (device i32; NULL = cap each).  keep[lists][cap] receives the kept indices
 *       in ascending order, num_keep[lists] their number, clipped to max_keep when max_keep > 0 (the reference's
 *       `selected[:post_max_size]`).  thresh = IoU threshold, or the squared centre distance for DF3D_NMS_CIRCLE.
 *       The bit matrix AND its greedy reduction stay on the device (the reference copies the matrix to the host,
 *       iou3d_nms.cpp:108-133).  cap <= 4096.  workspace: df3d_nms_bev_workspace_bytes(lists, cap). */
#define DF3D_NMS_NORMAL 0
#define DF3D_NMS_ROTATED 1
#define DF3D_NMS_CIRCLE 2
int df3d_boxes_bev_pairwise(const float *boxes_a, int na, const float *boxes_b, int nb, int mode, float *out,
                            void *stream);
size_t df3d_nms_bev_workspace_bytes(int lists, int cap);
int df3d_nms_bev(const float *boxes, const int32_t *counts, int lists, int cap, float thresh, int mode, int max_keep,
                 int32_t *keep, int32_t *num_keep, void *workspace, size_t workspace_bytes, void *stream);

/* CenterHead.predict + post_processing (CP/det3d/models/bbox_heads/center_head.py:302-501) for all tasks and samples in
 * one call, no host round trip: score = max_c sigmoid(hm), box = [(x + reg_x) * out_size_factor * voxel_size_x +
 * pc_range_x, ..y.., height, exp(dim) x3, vel x2, atan2(rot_0, rot_1)], mask = score > score_threshold and the centre
 * inside post_center_range (inclusive); candidates sorted by descending score (ties: lower pixel index), the first
 * pre_max go through rotate_nms_pcdet (box_torch_ops.py:248-279; nms_mode as for df3d_nms_bev, the circle variant
 * takes nms_threshold = min_radius), the first post_max kept boxes are returned.
 *   Head maps are channels-last rows [(b, y, x)][channels] with row strides ld_* in floats (the layout the row
 *   kernels of the neck / head produce; NCHW maps need the reference's own permute(0, 2, 3, 1) first).
 *   A segment is (task t, sample b) -> index t * batch + b.  Outputs: out_boxes [segments][post_max][9 or 7 without
 *   vel], out_scores / out_labels (label + label_base; -1 = empty slot) [segments][post_max], out_counts [segments].
 *   Limits: <= DF3D_MAX_HEAD_TASKS tasks, tasks * batch <= 255, H*W < 2^24, pre_max <= 4096. */
#define DF3D_MAX_HEAD_TASKS 8
typedef struct df3d_head_task {
  const float *hm, *reg, *height, *dim, *rot, *vel; /* vel NULL = 7-value boxes */
  int ld_hm, ld_reg, ld_height, ld_dim, ld_rot, ld_vel;
  int num_classes, label_base;
} df3d_head_task;
typedef struct df3d_head_decode_cfg {
  int batch, H, W;
  float out_size_factor, voxel_size[2], pc_range[2];
  int has_post_center_range;
  float post_center_range[6];
  float score_threshold;
  int nms_mode;
  float nms_threshold;
  int pre_max, post_max;
} df3d_head_decode_cfg;
size_t df3d_centerhead_predict_workspace_bytes(int ntasks, const df3d_head_decode_cfg *cfg);
int df3d_centerhead_predict(const df3d_head_task *tasks, int ntasks, const df3d_head_decode_cfg *cfg, float *out_boxes,
                            float *out_scores, int32_t *out_labels, int32_t *out_counts, void *workspace,
                            size_t workspace_bytes, void *stream);

/* df3d_head_final_conv: the LAST convolution of every (task, head) branch of a CenterPoint-style head in one launch
 * (SepHead, CP/det3d/models/bbox_heads/center_head.py:66-110: Conv2d 64 -> k, 3x3, padding 1, bias, k <= 4 output maps
 * per branch).  in_split: split rows [batch*H*W, in_channels] (32 B per 8 channels: hi | lo) whose columns
 * g*64 .. g*64+63 are branch g's activations; weights [groups][9][64][4] fp32 (tap-major, output maps padded to 4),
 * bias [groups][4]; out_cols [groups][2] = (first output column, valid maps) in out [batch*H*W, out_channels] fp32. */
int df3d_head_final_conv(const void *in_split, int in_channels, int batch, int H, int W, int groups, const float *weights,
                         const float *bias, const int32_t *out_cols, float *out, int out_channels, void *stream);
/* The same with the filters packed once as matrix-core operands (round 3: the kernel multiplies every halo pixel's 64
 * channels with all 9 taps' filters on the matrix cores -- split precision, 3 products -- and sums the shifted partial
 * maps in LDS).  df3d_head_final_conv splits the fp32 filters inside the kernel (48 scattered loads per lane);
 * `packed` = df3d_head_final_packed_bytes(groups) bytes written by df3d_head_final_pack from weights [groups][9][64][4]. */
size_t df3d_head_final_packed_bytes(int groups);
int df3d_head_final_pack(const float *weights, int groups, void *packed, void *stream);
int df3d_head_final_conv_packed(const void *in_split, int in_channels, int batch, int H, int W, int groups,
                                const void *packed, const float *bias, const int32_t *out_cols, float *out,
                                int out_channels, void *stream);
/* Backward of df3d_head_final_conv for training: `acts` [B*H*W][act_channels] are the fp32
 * activations the forward convolved (branch g at columns g*64 ..), `grad_out` [B*H*W][out_channels] the gradient of the
 * packed maps; grad_acts (same shape as acts; columns beyond groups*64 untouched) and / or grad_weights
 * [groups][9][64][4] (zero-filled here) may be NULL. */
int df3d_head_final_conv_backward(const float *acts, int act_channels, const float *grad_out, int out_channels, int batch,
                                  int H, int W, int groups, const float *weights, const int32_t *out_cols,
                                  float *grad_acts, float *grad_weights, void *stream);

/* df3d_centerhead_loss replaces CenterHead.loss for a no-grad evaluation of the detection losses
 * (CP/det3d/models/bbox_heads/center_head.py:250-298 over FastFocalLoss / RegLoss,
 * CP/det3d/models/losses/centernet_loss.py:6-58): per task the CornerNet focal loss of the clamped sigmoid heat map
 * and the code-weighted L1 loss of the box regressions gathered at the object slots -- the scalars the data-parallel
 * step reduces over the ranks (`reduce_dict`, CP/det3d/torchie/trainer/utils.py:157-183).  Two launches for all tasks
 * and samples, no host round trip (the reference: ~25 launches and two `.cpu()` per task).
 *   tasks: head maps as channels-last pixel rows (as for df3d_centerhead_predict; label_base unused);
 *   targets (per task, what the reference's assigner puts into `example`): hm [B, classes, H, W] f32,
 *   ind [B, max_objs] i64 (flat pixel), mask [B, max_objs] u8, cat [B, max_objs] i64, box = anno_box
 *   [B, max_objs, box_dim] f32 (10 codes: reg 2, height 1, dim 3, vel 2, rot 2; heads without vel compare columns
 *   [0..5, -2, -1], center_head.py:229); code_weights: HOST array [ncodes] (10 with vel, 8 without).
 *   out [ntasks][DF3D_LOSS_FIELDS] f32: loss, hm_loss, loc_loss, num_positive, loc_loss_elem[DF3D_LOSS_MAX_CODES]. */
#define DF3D_LOSS_MAX_CODES 10
#define DF3D_LOSS_FIELDS (4 + DF3D_LOSS_MAX_CODES)
typedef struct df3d_head_targets {
  const float *hm;
  const long long *ind;
  const unsigned char *mask;
  const long long *cat;
  const float *box;
} df3d_head_targets;
size_t df3d_centerhead_loss_workspace_bytes(int ntasks, int batch, int H, int W);
int df3d_centerhead_loss(const df3d_head_task *tasks, const df3d_head_targets *targets, int ntasks, int batch, int H, int W,
                         int max_objs, int box_dim, const float *code_weights, int ncodes, float weight, float *out,
                         void *workspace, size_t workspace_bytes, void *stream);
/* The same two launches for TRAINING: additionally d(sum over tasks of `loss`) / d(every head
 * map), written into `grad_tasks` (the layout of `tasks`; the heat-map gradients are stored, the box-code gradients added:
 * zero those maps first).  Replaces the backward of ~45 torch ops per task (centernet_loss.py:6-58 through autograd). */
int df3d_centerhead_loss_grad(const df3d_head_task *tasks, const df3d_head_task *grad_tasks, const df3d_head_targets *targets,
                              int ntasks, int batch, int H, int W, int max_objs, int box_dim, const float *code_weights,
                              int ncodes, float weight, float *out, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * TransFusionHead (LiDAR-only branch), the index and decode steps.
 *
 * df3d_heatmap_proposals replaces transfusion_head.py:843-878 (TF/mmdet3d/models/dense_heads/): sigmoid of the
 * dense heat map, 3x3 (nms_kernel_size) local-maximum suppression on interior pixels, classes of `exempt_classes`
 * (bit c) keep every pixel (:856-861, nuScenes 8 and 9 / Waymo 1 and 2), top num_proposals of the [C*H*W] scores of
 * each sample (descending; equal scores by ascending flat index, which the reference's unstable argsort leaves open),
 * query features  feat[pixel] + class_weight[:, class] + class_bias  (class_encoding Conv1d on the one-hot, :872-875)
 * and query positions (x + 0.5, y + 0.5) (:758-765,878).
 *   heat_rows [B*H*W, ld_heat] f32 logits (channels-last pixel rows), feat_rows [B*H*W, ld_feat] f32,
 *   class_weight [channels, C] f32 (Conv1d weight [channels, C, 1]), class_bias [channels];
 *   out: top_class, top_pixel [B, K] i32; query_score [B, C, K] f32 (= 'query_heatmap_score', :1013);
 *   query_pos [B, K, 2] f32; query_feat [B, K, channels] f32 (either may be NULL).
 *
 * df3d_transfusion_decode replaces get_bboxes with nms_type=None (:1285-1312) over TransFusionBBoxCoder.decode
 * (core/bbox/coders/transfusion_bbox_coder.py:41-128, filter=True): score = sigmoid(heatmap)[label] *
 * query_score[label], boxes [x, y, z, dx, dy, dz, rot, (vx, vy)], post_center_range mask and score threshold
 * (applied only when non-zero, like `if self.score_threshold:`), order-preserving compaction per sample.
 *   heads: rows [B*K, ld_*] of the LAST decoder layer; cfg: out_size_factor, voxel_size, pc_range,
 *   has_post_center_range / post_center_range, score_threshold (batch, H, W, nms_*, pre/post_max unused);
 *   out_boxes [B, K, 7|9], out_scores / out_labels [B, K], out_counts [B].
 */
typedef struct df3d_query_heads {
  const float *heatmap, *center, *height, *dim, *rot, *vel; /* vel may be NULL */
  int ld_heatmap, ld_center, ld_height, ld_dim, ld_rot, ld_vel;
} df3d_query_heads;
size_t df3d_heatmap_proposals_workspace_bytes(int batch, int num_classes, int H, int W);
int df3d_heatmap_proposals(const float *heat_rows, int ld_heat, int batch, int num_classes, int H, int W,
                           int nms_kernel_size, unsigned exempt_classes, int num_proposals, const float *feat_rows,
                           int ld_feat, int channels, const float *class_weight, const float *class_bias,
                           int32_t *top_class, int32_t *top_pixel, float *query_score, float *query_pos,
                           float *query_feat, void *workspace, size_t workspace_bytes, void *stream);
int df3d_transfusion_decode(const df3d_query_heads *heads, const float *query_score, const int32_t *query_label,
                            int batch, int num_proposals, int num_classes, const df3d_head_decode_cfg *cfg,
                            float *out_boxes, float *out_scores, int32_t *out_labels, int32_t *out_counts, void *stream);

/* ------------------------------------------------------------------------------------
 * Sparse max pooling.  Replaces sparse_conv_ext.indice_maxpool_fp32 / indice_maxpool_backward_fp32
 * (TF/mmdet3d/ops/spconv/src/all.cc:21-51 -> include/spconv/pool_ops.h:26-94, src/maxpool.cc:22-66, maxpool_cuda.cu):
 *   forward   out[o, c] = max(0, max over offsets k with nbr[k][o] >= 0 of features[nbr[k][o], c])
 *             (the reference starts from a ZERO output and only raises it; kept bug for bug)
 *   backward  grad_in[i, c] = sum over offsets k (ascending, like the reference's loop) with o = inv[k][i] >= 0 and
 *             out_features[o, c] == features[i, c] of grad_out[o, c];  inv = df3d_invert_neighbors(nbr).
 * One launch each over the neighbour table instead of one launch per kernel offset; channels % 4 == 0.
 */
int df3d_sparse_maxpool(const float *features, int n_in, int channels, const int32_t *nbr, int kvol, int n_out, float *out,
                        void *stream);
int df3d_sparse_maxpool_backward(const float *features, const float *out_features, const float *grad_out, int n_in,
                                 int channels, const int32_t *inv, int kvol, float *grad_in, void *stream);

/* ------------------------------------------------------------------------------------
 * Dynamic voxelisation.  Replaces voxel_layer.dynamic_voxelize (TF/mmdet3d/ops/voxel/src/voxelization.h:71-86,
 * voxelization_cpu.cpp:8-41,147-171, voxelization_cuda.cu:11-45,328-373): coors[i] = (z, y, x) voxel of point i,
 * c = floor((p - range_min) / voxel_size) per axis in fp32, grid = round((max - min) / voxel_size); all three -1 when
 * the point lies outside the grid.  points [P, F >= 3] f32, voxel_size [3], coors_range [6] (host), coors [P, 3] i32.
 */
int df3d_dynamic_voxelize(const float *points, long long num_points, int num_features, const float *voxel_size,
                          const float *coors_range, int32_t *coors, void *stream);

/* ------------------------------------------------------------------------------------
 * The k smallest 64-bit keys of each of `segments` equally long key arrays, in ascending order: the selection step
 * of both detection heads (keys = [segment | 0x3F800000 - score bits | index], so ascending key = descending score,
 * ties by ascending index).  Replaces the full sorts of the reference -- the per-(task, sample) score sort before
 * rotate_nms_pcdet (CP/det3d/models/bbox_heads/center_head.py:470-478, box_torch_ops.py:248-279) and the argsort of
 * all C*H*W scores (TF/mmdet3d/models/dense_heads/transfusion_head.py:866) -- by a two-level radix select
 * (histogram, histogram, compaction, one-workgroup sort); exact for any input.
 *   keys [segments, n] u64 (bits 63:56 equal within a segment; the all-ones key is "masked"), out [segments, k] u64
 *   (padded with all-ones keys when n < k), out_count [segments] i32 or NULL = keys of out below the all-ones key.
 *   1 <= k <= 4096.
 */
size_t df3d_topk_keys_workspace_bytes(int segments, long long n, int k);
int df3d_topk_keys(const unsigned long long *keys, int segments, long long n, int k, unsigned long long *out,
                   int32_t *out_count, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * Cross-attention of object queries over the BEV map (TransFusionHead decoder layer, transfusion_head.py:110-113 ->
 * multi_head_attention_forward :255-505, the bmm / softmax / bmm of :478-495):
 *   out[b, q, h*16 + d] = sum_k softmax_k(scale * <Q[b,q,h,:], K[b,k,h,:]>) * V[b,k,h,d]
 * q [B*nq, ld_q], k / v [B*nk, ld_k / ld_v], out [B*nq, ld_out] f32 row views whose first heads*16 columns are the
 * projected operands (head h in columns h*16 .. h*16+15); head_dim must be 16.  Keys are split over workgroups and
 * merged by log-sum-exp; the [B*heads, nq, nk] score tensor is never materialised.
 */
size_t df3d_cross_attention_workspace_bytes(int batch, int heads, int nq, int nk);
int df3d_cross_attention(const float *q, int ld_q, const float *k, int ld_k, const float *v, int ld_v, int batch, int nq,
                         int nk, int heads, int head_dim, float scale, float *out, int ld_out, void *workspace,
                         size_t workspace_bytes, void *stream);
/* training (round 6).  ..._train: the same product with dropout of the probabilities (F.dropout(attn_output_weights, p) of
 * multi_head_attention_forward, transfusion_head.py:489-490; keep(e) = the hash of df3d_relu_dropout over the element index
 * e = ((b * heads + h) * nq + q) * nk + key; p = 0: none) and the base-2 log-sum-exp of the scaled scores lse [B, heads, nq] for
 * the backward; same workspace.  ..._backward: dq [B*nq, ld_dq], dk / dv [B*nk, ld_dk / ld_dv] (first heads*16 columns written;
 * dq is zeroed by the call) from grad_out [B*nq, ld_do], lse and delta [B, heads, nq] = <grad_out, out> per head; the
 * probabilities are recomputed tile by tile on the fp32 matrix cores, the [B*heads, nq, nk] tensors never exist.  nq <= 256,
 * head_dim 16, strides multiples of 4 floats, 16-byte aligned operands. */
int df3d_cross_attention_train(const float *q, int ld_q, const float *k, int ld_k, const float *v, int ld_v, int batch, int nq,
                               int nk, int heads, int head_dim, float scale, float p, unsigned long long seed, float *out,
                               int ld_out, float *lse, void *workspace, size_t workspace_bytes, void *stream);
int df3d_cross_attention_backward(const float *q, int ld_q, const float *k, int ld_k, const float *v, int ld_v,
                                  const float *grad_out, int ld_do, const float *lse, const float *delta, int batch, int nq,
                                  int nk, int heads, int head_dim, float scale, float p, unsigned long long seed, float *dq,
                                  int ld_dq, float *dk, int ld_dk, float *dv, int ld_dv, void *stream);

/* ------------------------------------------------------------------------------------
 * Multi-scale deformable attention, forward.  Replaces
 * MultiScaleDeformableAttention.ms_deform_attn_forward (CP/det3d/models/model_utils/ops/src/
 * vision.cpp:13-16, ms_deform_attn.h:21-40, cuda/ms_deform_attn_cuda.cu:20-84, kernel
 * ms_deform_im2col_cuda.cuh:237-299).  value [N,S,M,D] f32, spatial_shapes [L,2] i64 (H,W),
 * level_start_index [L] i64 (both on the device, as the reference passes them),
 * sampling_loc [N,Lq,M,L,P,2] f32 (x,y in [0,1]), attn_weight [N,Lq,M,L,P] f32,
 * out [N,Lq,M*D] f32.  No im2col_step chunking is needed (kept in the Python wrapper's
 * signature only).
 * ---------------------------------------------------------------------------------- */
int df3d_ms_deform_attn_forward(const float *value, const int64_t *spatial_shapes,
                                const int64_t *level_start_index, const float *sampling_loc,
                                const float *attn_weight, int N, int S, int M, int D, int Lq,
                                int L, int P, float *out, void *stream);

/* Backward of the above.  Replaces MultiScaleDeformableAttention.ms_deform_attn_backward
 * (vision.cpp:13-16, ms_deform_attn.h:42-62, cuda/ms_deform_attn_cuda.cu:87-153, kernels
 * ms_deform_im2col_cuda.cuh:87-232,301-921).  grad_output [N,Lq,M*D]; grad_value [N,S,M,D] (zero-filled here, then
 * accumulated with fp32 atomics -- summation order varies between runs, as in the reference), grad_sampling_loc
 * [N,Lq,M,L,P,2], grad_attn_weight [N,Lq,M,L,P] (plain stores). */
int df3d_ms_deform_attn_backward(const float *value, const int64_t *spatial_shapes, const int64_t *level_start_index,
                                 const float *sampling_loc, const float *attn_weight, const float *grad_output, int N,
                                 int S, int M, int D, int Lq, int L, int P, float *grad_value,
                                 float *grad_sampling_loc, float *grad_attn_weight, void *stream);
/* round 6 -- the same gradients for ONE single-level map [N, H * W, M, 16] WITHOUT global atomics (csrc/msda.hip msda_bin_*): the
 * sampling points are counting-sorted by (map, 8 x 8 pixel tile, head); a wave per <= 512 points of a bin evaluates the col2im of
 * its 9 x 9 pixel footprint as a matrix product on v_mfma_f32_16x16x4_f32 (bilinear weights x gradient rows, exact fp32 products)
 * and stores it as a slab; a gather kernel adds the slabs and the neighbours' halos up -- every element of grad_value is written
 * once (no zero fill).  Same contract as df3d_ms_deform_attn_backward (which the reference's
 * ms_deform_attn_backward, CP/det3d/models/model_utils/ops/src/vision.cpp:13-16 -> ms_deform_attn_cuda.cu:93-153, maps to);
 * H, W: the map's size on the host; workspace of ..._workspace_bytes(N, M, Lq, P, H, W) bytes, `slabs` of ..._slab_bytes(...).
 * Served: D == 16, P <= 16, ceil(H / 8) * ceil(W / 8) * M <= 7680 (the LDS histogram), N <= 65535. */
size_t df3d_ms_deform_attn_backward_binned_workspace_bytes(int N, int M, int Lq, int P, int H, int W);
size_t df3d_ms_deform_attn_backward_binned_slab_bytes(int N, int M, int D, int Lq, int P, int H, int W);   /* `slabs` scratch */
int df3d_ms_deform_attn_backward_binned(const float *value, const int64_t *spatial_shapes, const int64_t *level_start_index,
                                        const float *sampling_loc, const float *attn_weight, const float *grad_output, int N,
                                        int M, int D, int Lq, int P, int H, int W, float *grad_value, float *grad_sampling_loc,
                                        float *grad_attn_weight, void *workspace, size_t workspace_bytes, float *slabs, void *stream);

/* Self-attention inside small token groups: nn.MultiheadAttention's scaled-dot-product core for the LocalTransformer of
 * ACTRv2 (VR/pcdet/models/backbones_3d/.../pointformer.py:10-44, 232-262): qkv [tokens*groups][3*heads*16] fp32 rows in
 * sequence-first order (row = token * groups + group; the in-projection's q | k | v blocks) -> out [tokens*groups][heads*16];
 * softmax(q k^T / 4) v per group and head, no masks. */
int df3d_group_attention(const float *qkv, int tokens, int groups, int heads, int head_dim, float *out, void *stream);
/* ... also (or only: out may be NULL) as split rows [tokens*groups][heads*16 * 4 bytes] (per 8 channels 16 B bf16 hi | 16 B
 * bf16 lo), what the out-projection on the split-precision kernels reads. */
int df3d_group_attention_split(const float *qkv, int tokens, int groups, int heads, int head_dim, float *out, void *out_split,
                               void *stream);

/* Grouped features + positional MLP of the LocalTransformer (pointformer.py:232-262) in one pass:
 * out[r] = feat[sel[r]] + W1 relu(W0 xyz[r] + b0) + b1 with feat [*, channels], sel [rows] int64, xyz [rows][3], W0 [hidden][3]
 * (BatchNorm folded), W1 [channels][hidden]. */
int df3d_pe_gather_add(const float *feat, const int64_t *sel, const float *xyz, const float *w0, const float *b0, const float *w1,
                       const float *b1, long long rows, int channels, int hidden, float *out, void *stream);

/* ------------------------------------------------------------------------------------
 * Point ops of LocalTransformer (CP/det3d/models/model_utils/pointformer.py:349-380).
 * furthest_point_sampling_wrapper (CP/det3d/ops/furthest_point_sample/src/
 *   furthest_point_sample.cpp, kernel furthest_point_sample_cuda.cu:26-141): xyz [B,N,3] ->
 *   idx [B,m] i32; temp [B,N] f32 scratch (the reference's caller fills it with 1e10; we
 *   initialise it ourselves).  Tie rule identical to the reference block reduction.
 * ball_query_wrapper (CP/det3d/ops/ball_query/src/ball_query_cuda.cu:11-54).
 * group_points (group_points_cuda.cu:56-78): features [B,C,N], idx [B,npoint,nsample] ->
 *   out [B,C,npoint,nsample].  gather_points (gather_points_cuda.cu:8-24): idx [B,npoint]
 *   -> out [B,C,npoint].
 * ---------------------------------------------------------------------------------- */
int df3d_furthest_point_sample(const float *xyz, int B, int N, int m, float *temp, int32_t *idx,
                               void *stream);
int df3d_ball_query(const float *new_xyz, const float *xyz, int B, int N, int m,
                    float min_radius, float max_radius, int nsample, int32_t *idx, void *stream);
int df3d_group_points(const float *features, const int32_t *idx, int B, int C, int N, int npoint,
                      int nsample, float *out, void *stream);
int df3d_gather_points(const float *features, const int32_t *idx, int B, int C, int N, int npoint,
                       float *out, void *stream);
/* The remaining entry points of the point-op extension modules (round 3):
 *   furthest_point_sampling_with_dist_wrapper (TF/mmdet3d/ops/furthest_point_sample/src/furthest_point_sample.cpp:45-57,
 *     furthest_point_sample_cuda.cu:213-330): dist [B,N,N] pairwise distances instead of coordinates, same scan / tie rule;
 *   group_points_ext.backward (group_points/src/group_points.cpp:35-47, group_points_cuda.cu:10-31) and
 *   gather_points_grad_wrapper (gather_points/src/gather_points.cpp:38-51, gather_points_cuda.cu:48-70):
 *     grad_points [B,C,N] += grad_out scattered by idx (the caller zeroes grad_points, as the reference's wrappers do). */
int df3d_furthest_point_sample_with_dist(const float *dist, int B, int N, int m, float *temp, int32_t *idx, void *stream);
int df3d_group_points_grad(const float *grad_out, const int32_t *idx, int B, int C, int N, int npoint, int nsample,
                           float *grad_points, void *stream);
int df3d_gather_points_grad(const float *grad_out, const int32_t *idx, int B, int C, int N, int npoint, float *grad_points,
                            void *stream);

/* ------------------------------------------------------------------------------------
 * Camera-fusion glue of the CenterPoint adapter (rows a7-a9 of SURVEY.md §8a).  The reference has
 * no native binding here: these replace the Python loops of
 * CP/det3d/models/fusion/voxel_with_point_projection.py:131-385,
 * point_to_image_projection.py:63-231 and model_utils/attention.py:422-468 (pts2img).
 * Image index convention: img = b * ncam + cam (the order ACTR receives, :330-342).
 *
 * df3d_project_voxels: per (camera, voxel): voxel index * scale_xyz + pc_min -> lidar2cam
 *   [B,ncam,4,4] -> intrinsic [B,ncam,3,3] -> pixel, truncated as the reference does (.long(),
 *   * image_scale .long(), * feat_scale .long()); mask = 0<x<W_raw, 0<y<H_raw, depth > thres.
 *   grid_xy [ncam,n,2] i32 (feature-map x,y; 0 where masked), mask [ncam,n] u8,
 *   point_inv [n,3] f32 (LiDAR xyz), depth [ncam,n] f32 or NULL.
 *   aug_inv [B][30] f32 or NULL: the inverse 3-D augmentation of every sample, applied to the voxel corner before the
 *   camera transform in the reference's order (point_to_image_projection.py:121-128, batch_dict['aug_matrix_inv']):
 *   + translate [3], then the row vector times rescale / rotate / flip [3][3] each (identity / zero where absent).
 * df3d_scatter_to_image ("pts2img"): canvas [B*ncam, C+3, H, W] <- (features | xyz) of the
 *   visible voxels, last writer wins (highest row); winner [B*ncam,H,W] i32 is scratch.
 * df3d_assemble_queries: zero-padded per-camera query tensors for ACTR; pos [ncam,n] i32 = slot
 *   of a visible voxel inside its (b, cam) list (exclusive count of visible rows before it).
 * df3d_fusion_writeback: out[row] = features[row] + sum_cam enh[b*ncam+cam][pos] in camera order
 *   (voxel_with_point_projection.py:368-377).
 * ---------------------------------------------------------------------------------- */
int df3d_project_voxels(const int32_t *indices, int n, int batch, int ncam,
                        const float *scale_xyz_host, const float *pc_min_host,
                        const float *lidar2cam, const float *intrinsic, const int32_t *raw_hw,
                        const float *depth_thres, float image_scale, const float *feat_scale,
                        int32_t *grid_xy, uint8_t *mask, float *point_inv, float *depth, const float *aug_inv,
                        void *stream);
int df3d_scatter_to_image(const float *features, const float *point_inv, const int32_t *indices,
                          const int32_t *grid_xy, const uint8_t *mask, int n, int channels,
                          int batch, int ncam, int H, int W, int32_t *winner, float *canvas,
                          void *stream);
int df3d_assemble_queries(const float *features, const float *point_inv, const int32_t *indices,
                          const int32_t *grid_xy, const uint8_t *mask, const int32_t *pos,
                          const float *img_feats, int n, int channels, int img_channels, int batch,
                          int ncam, int H, int W, int max_ne, float *v_feat, float *v_i_feat,
                          float *qgrid, float *qpts, void *stream);
int df3d_fusion_writeback(const float *features, const float *enh, const int32_t *indices,
                          const uint8_t *mask, const int32_t *pos, int n, int channels, int ncam,
                          int max_ne, float *out, void *stream);

/* Canvas-free image gate (attention.py:31-61 refactored by linearity, see csrc/fusion.hip):
 * df3d_gate_scatter: S [B*ncam, 9, H, W] += s9 [n, 9] at the winning voxel pixel of each (image, pixel)
 *   (last writer = highest row, as pts2img); clear != 0 zero-fills S first; winner is scratch [B*ncam,H,W].
 * df3d_gate_finish: att [nimg,H,W] = sigmoid(kg[18] + sum_taps inside*(kg[t] + kg[9+t]*gate[p+t] + S[t][p+t])).
 * df3d_assemble_queries2: df3d_assemble_queries that samples the UNGATED image and multiplies by att at the
 *   query pixel, and also writes the depth sine position embedding qpos [B*ncam, max_ne, C]
 *   (position_encoding.py:107-120); att / qpos may be NULL.  The image is either one tensor img_feats
 *   [B*ncam, Ci, H, W] or (img_feats NULL) a device table img_ptrs of B*ncam pointers to [Ci, H, W] maps, which is
 *   how the reference holds them (one dict entry per camera) -- no stacking copy.  counts (optional, from
 *   df3d_query_slots): when given, only the padding rows (slot >= counts[image]) are zeroed instead of the whole
 *   padded tensors. */
/* df3d_scatter_winner: winner [B*ncam, H, W] = highest voxel row projected onto each pixel (-1: none) -- the "last writer"
 * of the reference's pts2img index_put (voxel_with_point_projection.py:283-352) on its own, for callers that gather the
 * canvas themselves (the differentiable training path of dualfusion.fusion). */
int df3d_scatter_winner(const int32_t *indices, const int32_t *grid_xy, const uint8_t *mask, int n, int batch, int ncam,
                        int H, int W, int32_t *winner, void *stream);
int df3d_gate_scatter(const float *s9, const int32_t *indices, const int32_t *grid_xy, const uint8_t *mask,
                      int n, int batch, int ncam, int H, int W, int32_t *winner, float *S, int clear, void *stream);
/* df3d_gate_scatter with the row responses computed inside (round 3): s9[row] = T [9][channels + 3] . (features[row],
 * point_inv[row]) for the winning rows only -- replaces the caller's cat + GEMM per scale (attention.py:31-61 by linearity). */
int df3d_gate_scatter_rows(const float *features, int channels, const float *point_inv, const float *T,
                           const int32_t *indices, const int32_t *grid_xy, const uint8_t *mask, int n, int batch, int ncam,
                           int H, int W, int32_t *winner, float *S, int clear, void *stream);
int df3d_gate_finish(const float *gate, const float *S, const float *kg, int nimg, int H, int W, float *att,
                     void *stream);
/* round 4: the same steps with fewer passes around them --
 * df3d_gate_rows: df3d_gate_scatter_rows for a winner map the caller already holds (df3d_scatter_winner: it depends on the
 *   voxel coordinates alone, so the frame head builds it a frame ahead);
 * df3d_gate_finish_bias: df3d_gate_finish on an image summary WITHOUT the bias of its 1x1 convolution, gate_bias [1] added
 *   inside (attention.py:31-61: conv bias);
 * df3d_fusion_writeback_split: df3d_fusion_writeback with 16-byte accesses and, optionally, the split rows of the result
 *   (out_split may be NULL), which the convolution behind the adapter reads. */
int df3d_gate_rows(const float *features, int channels, const float *point_inv, const float *T, const int32_t *winner,
                   int nimg, int H, int W, float *S, int clear, void *stream);
int df3d_gate_finish_bias(const float *gate, const float *gate_bias, const float *S, const float *kg, int nimg, int H, int W,
                          float *att, void *stream);
int df3d_fusion_writeback_split(const float *features, const float *enh, const int32_t *indices, const uint8_t *mask,
                                const int32_t *pos, int n, int channels, int ncam, int max_ne, float *out, void *out_split,
                                void *stream);
/* df3d_query_slots: pos[cam][i] = number of visible voxels (mask[cam][.] != 0) of voxel i's sample before i,
 *   counts[b*ncam + cam] = visible voxels of sample b in camera cam (voxel_with_point_projection.py:318-335 builds
 *   these lists with boolean indexing per camera); indices must be batch-sorted. */
int df3d_query_slots(const uint8_t *mask, const int32_t *indices, int n, int batch, int ncam, int32_t *pos,
                     int32_t *counts, void *stream);
int df3d_assemble_queries2(const float *features, const float *point_inv, const int32_t *indices,
                           const int32_t *grid_xy, const uint8_t *mask, const int32_t *pos,
                           const float *img_feats, const float *const *img_ptrs, const float *att, int n,
                           int channels, int img_channels, int batch, int ncam, int H, int W, int max_ne,
                           float *v_feat, float *v_i_feat, float *qgrid, float *qpts, float *qpos,
                           const int32_t *counts, void *stream);
/* round 4 -- df3d_assemble_queries2 BY SLOT: a wave per slot of the padded [B*ncam][max_ne] tensors instead of a wave per
 * (camera, voxel) candidate (five of six candidates own no slot, and every live wave walked mask -> slot -> sample -> pixel before
 * its gathers).  slot_rows [B*ncam*max_ne][4] i32 is scratch (slot -> voxel row, pixel x, pixel y, compact row; filled by a
 * one-thread-per-candidate pass);
 * counts is required; padding rows are written by the same launch.  pixrow + compact (both or neither): the image features
 * come from the pixel-major rows compact[pixrow[image][pixel]] (df3d_query_pixel_rows + df3d_imgproj_split_compact) instead of
 * the channel-first maps.  Same values (voxel_with_point_projection.py:337-377). */
int df3d_assemble_queries2_slots(const float *features, const float *point_inv, const int32_t *indices,
                                 const int32_t *grid_xy, const uint8_t *mask, const int32_t *pos,
                                 const float *img_feats, const float *const *img_ptrs, const float *att, int n,
                                 int channels, int img_channels, int batch, int ncam, int H, int W, int max_ne,
                                 float *v_feat, float *v_i_feat, float *qgrid, float *qpts, float *qpos,
                                 const int32_t *counts, int32_t *slot_rows, const int32_t *pixrow, const float *compact,
                                 void *stream);
/* round 4 -- the queries' image features from PIXEL-MAJOR rows instead of 256 scattered elements of the channel-first map:
 * df3d_query_pixel_rows: pixrow [batch*ncam*H*W] i32 = rank of every pixel some visible voxel projects to (image-major, -1
 *   elsewhere), *total (device i32) = their number; depends on the projection alone (the frame head runs it a frame ahead);
 * df3d_imgproj_split_compact: df3d_imgproj_split that also copies the raw 256-channel rows of exactly these pixels into
 *   compact [total][256] (the tile is in LDS anyway);
 * df3d_assemble_queries2_compact: df3d_assemble_queries2 reading compact[pixrow[image][pixel]] -- same values. */
size_t df3d_query_pixel_rows_workspace_bytes(int batch, int ncam, int H, int W);
int df3d_query_pixel_rows(const int32_t *indices, const int32_t *grid_xy, const uint8_t *mask, int n, int batch, int ncam, int H,
                          int W, int32_t *pixrow, int32_t *total, void *workspace, size_t workspace_bytes, void *stream);
int df3d_assemble_queries2_compact(const float *features, const float *point_inv, const int32_t *indices, const int32_t *grid_xy,
                                   const uint8_t *mask, const int32_t *pos, const int32_t *pixrow, const float *compact,
                                   const float *att, int n, int channels, int img_channels, int batch, int ncam, int H, int W,
                                   int max_ne, float *v_feat, float *v_i_feat, float *qgrid, float *qpts, float *qpos,
                                   const int32_t *counts, void *stream);

/* ------------------------------------------------------------------------------------
 * Split-precision sparse convolution (csrc/spconv_split.hip): same contract as df3d_sparse_conv_fused
 * (replaces the same reference entry points, spconv_ops.h:260-361 + the BN/ReLU/residual passes), evaluated
 * on the bf16 matrix cores.  Every fp32 operand x is split once into hi = bf16(x), lo = bf16(x - hi) and the
 * contraction is A_hi*W_hi + A_lo*W_hi + A_hi*W_lo with fp32 accumulation: ~1e-5 relative error against the
 * exact fp32 result (parity bar 1e-3), 16/3 of the fp32 MFMA rate.
 *   df3d_conv_packed_weight_bytes: bytes of the packed filter bank, 0 when (cin, cout) has no split kernel
 *                                  (served: 32->32, 32->64, 64->64, 64->128, 128->128, 128->256, 256->128, 256->256)
 *   df3d_conv_pack_weights:  filters [kvol][cin][cout] fp32 -> packed MFMA B operands (once per weight)
 *   df3d_split_rows:         features [n][c] fp32 -> split rows [n][c/8][hi 8 x bf16 | lo 8 x bf16], c % 8 == 0
 *   df3d_sparse_conv_split:  out fp32 [n_out][cout]; out_split (optional) receives the split rows of `out`
 *                            so that the next convolution needs no df3d_split_rows pass.  out may be NULL when
 *                            out_split is given (a layer whose only reader is the next split-precision convolution).
 * ---------------------------------------------------------------------------------- */
size_t df3d_conv_packed_weight_bytes(int kvol, int cin, int cout);
int df3d_conv_pack_weights(const float *filters, int kvol, int cin, int cout, void *packed, void *stream);
/* `groups` filter banks [groups][K][Cin][Cout] (Cout <= 128) packed back to back for df3d_conv_rows_split: one launch. */
int df3d_conv_pack_weights_groups(const float *filters, int groups, int kvol, int cin, int cout, void *packed,
                                 void *stream);
int df3d_split_rows(const float *features, long long n, int c, void *split, void *stream);
/* Gradient rows (training): features [n][c] fp32 -> split rows of s * features, s = the power of two that puts the tensor's largest
 * |value| into [512, 1024) (a per-tensor block scale: the fixed-scale fp16 pair format holds activations, not gradients of arbitrary
 * magnitude).  scale [df3d_pow2_scale_floats()] receives s in its first entry (then the largest |value| and the reduction's workspace), inv_scale [inv_channels] receives 1 / s in every entry -- the per-channel `scale` vector of the
 * convolution that consumes the rows (its epilogue multiplies by it: exact, a power of two).  Replaces the three-bf16-part rows of
 * the input-gradient convolutions (indice_conv_backward's data gradient, spconv_ops.h:363-456). */
int df3d_split_rows_scaled(const float *features, long long n, int c, void *split, float *scale, float *inv_scale, int inv_channels,
                           void *stream);
int df3d_sparse_conv_split(const void *features_split, int n_in, int cin, const void *packed_filters, int kvol,
                           int cout, const int32_t *nbr, int n_out, const float *bias, const float *scale,
                           const float *shift, const float *residual, int relu, float *out, void *out_split,
                           const int32_t *tile_rows, int ntiles, void *stream);

/* ------------------------------------------------------------------------------------
 * Fused element-wise stages of the dual-query encoder layer (csrc/actr.hip); the reference runs them as
 * separate torch ops (actr_transformer.py:399-426, ms_deform_attn.py:129-166, attentions.py:111-117).
 * Rows = B*ncam*Q query rows of C channels, fp32, contiguous.
 *   df3d_actr_prep:      A = q + pos, Bw = (q + pos) + (qi + pos)
 *   df3d_add_layernorm:  out = LayerNorm(x + y) (y may be NULL), eps as nn.LayerNorm
 *   df3d_bigate_sum:     BiGateSum1D_2: g = q + qi; q' = q + qi*sigmoid(g.wb+bb); qi' = qi + q*sigmoid(g.wa+ba)
 *   df3d_ms_deform_attn_fused: df3d_ms_deform_attn_forward taking the RAW outputs of the sampling_offsets
 *     and attention_weights linears plus the 2-D reference points [N,Lq,2]: softmax over L*P and
 *     loc = ref + off / (W_l, H_l) happen in the kernel.  value_stride = floats between consecutive pixels
 *     of `value` (M*D when contiguous; larger when several layers' value projections share one buffer).
 *     Optional pixel_scale [N,S] / image_bias (rows bias_stride floats apart): value(p) = scale_p*value[p] + bias_n.
 * ---------------------------------------------------------------------------------- */
int df3d_actr_prep(const float *q, const float *qi, const float *pos, long long rows, int C, float *A, float *Bw,
                   void *stream);
/* training (round 6): h <- relu(h) . keep / (1 - p) over n floats IN PLACE -- the activation + dropout between the two linears of
 * forward_ffn (CP/det3d/models/model_utils/actr_transformer.py:309-311, 388-395: linear2(dropout(activation(linear1(src))))) as
 * one pass over the [rows x d_ffn] hidden tensor.  keep(i) = hash(i, seed) >= p (24 uniform bits; p = 0: plain ReLU); the
 * kept-and-active elements are the non-zeros of the result, so ..._backward (grad_in = grad / (1 - p) where h != 0, else 0;
 * h = the forward's result) takes no mask.  16-byte aligned pointers. */
int df3d_relu_dropout(float *h, long long n, float p, unsigned long long seed, void *stream);
int df3d_relu_dropout_backward(const float *h, const float *grad, long long n, float p, float *grad_in, void *stream);
/* the same over bfloat16 elements (the bf16 mixed-precision mode: hidden rows produced by a bfloat16 product) */
int df3d_relu_dropout_bf16(void *h, long long n, float p, unsigned long long seed, void *stream);
int df3d_relu_dropout_backward_bf16(const void *h, const void *grad, long long n, float p, void *grad_in, void *stream);
/* training (round 6): out = LayerNorm(x + dropout(y, p)) over [rows, C] fp32 rows -- the residual steps of the encoder layers
 * (CP/det3d/models/model_utils/actr_transformer.py:311-312, 330-331, 389-390, 395-396, 416-417) -- in one kernel; also writes the
 * normalised rows `xhat` [rows, C] and `rstd` [rows] for the backward.  keep(i) = the hash of df3d_relu_dropout over the element
 * index row * C + c (p = 0: no dropout).  ..._backward: dx = d out / d x [rows, C]; dy = dx . keep / (1 - p) (may be NULL when
 * p = 0: dy = dx); dgamma / dbeta [C] are ADDED to (zero them first).  C % 4 == 0, C <= 1024. */
int df3d_dropout_add_layernorm(const float *x, const float *y, const float *gamma, const float *beta, float eps, float p,
                               unsigned long long seed, long long rows, int C, float *out, float *xhat, float *rstd, void *stream);
int df3d_dropout_add_layernorm_backward(const float *grad, const float *xhat, const float *rstd, const float *gamma, float p,
                                        unsigned long long seed, long long rows, int C, float *dx, float *dy, float *dgamma,
                                        float *dbeta, void *stream);
int df3d_add_layernorm(const float *x, const float *y, const float *gamma, const float *beta, float eps,
                       long long rows, int C, float *out, void *stream);
/* df3d_add_layernorm that also writes the split rows of its output (C % 8 == 0) for a following split-precision layer. */
int df3d_add_layernorm_split(const float *x, const float *y, const float *gamma, const float *beta, float eps, long long rows,
                             int C, float *out, void *out_split, void *stream);
int df3d_bigate_sum(const float *q, const float *qi, const float *wb, const float *bb, const float *wa,
                    const float *ba, long long rows, int C, float *q_out, float *qi_out, void *stream);
int df3d_ms_deform_attn_fused(const float *value, long long value_stride, const int64_t *spatial_shapes,
                              const int64_t *level_start_index, const float *ref_xy, const float *offsets,
                              const float *logits, const float *pixel_scale, const float *image_bias,
                              long long bias_stride, int N, int S, int M, int D, int Lq, int L, int P,
                              float *out, void *stream);
/* The same sampler on bf16 value rows (value_stride in bf16 elements): the reduced-precision mode of the reference's
 * fp16-AMP configurations (TF/configs/...: fp16 = dict(loss_scale=512.)); offsets, logits, weights, the accumulation and
 * the output stay fp32.  Produced by df3d_value_fold_gemm_bf16. */
int df3d_ms_deform_attn_fused_bf16(const void *value_bf16, long long value_stride, const int64_t *spatial_shapes,
                                   const int64_t *level_start_index, const float *ref_xy, const float *offsets,
                                   const float *logits, const float *pixel_scale, const float *image_bias,
                                   long long bias_stride, int N, int S, int M, int D, int Lq, int L, int P,
                                   float *out, void *stream);

/* GroupNorm of the gated input projection folded into the value projection (actr.py:139-149 input_proj[l] =
 * Conv2d 1x1 + GroupNorm, ms_deform_attn.py:139 value_proj).  With x = a_p*u + b (u = W_ip*img without bias,
 * a_p the per-pixel image gate, a = NULL -> 1):
 *   df3d_scaled_moments: moments[n][c] = (sum_p a_p u_cp, sum_p (a_p u_cp)^2), u channel-first with the given
 *     image / channel strides (floats);
 *   df3d_groupnorm_fold: per image, Wf[n][o][c] = W[o][c]*rstd_g*gamma_c and cf[n][o] so that
 *     W*GroupNorm(x)+wb = a_p * (Wf[n] u_p) + cf[n].  The sampler applies a_p and cf through
 *     pixel_scale / image_bias, so neither x, GroupNorm(x) nor its transpose is ever materialised. */
/* df3d_rows_groupnorm: GroupNorm of [N, Q, C] rows with statistics per (n, group) over Q x C/groups values -- what
 * i_input_proj's GroupNorm computes on the Conv1d layout [N, C, Q] (actr.py:150-158) -- without transposing;
 * stats = scratch of N*groups*2 doubles. */
int df3d_rows_groupnorm(const float *x, int N, int Q, int C, int groups, const float *gamma, const float *beta,
                        float eps, double *stats, float *out, void *stream);
int df3d_scaled_moments(const float *u, long long image_stride, long long channel_stride, const float *a, int N,
                        int S, int C, double *moments, void *stream);
int df3d_groupnorm_fold(const double *moments, const float *b, const float *gamma, const float *beta, float eps,
                        int N, int S, int C, int groups, const float *W, const float *wb, int O, float *Wf,
                        float *cf, void *stream);

/* ------------------------------------------------------------------------------------
 * Fused feed-forward block of the encoder layer (csrc/ffn.hip):
 *   out = LayerNorm(residual + W2 relu(W1 x + b1) + b2)     (actr_transformer.py:413-424: linear1/activation/
 *   linear2/dropout/norm of either query stream; nn.Linear weights W1 [d_ffn, d_model], W2 [d_model, d_ffn])
 * in one kernel on the bf16 matrix cores with split-precision operands (~1e-5 relative error); the hidden
 * activation stays in registers.  Serves d_model == 128, d_ffn % 128 == 0 (df3d_ffn_packed_bytes returns 0
 * otherwise).  residual / ln_weight+ln_bias may be NULL (no residual / no normalisation).
 * ---------------------------------------------------------------------------------- */
size_t df3d_ffn_packed_bytes(int d_model, int d_ffn);
int df3d_ffn_pack(const float *w1, const float *w2, int d_model, int d_ffn, void *packed, void *stream);
int df3d_ffn_fused(const float *x, long long rows, int d_model, int d_ffn, const void *packed, const float *b1,
                   const float *b2, const float *residual, const float *ln_weight, const float *ln_bias, float eps,
                   float *out, void *stream);
/* Up to four independent FFN jobs of the same sizes in ONE launch (the dual-query layer's image-query and LiDAR-query
 * FFNs, actr_transformer.py:413-424): rows < 0 / 0 skips a job. */
typedef struct df3d_ffn_job {
  const float *x;
  long long rows;
  const void *packed;
  const float *b1, *b2, *residual, *ln_weight, *ln_bias;
  float eps;
  float *out;
} df3d_ffn_job;
int df3d_ffn_fused_jobs(const df3d_ffn_job *jobs, int njobs, int d_model, int d_ffn, void *stream);
/* Reduced-precision mode of the fused feed-forward kernel (BASELINE configs[2], "bf16 with fp32 accumulate"): 1 = the
 * activations and weights enter the matrix cores rounded to bf16 (one product per operand pair instead of three), the
 * accumulation, bias, residual and LayerNorm stay fp32; 0 (default) = split precision, fp32-grade.  Process-wide. */
int df3d_ffn_set_precision(int bf16);

/* ------------------------------------------------------------------------------------
 * Image side of the fusion on the bf16 matrix cores, split precision (csrc/imgproj.hip).  Replaces the 1x1
 * convolutions over the camera maps: the image gate's summary (attention.py:456), ACTR's input_proj conv +
 * GroupNorm (actr.py:139-149) and every layer's value_proj (ms_deform_attn.py:139), for 256-channel maps and a
 * 128-channel model.
 *   df3d_imgproj_pack:    Wcat [rows <= 144][256] fp32 (input_proj weight rows, then the gate's row) -> packed
 *   df3d_imgproj_split:   u = Wcat . img per pixel, from the nimg channel-first maps [256][S] behind img_ptrs:
 *                         u_split [nimg][S][128] split rows (hi|lo bf16, 512 B per pixel) and gate [nimg][S] = row 128
 *   df3d_value_fold_gemm: GroupNorm statistics of x = att*u + conv_bias (att may be NULL), their fold into the
 *                         stacked value projections W [256][128] / wb [256] and value [nimg][S][256] = u . Wf[n]^T;
 *                         cf [nimg][256] is the folded constant: W*GroupNorm(x)+wb = att_p * value_p + cf[n].
 *                         Scratch: moments nimg*128*2 doubles, packed_w nimg*128 KiB.
 * ---------------------------------------------------------------------------------- */
size_t df3d_imgproj_packed_bytes(int rows, int cin);
int df3d_imgproj_pack(const float *wcat, int rows, int cin, void *packed, void *stream);
int df3d_imgproj_split(const float *const *img_ptrs, int nimg, int cin, int S, const void *packed, void *u_split,
                       float *gate, void *stream);
int df3d_imgproj_split_compact(const float *const *img_ptrs, int nimg, int cin, int S, const void *packed, void *u_split,
                               float *gate, const int32_t *pixrow, float *compact, void *stream);
int df3d_value_fold_gemm(const void *u_split, const float *att, int nimg, int S, const float *conv_bias,
                         const float *gn_weight, const float *gn_bias, float eps, int groups, const float *W,
                         const float *wb, double *moments, void *packed_w, float *cf, float *value, void *stream);
/* ... writing the value rows as bf16 [nimg, S, 256] (round to nearest even) for df3d_ms_deform_attn_fused_bf16 */
int df3d_value_fold_gemm_bf16(const void *u_split, const float *att, int nimg, int S, const float *conv_bias,
                         const float *gn_weight, const float *gn_bias, float eps, int groups, const float *W,
                         const float *wb, double *moments, void *packed_w, float *cf, void *value_bf16, void *stream);

/* ------------------------------------------------------------------------------------
 * Native executor for a chain of fused sparse-convolution layers (csrc/executor.hip): the sparse backbones'
 * forward (CP/det3d/models/backbones/scn.py:97-201; TF/mmdet3d/models/middle_encoders/sparse_encoder.py;
 * VR/pcdet/models/backbones_3d/spconv_backbone.py) as one call.  Rulebooks, occupancy directories, features
 * and split rows of every layer are bump-allocated from `arena`; DF3D_ENOMEM (with *arena_used = bytes needed
 * so far) asks for a larger one.  Each layer is conv -> (+bias)*scale+shift -> +residual -> ReLU, i.e. the
 * reference's conv / BatchNorm(eval) / residual add / ReLU group.
 *   kind      0 = SubMConv3d (outputs = inputs), 1 = SparseConv3d (strided; one D2H read of the output count)
 *   input     index of the producing layer, -1 = the network input
 *   residual  layer whose output features are added in the epilogue, -1 = none
 *   rulebook  layers with the same id >= 0 share one neighbour table (the reference's indice_key); -1 = private
 *   packed    split-precision filter bank (df3d_conv_pack_weights) or NULL for the exact fp32 kernels
 * views[i] describes layer i's output inside the arena (grid = its occupancy directory when one was built).
 * The call first builds EVERY rulebook (geometry depends on the coordinates only; all host round trips happen
 * there), then enqueues the convolutions back to back and returns without waiting for them.
 * ---------------------------------------------------------------------------------- */
typedef struct df3d_layer {
  int kind, input, residual, rulebook;
  int cin, cout;
  int ksize[3], stride[3], padding[3], dilation[3];
  int relu;
  int reserved;          /* flags: bit 0 = geometry only (build the rulebook, export it, do not run the conv);
                          *        bit 1 = `packed` holds bf16 weights (df3d_conv_pack_weights_bf16): the layer runs on
                          *                df3d_sparse_conv_bf16 with bf16 rows;
                          *        bit 3 = `packed` holds THREE-part filters (df3d_conv_pack_weights3): the layer runs on
                          *                df3d_conv_rows_split3 with three-part rows ("split3" precision);
                          *        bit 2 = the caller reads this layer's fp32 rows (an exported stage).  When any layer of
                          *                the table carries it, split-precision layers that are neither flagged, nor a
                          *                residual source, nor read by a non-split layer write split rows ONLY (their view
                          *                has features = NULL); a table without the flag keeps fp32 rows everywhere */
  const float *weight;   /* [kvol][cin][cout] fp32 */
  const void *packed;
  const float *bias, *scale, *shift;
} df3d_layer;

typedef struct df3d_layer_view {
  float *features;
  void *split;
  const int32_t *indices;
  void *grid;
  size_t grid_bytes;
  int n, channels, rows_sorted;
  int shape[3];
  const int32_t *nbr;    /* the layer's neighbour table [kvol][n] */
  int kvol;
  int reserved;          /* bit 1: `split` holds bf16 rows [n][channels] instead of split rows; bit 2: three-part rows */
} df3d_layer_view;

/* Optional, before df3d_backbone_run on the same host thread (consumed by that call): `event` (hipEvent_t) marks the input
 * COORDINATES complete -- e.g. recorded by a voxeliser that ran on its own stream.  The geometry stream then waits for this
 * event instead of for everything queued on the caller's stream, so the rulebooks (and the host's round trips for the output
 * counts) of frame k + 1 proceed while frame k is still running on the caller's stream.  The caller guarantees that `arena` is
 * not read or written by work still queued on `stream`, bounds the runs in flight (16 event sets rotate), and that NOTHING
 * it later reads on other streams relied on the old implicit order "host passed a count round trip => the caller's stream has
 * reached this call".  In particular the arena must not be a block a stream-ordered allocator has just recycled from the
 * caller's stream: dualfusion/executor.py rotates three persistent arenas, each guarded by an event.  Without this call the
 * geometry waits for the caller's stream, as before. */
int df3d_backbone_inputs_ready(void *event);
int df3d_backbone_run(const df3d_layer *layers, int nlayers, const float *features, const int32_t *indices, int n,
                      int in_channels, int batch, const int *shape_host, void *arena, size_t arena_bytes,
                      df3d_layer_view *views, size_t *arena_used, void *stream);

/* The same run as TWO calls, for callers that build a frame's geometry ahead of its convolutions -- the reference's
 * data-loader workers voxelise the next sample while the GPU step runs (CP/det3d/datasets/pipelines/preprocess.py `Voxelization`
 * inside the DataLoader's worker processes); here the rulebooks (which depend on the coordinates alone) go ahead as well:
 *   df3d_backbone_geometry  every neighbour table / index set / directory of the table's layers on the calling HOST THREAD's
 *                           geometry stream, which first waits for `inputs_ready` (hipEvent_t: the coordinates are complete;
 *                           NULL = they are complete now).  Blocks the calling thread for the output-count round trips of
 *                           the strided layers only; allocates from `arena` (DF3D_ENOMEM + *arena_used as above).  Fills the
 *                           geometry fields of `views` (indices, n, shape, grid, nbr, kvol; features = split = NULL) and
 *                           returns an opaque `handle`.
 *   df3d_backbone_convs     the fused convolutions of the same table on `stream` (any host thread), after a stream-side wait
 *                           for the geometry; features / split rows come from a SECOND arena (may be retried with a larger one
 *                           after DF3D_ENOMEM); completes `views`.  Launches exactly the kernels df3d_backbone_run launches:
 *                           results are bit-identical.
 *   df3d_backbone_release   frees the handle (not the arenas: the caller owns them and keeps the geometry arena alive while
 *                           anything reads the views).
 * `layers` and `indices` must stay valid until the handle is released. */
int df3d_backbone_geometry(const df3d_layer *layers, int nlayers, const int32_t *indices, int n, int in_channels, int batch,
                           const int *shape_host, void *arena, size_t arena_bytes, void *inputs_ready,
                           df3d_layer_view *views, size_t *arena_used, void **handle);
int df3d_backbone_convs(void *handle, const float *features, void *arena, size_t arena_bytes, df3d_layer_view *views,
                        size_t *arena_used, void *stream);
/* Round 5: the same convolutions range by range, for chains that a feature-modifying step interrupts (the fusion layers of
 * VR/pcdet/models/backbones_3d/spconv_backbone.py:829-916 sit between conv1 / conv2 and behind conv4): the geometry of the
 * WHOLE chain is built ahead (it depends on the coordinates alone), then
 *   df3d_backbone_convs_range(handle, features, first, last, ...)   layers [first, last) on `stream`;
 * `features` of the first range (first == 0) is the network input, of a later range it REPLACES the fp32 rows of layer
 * first - 1 (same index set / channels; their operand split is rebuilt).  Ranges are consecutive and share one arena; a
 * DF3D_ENOMEM of the first range may be retried with a larger arena, of a later one not (the caller finishes the frame on
 * df3d_backbone_run over the remaining layers). */
int df3d_backbone_convs_range(void *handle, const float *features, int first, int last, void *arena, size_t arena_bytes,
                              df3d_layer_view *views, size_t *arena_used, void *stream);
/* ------------------------------------------------------------------------------------
 * Frame head on a native worker thread (round 4): everything of a frame that depends on its RAW INPUTS alone, with all its
 * count round trips, built while the caller still queues the previous frame -- what the reference's DataLoader worker
 * processes do for the voxelisation (CP/det3d/datasets/pipelines/preprocess.py:`Voxelization`, torch DataLoader prefetch),
 * extended to the rulebooks (functions of the voxel coordinates) and the fusion adapter's projection / query slots
 * (coordinates + calibration; CP/det3d/models/fusion/voxel_with_point_projection.py:294-335).
 *   df3d_head_worker_create / _destroy   one worker thread (with its own HIP stream) per detector
 *   df3d_frame_head_submit               copies `desc`, queues the job, returns at once.  The job, on the worker's stream:
 *       df3d_hard_voxelize_batched of every cloud (mean VFE rows [n, C], coors [n, 4]) -> ONE round trip for the voxel counts
 *       -> the geometry phase of df3d_backbone_geometry over `layers` -> df3d_project_voxels of the listed stages and
 *       df3d_query_slots of stage `slots_proj` -> the round trip for the longest camera list.  Every buffer comes from `arena`
 *       (bump allocation; the caller keeps it alive while anything reads the results).
 *   df3d_frame_head_wait                 blocks until the job is done (normally it is), fills `views` (geometry fields), `out`
 *       and `handle` -- the df3d_backbone_geometry handle: df3d_backbone_geometry_wait(handle, stream) before anything reads the
 *       results, then df3d_backbone_convs / df3d_backbone_release as usual.  DF3D_ENOMEM with *arena_used: resubmit with a
 *       larger arena.  Frees the ticket in every case.
 * All device pointers of `desc` (point clouds, calibration tables, the layer table's weights) must stay valid and unchanged
 * until df3d_frame_head_wait returns; the inputs must be COMPLETE in device memory at submit (the worker's stream does not
 * wait for the caller's). */
#define DF3D_HEAD_MAX_PROJ 4
typedef struct df3d_head_project {
  int layer;              /* the stage: index into `layers`, its OUTPUT index set is projected */
  float scale_xyz[3];     /* voxel size of that stage (fp32 voxel size * down-sampling factor) */
  int want_winner;        /* != 0: also df3d_scatter_winner of this stage (the image gate's "last writer" per pixel) */
} df3d_head_project;

typedef struct df3d_frame_head_desc {
  int batch, point_channels;
  const float *const *points;       /* host array of `batch` device pointers [P_b, point_channels] f32 */
  const int *num_points;            /* host array P_b */
  float voxel_size[3], coors_range[6];
  int max_points, max_voxels, break_at_cap;
  const df3d_layer *layers;         /* may be NULL with nlayers = 0: voxelisation only */
  int nlayers;
  int shape[3];
  int ncam;                         /* 0: no camera side */
  const float *lidar2cam, *intrinsic;
  const int32_t *raw_hw;
  const float *depth_thres;
  float image_scale;
  const float *feat_scale;
  const float *aug_inv;             /* may be NULL */
  float pc_min[3];
  int nproj;
  df3d_head_project proj[DF3D_HEAD_MAX_PROJ];
  int slots_proj;                   /* index into proj[] of the stage whose visible voxels become queries; -1: none */
  void *inputs_ready;               /* hipEvent_t or NULL: the worker's stream waits for it first (e.g. calibration tables the
                                     * caller has just queued on a side stream) */
  /* optional: df3d_imgproj_split of the frame's camera maps (it depends on the maps alone), on a SECOND worker stream of
   * default priority, beside the head and whatever the GPU is doing for the previous frame */
  const float *const *img_ptrs;     /* device table of img_count pointers, or NULL */
  int img_count, img_cin, img_pixels;
  const void *img_packed;           /* df3d_imgproj_pack */
  int feat_h, feat_w;               /* camera feature map size (winner maps) */
  int want_pixrow;                  /* != 0: also df3d_query_pixel_rows of the query stage */
} df3d_frame_head_desc;

typedef struct df3d_frame_head_out {
  float *features;                  /* [n, point_channels] mean VFE rows */
  int32_t *coors;                   /* [n, 4] (b, z, y, x) */
  int n;
  int max_ne;                       /* longest (sample, camera) query list */
  int32_t *grid_xy[DF3D_HEAD_MAX_PROJ];   /* per projection: [ncam, proj_n, 2] */
  uint8_t *mask[DF3D_HEAD_MAX_PROJ];      /* [ncam, proj_n] */
  float *point_inv[DF3D_HEAD_MAX_PROJ];   /* [proj_n, 3] */
  int proj_n[DF3D_HEAD_MAX_PROJ];
  int32_t *pos;                     /* [ncam, proj_n[slots_proj]] */
  int32_t *counts;                  /* [batch * ncam] */
  void *img_split;                  /* [img_count][img_pixels][128] split rows, or NULL */
  float *img_gate;                  /* [img_count][img_pixels] */
  void *img_done;                   /* hipEvent_t (owned by the handle): the projection is complete */
  int32_t *winner[DF3D_HEAD_MAX_PROJ];    /* [batch * ncam, feat_h, feat_w] or NULL */
  int32_t *pixrow;                  /* df3d_query_pixel_rows of the query stage ([batch * ncam * feat_h * feat_w]) or NULL */
  int pixrow_total;                 /* pixels that carry a query */
} df3d_frame_head_out;

void *df3d_head_worker_create(int device);
int df3d_head_worker_destroy(void *worker);
int df3d_frame_head_submit(void *worker, const df3d_frame_head_desc *desc, void *arena, size_t arena_bytes, void **ticket);
int df3d_frame_head_wait(void *ticket, df3d_layer_view *views, df3d_frame_head_out *out, size_t *arena_used, void **handle);

/* `stream` waits (on the device) until the geometry of `handle` is complete: for work other than the table's own convolutions
 * that reads the index sets / tables of `views` (e.g. the camera projection of the fusion adapter). */
int df3d_backbone_geometry_wait(void *handle, void *stream);
/* the same for the frame head's image projection (df3d_frame_head_out.img_split / img_gate) */
int df3d_frame_head_image_wait(void *handle, void *stream);
int df3d_backbone_release(void *handle);

/* ------------------------------------------------------------------------------------
 * TransFusion head: target-assignment costs and losses (round 3).
 *
 * df3d_boxes_overlap_bev_xyxyr replaces `iou3d_cuda.boxes_overlap_bev_gpu` of the TransFusion tree
 * (TF/mmdet3d/ops/iou3d/src/iou3d.cpp:66-90 over iou3d_kernel.cu:122-258): boxes [n][5] f32 (x1, y1, x2, y2, angle),
 * out[na][nb] = overlap area.  Called by LiDARInstance3DBoxes.overlaps (core/bbox/structures/base_box3d.py:414-423).
 *
 * df3d_tf_match_cost: what HungarianAssigner3D.assign computes before the host-side linear_sum_assignment
 * (TF/mmdet3d/core/bbox/assigners/hungarian_assigner.py:121-131) for every sample of a batch in ONE launch, from the
 * head's raw predictions: `rows` [batch][proposals][ld] f32 with columns center 0:2 (feature-map units), height 2
 * (gravity centre), dim 3:6 (log), rot 6:8 (sin, cos), class logits at [col_cls, col_cls + num_classes) -- the column
 * order of the reference's `preds` (transfusion_head.py:1264-1268).  Proposals are decoded as
 * TransFusionBBoxCoder.decode does (transfusion_bbox_coder.py:62-71); ground truth gt[total][gt_dim] f32
 * (x, y, z_bottom, w, l, h, yaw, ...), gt_labels[total] i32, sample b owns rows [gt_off[b], gt_off[b+1]) (device i32).
 *   cost[batch][proposals][gmax] = FocalLossCost + BBoxBEVL1Cost + IoU3DCost (columns >= the sample's count: 0),
 *   iou [batch][proposals][gmax] = 3-D IoU (BboxOverlaps3D 'lidar'), boxes[batch][proposals][7] decoded (may be NULL). */
typedef struct df3d_tf_match_cfg {
  float out_size_factor, voxel_size[2], pc_range[2];   /* bbox coder */
  float point_cloud_range[6];                          /* train_cfg */
  float cls_weight, cls_alpha, cls_gamma, cls_eps;     /* FocalLossCost */
  float reg_weight, iou_weight;
} df3d_tf_match_cfg;
int df3d_boxes_overlap_bev_xyxyr(const float *boxes_a, int na, const float *boxes_b, int nb, float *out, void *stream);
int df3d_tf_match_cost(const float *rows, int batch, int proposals, int ld, int col_cls, int num_classes, const float *gt,
                       const int32_t *gt_labels, const int32_t *gt_off, int gt_dim, int gmax, const df3d_tf_match_cfg *cfg,
                       float *cost, float *iou, float *boxes, void *stream);

/* df3d_draw_heatmap_gaussian replaces the per-box Python loop of get_targets_single (transfusion_head.py:1186-1207 over
 * gaussian_radius / draw_heatmap_gaussian, TF/mmdet3d/core/utils/gaussian.py:25-86): heatmap[batch][num_classes][height]
 * [width] f32 is zeroed and every ground-truth box of every sample is splatted with max() in one launch.
 * gaussian_overlap is a double because Python evaluates (1 - o), 4 * o ... in double before torch rounds them. */
typedef struct df3d_tf_splat_cfg {
  float voxel_size[2], out_size_factor, point_cloud_range[2];
  double gaussian_overlap;
  int min_radius;
} df3d_tf_splat_cfg;
int df3d_draw_heatmap_gaussian(const float *gt, const int32_t *gt_labels, const int32_t *gt_off, int gt_dim, int total,
                               int batch, int num_classes, int height, int width, const df3d_tf_splat_cfg *cfg,
                               float *heatmap, void *stream);

/* df3d_gaussian_focal_loss = loss_heatmap of TransFusionHead.loss (transfusion_head.py:1241-1244): mmdet's
 * GaussianFocalLoss(alpha, gamma) of clip_sigmoid(logits) against the dense target, normalised by max(#target == 1, 1)
 * counted on the device (the reference: `.item()`).  logits element (b, c, pix) at b*stride_b + c*stride_c + pix*stride_pix
 * (NCHW or the row kernels' channels-last maps); target and grad contiguous [batch][num_classes][hw].
 *   out[0] = loss, out[1] = #ones, out[2] = loss_weight / max(#ones, 1);  grad (may be NULL) = d(sum)/d logits UNSCALED:
 *   multiply by out[2].  workspace: df3d_gaussian_focal_loss_workspace_bytes(batch * num_classes * hw). */
size_t df3d_gaussian_focal_loss_workspace_bytes(long long n);
int df3d_gaussian_focal_loss(const float *logits, long long stride_b, long long stride_c, long long stride_pix,
                             const float *target, int batch, int num_classes, int hw, float alpha, float gamma,
                             float loss_weight, float *grad, float *out, void *workspace, size_t workspace_bytes,
                             void *stream);

/* df3d_tf_query_loss = the per-layer classification / box losses of TransFusionHead.loss (transfusion_head.py:1246-1281)
 * with the targets of get_targets_single (:1154-1181) built on the fly from the matching: assigned[batch][proposals_all]
 * i32 = global ground-truth row matched to the proposal or -1; proposals_all = layers * proposals (auxiliary heads).
 *   out[2*l], out[2*l+1] = loss_cls, loss_bbox of layer l (FocalLoss / L1Loss, avg_factor max(num_pos, 1));
 *   out[2*layers] = num_pos, out[2*layers+1] = matched_ious;  grad (may be NULL) [batch][proposals_all][ld] = gradient of the
 *   SUM of all layer losses w.r.t. `rows` (class-logit and box-code columns; other columns untouched). */
typedef struct df3d_tf_loss_cfg {
  float encode_step[2];     /* out_size_factor * voxel_size, evaluated in double and rounded once (coder.encode) */
  float pc_range[2];
  float cls_alpha, cls_gamma, cls_loss_weight, bbox_loss_weight, pos_weight;
  float code_weights[DF3D_LOSS_MAX_CODES];
} df3d_tf_loss_cfg;
int df3d_tf_query_loss(const float *rows, const int32_t *assigned, const float *iou, int batch, int proposals_all,
                       int proposals, int ld, int col_cls, int num_classes, int code_size, const float *gt,
                       const int32_t *gt_labels, const int32_t *gt_off, int gt_dim, int gmax, const df3d_tf_loss_cfg *cfg,
                       float *grad, float *out, void *stream);

/* ------------------------------------------------------------------------------------
 * Query-side dense layers of the dual-query fusion encoder layer on the matrix cores (round 3, csrc/rowlinear.hip):
 * y = W x + b for fp32 rows [rows][cin] (cin 128 | 256) and cout <= 128 outputs, fp32-grade (operands split into bf16
 * hi + lo on load, three MFMA products).  Replaces the library GEMMs behind `sampling_offsets` / `attention_weights`
 * (CP/det3d/models/model_utils/ops/modules/ms_deform_attn.py:129-157 on the mixed queries of actr_transformer.py:399-411),
 * `output_proj` + residual + `norm1` (ms_deform_attn.py:188, actr_transformer.py:412-414) and `i_input_proj`'s 1x1 Conv1d
 * (actr.py:96-104).
 *   operands: a0 = x0 (+ x2), a1 = a0 + (x1 + x2) (x1, x2 may be NULL); output columns < csplit_cols (a multiple of 16)
 *   use a0, the others a1.  Columns [0, n0) go to out0 (row stride ld0), [n0, n0 + n1) to out1 (row stride ld1).
 *   ln_res != NULL: out0 = LayerNorm(ln_res + y) over the n0 == 16 * ceil(cout / 16) columns (ln_res contiguous [rows][n0]).
 *   packed: df3d_rows_linear_packed_bytes(cin, cout) bytes = [cin/32][ceil(cout/16)][hi|lo][lane 64][8 x bf16], lane (n, g) =
 *   column tile*16 + n, channels block*32 + g*8 .. +7 (W[cout][cin] row-major as nn.Linear keeps it; padding columns zero);
 *   dualfusion.ops.rows_linear_pack builds it.  0 bytes = shape not served. */
size_t df3d_rows_linear_packed_bytes(int cin, int cout);
int df3d_rows_linear(const float *x0, const float *x1, const float *x2, long long rows, int cin, const void *packed, int cout,
                     int csplit_cols, const float *bias, float *out0, int ld0, int n0, float *out1, int ld1, int n1,
                     const float *ln_res, const float *ln_gamma, const float *ln_beta, float eps, void *stream);

/* One pre-norm transformer encoder layer of the ACTRv2 LocalTransformer over groups of 32 tokens x 64 channels as ONE
 * kernel (csrc/ltlayer.hip): x1 = norm1(x); x2 = x1 + out_proj(MHA(x1)); x3 = norm2(x2); out = x3 + linear2(relu(linear1(x3)))
 * -- CP/det3d/models/model_utils/pointformer.py:10-44 (TransformerEncoderLayerPreNorm with nn.MultiheadAttention, 4 heads of
 * 16), applied to [L, G, C] sequence-first rows (row = token * G + group) as pointformer.py:349-380 does, or -- group_major
 * = 1 -- to [G, L, C] rows (a group's 32 tokens are one contiguous 8 KB block; in and out alike).  Replaces the chain
 * LayerNorm / in-projection / attention / out-projection / add + LayerNorm / feed-forward of eight launches.
 *   packed: df3d_lt_layer_packed_bytes() bytes = 64 fragment pairs [pair][hi | lo][lane 64][8 x bf16] of in_proj_weight
 *           [192, 64] (pairs 0..23: out tile * 2 + k-step), out_proj.weight (24..31), linear1.weight [128, 64] (32..47),
 *           linear2.weight [64, 128] (48..63: out tile * 4 + k-step); element j of lane (n, g) of the fragment (out tile ot,
 *           k-step s) = W[16 ot + n][16 (2 s + (j >> 2)) + 4 g + (j & 3)]; dualfusion.ops.lt_layer_pack builds it.
 *   vec:    df3d_lt_layer_vector_floats() floats = in_proj_bias 192 | out_proj.bias 64 | linear1.bias 128 | linear2.bias 64 |
 *           norm1.weight | norm1.bias | norm2.weight | norm2.bias (64 each).
 * Served: L = 32, C = 64, heads = 4, ffn = 128 (the ACTRv2 configuration); anything else is refused. */
long long df3d_lt_layer_packed_bytes(void);
int df3d_lt_layer_vector_floats(void);
int df3d_lt_layer(const float *x, int L, int G, int C, int heads, int ffn, int group_major, const void *packed, const float *vec,
                  float eps1, float eps2, float *out, void *stream);
/* The first and the last layer of a LocalTransformer chunk with the module's gather / scatter in their load / store
 * (pointformer.py:287-290,315-347,349-380; 32 tokens x 64 channels, 4 heads, feed-forward 128 as above):
 *   df3d_lt_layer_gather:  token (t, grp) = points[sel[t * G + grp]] + pe(gxyz[t * G + grp]) with the positional MLP
 *                          3 -> 32 (BatchNorm folded, ReLU) -> 64; `packed` = the layer's fragments followed by
 *                          df3d_lt_layer_pe_packed_bytes() more (pairs 64..67 = the MLP's second linear [64, 32], same
 *                          fragment format), `vec` = the layer's vector followed by df3d_lt_layer_pe_vector_floats() floats
 *                          (w0 [32][3] | b0 [32] | b1 [64]); out [32, G, 64] sequence-first rows.
 *   df3d_lt_layer_scatter: x [32, G, 64] sequence-first rows in; token (t, grp) writes its output row to
 *                          points[dst[t * G + grp]] when dst >= 0 (the 'unique' winner of its point, 'replace' aggregation),
 *                          nothing otherwise.  points may be the tensor df3d_lt_layer_gather read (stream order). */
long long df3d_lt_layer_pe_packed_bytes(void);
int df3d_lt_layer_pe_vector_floats(void);
int df3d_lt_layer_gather(const float *points, const long long *sel, const float *gxyz, int G, const void *packed,
                         const float *vec, float eps1, float eps2, float *out, void *stream);
int df3d_lt_layer_scatter(const float *x, int G, const void *packed, const float *vec, float eps1, float eps2,
                          const long long *dst, float *points, void *stream);

/* Point fusion of the Voxel-RCNN tree in one launch (csrc/mvx.hip): voxel (b, z, y, x) -> LiDAR corner
 * ((index * voxel_stride) * voxel size + range minimum) -> the point the camera saw (per sample `aug` [B, 5] = global scale,
 * cos(-rot), sin(-rot), flip_x sign on y, flip_y sign on x; identity = 1, 1, 0, 1, 1) -> pixel through lidar2img [B, 3, 4]
 * (u, v = rows 0, 1 over row 2) -> truncated pixel -> the value torch's bilinear upsample (align_corners = False) of `fmap`
 * [B, C, Hin, Win] to the image size [img_h, img_w] has at that pixel, zero outside the image
 * (VR/pcdet/models/backbones_3d/spconv_backbone.py:682-756, 760-814).  out[row] = (add[i] +) feature, row = out_rows[i] or
 * i; uv [n, 2] (optional) = the float pixel coordinates; grid [rows, 2] (optional) = (u / img_w, v / img_h).
 * voxel_size_zyx / range_min_zyx: HOST arrays of 3 floats; scale_y / scale_x = float32(Hin) / float32(img_h), float32(Win) /
 * float32(img_w) as torch computes them.  C % 4 == 0.  Same operations in the same order as the torch composition. */
int df3d_voxel_image_sample(const int32_t *indices, int n, int batch, float voxel_stride, const float *voxel_size_zyx,
                            const float *range_min_zyx, const float *aug, const float *lidar2img, const float *fmap, int C,
                            int Hin, int Win, int img_h, int img_w, float scale_y, float scale_x, const float *add,
                            const long long *out_rows, float *out, float *uv, float *grid, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DF3D_HIP_H_ */
