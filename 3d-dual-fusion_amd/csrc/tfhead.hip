// Detection tail, part 3: the index / decode steps of TransFusionHead (LiDAR-only branch).
//
// Reference: TF/mmdet3d/models/dense_heads/transfusion_head.py:843-878 -- sigmoid of the dense heat map, a zero map
// with the 3x3 max-pool written into its interior, two class planes overwritten (pedestrian / traffic cone keep every
// pixel), heatmap * (heatmap == local_max), a full argsort of [C*H*W] per sample for the first num_proposals entries,
// then five gathers / a one-hot / a Conv1d to build the query features and positions (~25 launches, 4 full-map
// temporaries); :1285-1312 + core/bbox/coders/transfusion_bbox_coder.py:41-128 -- score product, max over the
// one-hot classes, box decode, range / score mask, boolean compaction per sample (~40 launches).
//
// Here:
//   proposal_keys     one 64-bit key per (sample, class, pixel): [sample | 0x3F800000 - score bits | class*HW + pixel]
//                     where score = sigmoid(logit) if it is the maximum of its window (interior pixels only) or its
//                     class is exempt, else 0.  Reads the heat map as channels-last rows straight from the conv kernel.
//   top-k select      (topk.hip): the K smallest keys per sample = descending score, equal scores by ascending flat
//                     index (= a stable descending argsort; the reference's argsort leaves the order of equal scores open).
//   proposal_gather   class / pixel / position of the first K keys, the suppressed scores of ALL classes at those
//                     pixels (query_heatmap_score) and the query feature rows  feat[pixel] + W_cls[:, class] + b_cls.
//   decode            one workgroup per sample: score, label, box, masks and the order-preserving compaction.
#include <cstring>

#include "common.h"

#pragma clang fp contract(off)

namespace df3d {

__device__ __forceinline__ float tf_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

struct ProposalArgs {
  const float *heat;       // [B*H*W, ld] logits
  int ld, batch, C, H, W, pad;
  unsigned exempt;         // bit c: class c keeps every pixel
};

// sigmoid(logit) if (b, c, pix) survives the local-maximum test, else 0
__device__ __forceinline__ float suppressed_score(const ProposalArgs &a, int b, int c, int y, int x) {
  const long long row = ((long long)b * a.H + y) * a.W + x;
  const float s = tf_sigmoid(a.heat[row * a.ld + c]);
  if ((a.exempt >> c) & 1u) return s;
  if (y < a.pad || y >= a.H - a.pad || x < a.pad || x >= a.W - a.pad) return 0.f;
  for (int dy = -a.pad; dy <= a.pad; ++dy)
    for (int dx = -a.pad; dx <= a.pad; ++dx) {
      if (dy == 0 && dx == 0) continue;
      const float v = tf_sigmoid(a.heat[(row + (long long)dy * a.W + dx) * a.ld + c]);
      if (v > s) return 0.f;
    }
  return s;
}

__global__ __launch_bounds__(256) void proposal_keys_kernel(ProposalArgs a, unsigned long long *__restrict__ keys) {
  const int hw = a.H * a.W;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;       // (b, pix, c), c fastest like the rows
  if (i >= (long long)a.batch * hw * a.C) return;
  const int c = (int)(i % a.C);
  const long long row = i / a.C;
  const int b = (int)(row / hw), pix = (int)(row - (long long)b * hw);
  const int y = pix / a.W, x = pix - y * a.W;
  const float s = suppressed_score(a, b, c, y, x);
  const unsigned inv = 0x3F800000u - __float_as_uint(s);               // s in [0, 1]
  keys[i] = ((unsigned long long)b << 56) | ((unsigned long long)inv << 24) | (unsigned)(c * hw + pix);
}

// grid (ceil(K / 16), B), 256 threads: 16 proposals per workgroup -- thread (q, c) evaluates class c at proposal q's
// pixel, then the 256 threads copy the 16 feature rows
__global__ __launch_bounds__(256) void proposal_gather_kernel(ProposalArgs a, const unsigned long long *__restrict__ keys,
                                                              int K, const float *__restrict__ feat, int ld_feat,
                                                              int channels, const float *__restrict__ cls_w,
                                                              const float *__restrict__ cls_b,
                                                              int32_t *__restrict__ top_class, int32_t *__restrict__ top_pixel,
                                                              float *__restrict__ query_score, float *__restrict__ query_pos,
                                                              float *__restrict__ query_feat) {
  const int b = blockIdx.y, r0 = blockIdx.x * 16, hw = a.H * a.W;
  const unsigned long long *seg = keys + (size_t)b * K;                  // the K best keys of sample b, ascending
  __shared__ int s_cls[16], s_pix[16];
  const int n = min(16, K - r0);
  if (threadIdx.x < n) {
    const int r = r0 + threadIdx.x;
    const unsigned idx = (unsigned)(seg[r] & 0xFFFFFFu);
    const int c = idx / hw, pix = idx - c * hw;
    s_cls[threadIdx.x] = c;
    s_pix[threadIdx.x] = pix;
    top_class[(size_t)b * K + r] = c;
    top_pixel[(size_t)b * K + r] = pix;
    if (query_pos) {
      const int y = pix / a.W, x = pix - y * a.W;
      query_pos[((size_t)b * K + r) * 2] = (float)x + 0.5f;
      query_pos[((size_t)b * K + r) * 2 + 1] = (float)y + 0.5f;
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < n * a.C; e += 256) {
    const int q = e / a.C, cc = e - q * a.C;
    const int y = s_pix[q] / a.W, x = s_pix[q] - y * a.W;
    query_score[((size_t)b * a.C + cc) * K + r0 + q] = suppressed_score(a, b, cc, y, x);
  }
  if (query_feat) {
    for (int e = threadIdx.x; e < n * channels; e += 256) {
      const int q = e / channels, ch = e - q * channels;
      float v = feat[((size_t)b * hw + s_pix[q]) * ld_feat + ch];
      if (cls_w) v += cls_w[(size_t)ch * a.C + s_cls[q]];
      if (cls_b) v += cls_b[ch];
      query_feat[((size_t)b * K + r0 + q) * channels + ch] = v;
    }
  }
}

struct DecodeArgs {
  const float *heat, *center, *height, *dim, *rot, *vel;   // rows [B*K, ld_*]
  int ld_heat, ld_center, ld_height, ld_dim, ld_rot, ld_vel;
  const float *query_score;                                // [B, C, K]
  const int32_t *query_label;                              // [B, K]
  int batch, K, C;
  float osf, vs_x, vs_y, pc_x, pc_y, rng[6], score_thr;
  int has_range, has_thr;
};

__global__ __launch_bounds__(256) void tf_decode_kernel(DecodeArgs a, float *__restrict__ boxes, float *__restrict__ scores,
                                                        int32_t *__restrict__ labels, int32_t *__restrict__ counts) {
  const int b = blockIdx.x, nd = a.vel ? 9 : 7;
  __shared__ int s_wave[4], s_base;
  if (threadIdx.x == 0) s_base = 0;
  for (int q0 = 0; q0 < a.K; q0 += 256) {
    const int q = q0 + threadIdx.x;
    bool ok = false;
    float box[9], score = 0.f;
    int label = 0;
    if (q < a.K) {
      const size_t row = (size_t)b * a.K + q;
      const int ql = a.query_label[row];
      // sigmoid(heatmap) * query_heatmap_score * one_hot, max over classes (first maximum): only class ql is non-zero
      score = tf_sigmoid(a.heat[row * a.ld_heat + ql]) * a.query_score[((size_t)b * a.C + ql) * a.K + q];
      label = score > 0.f ? ql : 0;
      box[0] = a.center[row * a.ld_center] * a.osf * a.vs_x + a.pc_x;
      box[1] = a.center[row * a.ld_center + 1] * a.osf * a.vs_y + a.pc_y;
      box[3] = expf(a.dim[row * a.ld_dim]);
      box[4] = expf(a.dim[row * a.ld_dim + 1]);
      box[5] = expf(a.dim[row * a.ld_dim + 2]);
      box[2] = a.height[row * a.ld_height] - box[5] * 0.5f;
      box[6] = atan2f(a.rot[row * a.ld_rot], a.rot[row * a.ld_rot + 1]);
      box[7] = a.vel ? a.vel[row * a.ld_vel] : 0.f;
      box[8] = a.vel ? a.vel[row * a.ld_vel + 1] : 0.f;
      ok = true;
      if (a.has_range)
        ok = box[0] >= a.rng[0] && box[1] >= a.rng[1] && box[2] >= a.rng[2] && box[0] <= a.rng[3] && box[1] <= a.rng[4] &&
             box[2] <= a.rng[5];
      if (a.has_thr) ok = ok && score > a.score_thr;
    }
    const unsigned long long m = __ballot(ok);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int pos = s_base + __popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) pos += s_wave[w];
    if (ok) {
      float *o = boxes + ((size_t)b * a.K + pos) * nd;
      for (int e = 0; e < nd; ++e) o[e] = box[e];
      scores[(size_t)b * a.K + pos] = score;
      labels[(size_t)b * a.K + pos] = label;
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
  }
  __syncthreads();
  if (threadIdx.x == 0) counts[b] = s_base;
}

struct TopkPlan {
  size_t keys, top, select, select_bytes, total;
};

constexpr int TF_MAX_PROPOSALS = 4096;

static int topk_plan(int batch, int C, int H, int W, TopkPlan &p) {
  const size_t n = (size_t)C * H * W;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = align_up(off, 256);
    off = o + bytes;
    return o;
  };
  p.keys = take((size_t)batch * n * 8);
  p.top = take((size_t)batch * TF_MAX_PROPOSALS * 8);
  p.select_bytes = topk_keys_workspace(batch, (long long)n, TF_MAX_PROPOSALS);
  p.select = take(p.select_bytes);
  p.total = align_up(off, 256);
  return p.select_bytes ? DF3D_OK : DF3D_EINVAL;
}

}  // namespace df3d

using namespace df3d;

static bool topk_sizes_ok(int batch, int C, int H, int W) {
  return batch > 0 && batch <= 255 && C > 0 && C <= 32 && H > 0 && W > 0 && (long long)C * H * W < (1 << 24);
}

extern "C" size_t df3d_heatmap_proposals_workspace_bytes(int batch, int num_classes, int H, int W) {
  if (!topk_sizes_ok(batch, num_classes, H, W)) return 0;
  TopkPlan p;
  if (topk_plan(batch, num_classes, H, W, p)) return 0;
  return p.total;
}

extern "C" int df3d_heatmap_proposals(const float *heat_rows, int ld_heat, int batch, int num_classes, int H, int W,
                                      int nms_kernel_size, unsigned exempt_classes, int num_proposals,
                                      const float *feat_rows, int ld_feat, int channels, const float *class_weight,
                                      const float *class_bias, int32_t *top_class, int32_t *top_pixel, float *query_score,
                                      float *query_pos, float *query_feat, void *workspace, size_t workspace_bytes,
                                      void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(topk_sizes_ok(batch, num_classes, H, W),
                 "heatmap_proposals: need 1..255 samples, 1..32 classes and fewer than 2^24 (class, pixel) pairs");
  DF3D_CHECK_ARG(nms_kernel_size >= 3 && (nms_kernel_size & 1) && nms_kernel_size <= 2 * (H < W ? H : W),
                 "heatmap_proposals: nms_kernel_size must be odd and >= 3 (got %d)", nms_kernel_size);
  DF3D_CHECK_ARG(num_proposals > 0 && num_proposals <= TF_MAX_PROPOSALS &&
                     (long long)num_proposals <= (long long)num_classes * H * W,
                 "heatmap_proposals: num_proposals %d out of range (1..min(%d, C*H*W))", num_proposals, TF_MAX_PROPOSALS);
  DF3D_CHECK_ARG(heat_rows && ld_heat >= num_classes, "heatmap_proposals: heat map rows");
  DF3D_CHECK_ARG(top_class && top_pixel && query_score && workspace, "heatmap_proposals: null output");
  DF3D_CHECK_ARG(!query_feat || (feat_rows && channels > 0 && ld_feat >= channels), "heatmap_proposals: feature rows");
  TopkPlan p;
  DF3D_CHECK_ARG(topk_plan(batch, num_classes, H, W, p) == DF3D_OK, "heatmap_proposals: unsupported map size");
  DF3D_CHECK_ARG(workspace_bytes >= p.total, "heatmap_proposals: workspace %zu < %zu bytes", workspace_bytes, p.total);
  ProposalArgs a = {heat_rows, ld_heat, batch, num_classes, H, W, nms_kernel_size / 2, exempt_classes};
  char *ws = (char *)workspace;
  const long long per_sample = (long long)num_classes * H * W, nkeys = (long long)batch * per_sample;
  unsigned long long *kin = (unsigned long long *)(ws + p.keys), *kout = (unsigned long long *)(ws + p.top);
  hipLaunchKernelGGL(proposal_keys_kernel, dim3(cdiv(nkeys, 256)), dim3(256), 0, stream, a, kin);
  int rc = topk_keys(kin, batch, per_sample, num_proposals, kout, nullptr, ws + p.select, p.select_bytes, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(proposal_gather_kernel, dim3(cdiv(num_proposals, 16), batch), dim3(256), 0, stream, a, kout, num_proposals, feat_rows, ld_feat,
                     channels, class_weight, class_bias, top_class, top_pixel, query_score, query_pos, query_feat);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_transfusion_decode(const df3d_query_heads *heads, const float *query_score, const int32_t *query_label,
                                       int batch, int num_proposals, int num_classes, const df3d_head_decode_cfg *cfg,
                                       float *out_boxes, float *out_scores, int32_t *out_labels, int32_t *out_counts,
                                       void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(heads && cfg && query_score && query_label, "transfusion_decode: null argument");
  DF3D_CHECK_ARG(batch > 0 && num_proposals > 0 && num_classes > 0, "transfusion_decode: bad sizes");
  DF3D_CHECK_ARG(heads->heatmap && heads->center && heads->height && heads->dim && heads->rot,
                 "transfusion_decode: a head output is missing");
  DF3D_CHECK_ARG(heads->ld_heatmap >= num_classes && heads->ld_center >= 2 && heads->ld_height >= 1 && heads->ld_dim >= 3 &&
                     heads->ld_rot >= 2 && (!heads->vel || heads->ld_vel >= 2),
                 "transfusion_decode: a row stride is smaller than its channel count");
  DF3D_CHECK_ARG(out_boxes && out_scores && out_labels && out_counts, "transfusion_decode: null output");
  DecodeArgs a;
  memset(&a, 0, sizeof(a));
  a.heat = heads->heatmap;
  a.center = heads->center;
  a.height = heads->height;
  a.dim = heads->dim;
  a.rot = heads->rot;
  a.vel = heads->vel;
  a.ld_heat = heads->ld_heatmap;
  a.ld_center = heads->ld_center;
  a.ld_height = heads->ld_height;
  a.ld_dim = heads->ld_dim;
  a.ld_rot = heads->ld_rot;
  a.ld_vel = heads->ld_vel;
  a.query_score = query_score;
  a.query_label = query_label;
  a.batch = batch;
  a.K = num_proposals;
  a.C = num_classes;
  a.osf = cfg->out_size_factor;
  a.vs_x = cfg->voxel_size[0];
  a.vs_y = cfg->voxel_size[1];
  a.pc_x = cfg->pc_range[0];
  a.pc_y = cfg->pc_range[1];
  a.has_range = cfg->has_post_center_range;
  for (int e = 0; e < 6; ++e) a.rng[e] = cfg->post_center_range[e];
  a.score_thr = cfg->score_threshold;
  a.has_thr = cfg->score_threshold != 0.f;      // `if self.score_threshold:` (transfusion_bbox_coder.py:110)
  hipLaunchKernelGGL(tf_decode_kernel, dim3(batch), dim3(256), 0, stream, a, out_boxes, out_scores, out_labels, out_counts);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}
