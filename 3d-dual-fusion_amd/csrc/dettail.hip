// Detection tail, part 2: CenterHead.predict + post_processing on the device.
//
// Reference: CP/det3d/models/bbox_heads/center_head.py:302-501 -- per task: permute the head maps to NHWC, sigmoid /
// exp / atan2, mesh-grid centres, concatenate boxes [B, H*W, 9]; then PER SAMPLE in Python: max over classes,
// score + range mask, boolean compaction, sort, top pre_max, rotate_nms_pcdet (bit matrix to the host, host
// reduction), index_select.  ~30 launches and 3 host round trips per (task, sample).
//
// Here the whole tail of all tasks and samples is five launches and no host round trip:
//   head_keys      one 64-bit key per (task, sample, pixel): [segment | 0x3F800000 - score bits | pixel]; pixels that
//                  fail the score / range mask get the all-ones key.  No atomics, no compaction pass.
//   top-k select   (topk.hip) -- the pre_max smallest keys of every (task, sample) segment: descending score, ties by pixel.
//   head_gather    segment bounds by binary search; the first pre_max candidates are decoded again from the maps
//                  (boxes [x, y, z, dx, dy, dz, vx, vy, rot]) together with their NMS boxes in pcdet's frame
//                  (box_torch_ops.py:255-257: dx <-> dy, heading -> -heading - pi/2).
//   nms mask + reduce   csrc/nms.hip, all segments at once.
//   head_select    the kept boxes / scores / labels, at most post_max per segment.
// Head maps are read as channels-last rows with a row stride, i.e. straight from the row kernels of the neck / head
// (NCHW maps go through one permute on the caller's side, as the reference does).
#include <cstring>

#include "common.h"

// the reference's float expressions, evaluated operation by operation (no fused multiply-adds)
#pragma clang fp contract(off)

namespace df3d {

struct TailArgs {
  df3d_head_task task[DF3D_MAX_HEAD_TASKS];
  int ntasks, batch, H, W;
  float osf, vs_x, vs_y, pc_x, pc_y;   // out_size_factor, voxel_size, pc_range
  float rng[6];
  float score_thr;
  int has_range;
};

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// score = max_c sigmoid(hm[c]) (first maximum wins like torch.max), label = its index
__device__ __forceinline__ void score_label(const float *hm, int ncls, float &score, int &label) {
  score = sigmoidf(hm[0]);
  label = 0;
  for (int c = 1; c < ncls; ++c) {
    const float v = sigmoidf(hm[c]);
    if (v > score) {
      score = v;
      label = c;
    }
  }
}

__global__ __launch_bounds__(256) void head_keys_kernel(TailArgs a, unsigned long long *__restrict__ keys) {
  const int hw = a.H * a.W;
  const long long per_task = (long long)a.batch * hw;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= per_task * a.ntasks) return;
  const int t = (int)(i / per_task);
  const long long row = i - (long long)t * per_task;
  const int b = (int)(row / hw), pix = (int)(row - (long long)b * hw);
  const df3d_head_task &k = a.task[t];
  float score;
  int label;
  score_label(k.hm + row * k.ld_hm, k.num_classes, score, label);
  bool ok = score > a.score_thr;
  if (ok && a.has_range) {
    const int y = pix / a.W, x = pix - y * a.W;
    const float xs = ((float)x + k.reg[row * k.ld_reg]) * a.osf * a.vs_x + a.pc_x;
    const float ys = ((float)y + k.reg[row * k.ld_reg + 1]) * a.osf * a.vs_y + a.pc_y;
    const float zs = k.height[row * k.ld_height];
    ok = xs >= a.rng[0] && ys >= a.rng[1] && zs >= a.rng[2] && xs <= a.rng[3] && ys <= a.rng[4] && zs <= a.rng[5];
  }
  unsigned long long key = ~0ull;
  if (ok) {
    const unsigned inv = 0x3F800000u - __float_as_uint(score);          // score in (0, 1]: descending score = ascending inv
    key = ((unsigned long long)(t * a.batch + b) << 56) | ((unsigned long long)inv << 24) | (unsigned)pix;
  }
  keys[i] = key;
}

// grid (ceil(pre_max / 256), segments)
__global__ __launch_bounds__(256) void head_gather_kernel(TailArgs a, const unsigned long long *__restrict__ keys,
                                                          int pre_max, float *__restrict__ cand_boxes,
                                                          float *__restrict__ cand_scores, int32_t *__restrict__ cand_labels,
                                                          float *__restrict__ nms_boxes, const int32_t *__restrict__ counts) {
  const int seg = blockIdx.y, r = blockIdx.x * 256 + threadIdx.x;
  const int n = counts[seg];                     // valid keys among the segment's pre_max best (written by the select)
  if (r >= n) return;
  const unsigned long long key = keys[(size_t)seg * pre_max + r];
  const int t = seg / a.batch, b = seg - t * a.batch;
  const int pix = (int)(key & 0xFFFFFFull);
  const float score = __uint_as_float(0x3F800000u - (unsigned)((key >> 24) & 0xFFFFFFFFull));
  const df3d_head_task &k = a.task[t];
  const long long row = (long long)b * a.H * a.W + pix;
  float sc;
  int label;
  score_label(k.hm + row * k.ld_hm, k.num_classes, sc, label);
  const int y = pix / a.W, x = pix - y * a.W;
  float box[9];
  box[0] = ((float)x + k.reg[row * k.ld_reg]) * a.osf * a.vs_x + a.pc_x;       // center_head.py:402-406
  box[1] = ((float)y + k.reg[row * k.ld_reg + 1]) * a.osf * a.vs_y + a.pc_y;
  box[2] = k.height[row * k.ld_height];
  box[3] = expf(k.dim[row * k.ld_dim]);
  box[4] = expf(k.dim[row * k.ld_dim + 1]);
  box[5] = expf(k.dim[row * k.ld_dim + 2]);
  box[6] = k.vel ? k.vel[row * k.ld_vel] : 0.f;
  box[7] = k.vel ? k.vel[row * k.ld_vel + 1] : 0.f;
  box[8] = atan2f(k.rot[row * k.ld_rot], k.rot[row * k.ld_rot + 1]);
  const size_t o = (size_t)seg * pre_max + r;
#pragma unroll
  for (int e = 0; e < 9; ++e) cand_boxes[o * 9 + e] = box[e];
  cand_scores[o] = score;
  cand_labels[o] = label + k.label_base;
  float *nb = nms_boxes + o * 7;
  nb[0] = box[0];
  nb[1] = box[1];
  nb[2] = box[2];
  nb[3] = box[4];
  nb[4] = box[3];
  nb[5] = box[5];
  nb[6] = -box[8] - 1.5707963267948966f;       // float(-theta - pi/2): the reference adds a Python double to a float tensor
}

// grid (segments); out_* [segments][post_max]
__global__ __launch_bounds__(128) void head_select_kernel(const float *__restrict__ cand_boxes,
                                                          const float *__restrict__ cand_scores,
                                                          const int32_t *__restrict__ cand_labels,
                                                          const int32_t *__restrict__ keep,
                                                          const int32_t *__restrict__ num_keep, int pre_max, int post_max,
                                                          int box_dim, int has_vel, float *__restrict__ out_boxes,
                                                          float *__restrict__ out_scores, int32_t *__restrict__ out_labels) {
  const int seg = blockIdx.x;
  const int n = min(num_keep[seg], post_max);
  for (int r = threadIdx.x; r < post_max; r += blockDim.x) {
    const size_t o = (size_t)seg * post_max + r;
    if (r < n) {
      const size_t c = (size_t)seg * pre_max + keep[(size_t)seg * pre_max + r];
      const float *bx = cand_boxes + c * 9;
      float *ob = out_boxes + o * box_dim;
      if (has_vel) {
        for (int e = 0; e < 9; ++e) ob[e] = bx[e];
      } else {
        for (int e = 0; e < 6; ++e) ob[e] = bx[e];
        ob[6] = bx[8];
      }
      out_scores[o] = cand_scores[c];
      out_labels[o] = cand_labels[c];
    } else {
      for (int e = 0; e < box_dim; ++e) out_boxes[o * box_dim + e] = 0.f;
      out_scores[o] = 0.f;
      out_labels[o] = -1;
    }
  }
}

struct TailPlan {
  size_t keys_in, keys_out, select, select_bytes, cand_boxes, cand_scores, cand_labels, nms_boxes, counts, keep, mask,
      mask_bytes, total;
};

static int tail_plan(int ntasks, int batch, int H, int W, int pre_max, TailPlan &p) {
  const size_t nkeys = (size_t)ntasks * batch * H * W, S = (size_t)ntasks * batch;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = align_up(off, 256);
    off = o + bytes;
    return o;
  };
  p.keys_in = take(nkeys * 8);
  p.keys_out = take(S * pre_max * 8);
  p.select_bytes = topk_keys_workspace((int)S, (long long)H * W, pre_max);
  p.select = take(p.select_bytes);
  p.cand_boxes = take(S * pre_max * 9 * 4);
  p.cand_scores = take(S * pre_max * 4);
  p.cand_labels = take(S * pre_max * 4);
  p.nms_boxes = take(S * pre_max * 7 * 4);
  p.counts = take(S * 4);
  p.keep = take(S * pre_max * 4);
  p.mask_bytes = df3d_nms_bev_workspace_bytes((int)S, pre_max);
  p.mask = take(p.mask_bytes);
  p.total = align_up(off, 256);
  return DF3D_OK;
}

}  // namespace df3d

using namespace df3d;

static int check_tail(const df3d_head_task *tasks, int ntasks, const df3d_head_decode_cfg *cfg) {
  DF3D_CHECK_ARG(tasks && cfg, "centerhead_predict: null argument");
  DF3D_CHECK_ARG(ntasks > 0 && ntasks <= DF3D_MAX_HEAD_TASKS, "centerhead_predict: 1..%d tasks", DF3D_MAX_HEAD_TASKS);
  DF3D_CHECK_ARG(cfg->batch > 0 && cfg->H > 0 && cfg->W > 0, "centerhead_predict: bad map size");
  DF3D_CHECK_ARG((long long)cfg->H * cfg->W < (1 << 24), "centerhead_predict: at most 2^24 pixels per map");
  DF3D_CHECK_ARG(ntasks * cfg->batch <= 255, "centerhead_predict: at most 255 (task, sample) segments");
  DF3D_CHECK_ARG(cfg->pre_max > 0 && cfg->pre_max <= 4096 && cfg->post_max > 0,
                 "centerhead_predict: 0 < pre_max <= 4096, post_max > 0");
  return DF3D_OK;
}

extern "C" size_t df3d_centerhead_predict_workspace_bytes(int ntasks, const df3d_head_decode_cfg *cfg) {
  if (!cfg || ntasks <= 0 || cfg->batch <= 0 || cfg->H <= 0 || cfg->W <= 0 || cfg->pre_max <= 0) return 0;
  TailPlan p;
  if (tail_plan(ntasks, cfg->batch, cfg->H, cfg->W, cfg->pre_max, p)) return 0;
  return p.total;
}

extern "C" int df3d_centerhead_predict(const df3d_head_task *tasks, int ntasks, const df3d_head_decode_cfg *cfg,
                                       float *out_boxes, float *out_scores, int32_t *out_labels, int32_t *out_counts,
                                       void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_tail(tasks, ntasks, cfg);
  if (rc) return rc;
  DF3D_CHECK_ARG(out_boxes && out_scores && out_labels && out_counts && workspace, "centerhead_predict: null output");
  int has_vel = tasks[0].vel != nullptr;
  for (int t = 0; t < ntasks; ++t) {
    const df3d_head_task &k = tasks[t];
    DF3D_CHECK_ARG(k.hm && k.reg && k.height && k.dim && k.rot, "centerhead_predict: task %d lacks a head map", t);
    DF3D_CHECK_ARG((k.vel != nullptr) == (has_vel != 0), "centerhead_predict: 'vel' must be present in all tasks or none");
    DF3D_CHECK_ARG(k.num_classes > 0 && k.ld_hm >= k.num_classes && k.ld_reg >= 2 && k.ld_height >= 1 && k.ld_dim >= 3 &&
                       k.ld_rot >= 2 && (!k.vel || k.ld_vel >= 2),
                   "centerhead_predict: task %d has a row stride smaller than its channel count", t);
  }
  TailPlan p;
  rc = tail_plan(ntasks, cfg->batch, cfg->H, cfg->W, cfg->pre_max, p);
  if (rc) {
    set_error("centerhead_predict: rocPRIM temp-storage query failed");
    return rc;
  }
  DF3D_CHECK_ARG(workspace_bytes >= p.total, "centerhead_predict: workspace %zu < %zu bytes", workspace_bytes, p.total);
  TailArgs a;
  memset(&a, 0, sizeof(a));
  for (int t = 0; t < ntasks; ++t) a.task[t] = tasks[t];
  a.ntasks = ntasks;
  a.batch = cfg->batch;
  a.H = cfg->H;
  a.W = cfg->W;
  a.osf = cfg->out_size_factor;
  a.vs_x = cfg->voxel_size[0];
  a.vs_y = cfg->voxel_size[1];
  a.pc_x = cfg->pc_range[0];
  a.pc_y = cfg->pc_range[1];
  a.has_range = cfg->has_post_center_range;
  for (int e = 0; e < 6; ++e) a.rng[e] = cfg->post_center_range[e];
  a.score_thr = cfg->score_threshold;

  char *ws = (char *)workspace;
  const long long nkeys = (long long)ntasks * cfg->batch * cfg->H * cfg->W;
  const int S = ntasks * cfg->batch;
  unsigned long long *kin = (unsigned long long *)(ws + p.keys_in), *kout = (unsigned long long *)(ws + p.keys_out);
  hipLaunchKernelGGL(head_keys_kernel, dim3(cdiv(nkeys, 256)), dim3(256), 0, stream, a, kin);
  float *cb = (float *)(ws + p.cand_boxes), *cs = (float *)(ws + p.cand_scores), *nb = (float *)(ws + p.nms_boxes);
  int32_t *cl = (int32_t *)(ws + p.cand_labels), *cnt = (int32_t *)(ws + p.counts), *keep = (int32_t *)(ws + p.keep);
  rc = topk_keys(kin, S, (long long)cfg->H * cfg->W, cfg->pre_max, kout, cnt, ws + p.select, p.select_bytes, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(head_gather_kernel, dim3(cdiv(cfg->pre_max, 256), S), dim3(256), 0, stream, a, kout, cfg->pre_max, cb,
                     cs, cl, nb, cnt);
  DF3D_LAUNCH_CHECK();
  rc = df3d_nms_bev(nb, cnt, S, cfg->pre_max, cfg->nms_threshold, cfg->nms_mode, cfg->post_max, keep, out_counts,
                    ws + p.mask, p.mask_bytes, stream_);
  if (rc) return rc;
  hipLaunchKernelGGL(head_select_kernel, dim3(S), dim3(128), 0, stream, cb, cs, cl, keep, out_counts, cfg->pre_max,
                     cfg->post_max, has_vel ? 9 : 7, has_vel, out_boxes, out_scores, out_labels);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}
