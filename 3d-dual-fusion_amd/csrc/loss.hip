// Detection losses of the CenterPoint head on the device (the values the N > 1 step all-reduces).
//
// Reference: CP/det3d/models/bbox_heads/center_head.py:250-298 -- per task
//   hm       = clamp(sigmoid(hm), 1e-4, 1 - 1e-4)                                              (:246-248)
//   hm_loss  = FastFocalLoss(hm, target hm, ind, mask, cat)       (losses/centernet_loss.py:29-58)
//            = -(sum_pos log(p) (1-p)^2 + sum_all log(1-p) p^2 (1-gt)^4) / num_pos     (-neg alone when num_pos == 0)
//   box_loss = RegLoss(cat(reg, height, dim, vel, rot), mask, ind, anno_box)                     (:6-27)
//            = sum_{b,m} |pred mask - target mask| / (sum(mask) + 1e-4)          one value per box code
//   loc_loss = sum(box_loss * code_weights);  loss = hm_loss + weight * loc_loss
// as ~25 torch launches and two host round trips (`.detach().cpu()`) per task.
//
// Here: two launches for all tasks and samples, no host round trip.
//   loss_neg_kernel     grid (chunks, tasks): the dense negative term over every (sample, pixel, class), one fp64 partial
//                       per workgroup (fixed order -> run-to-run deterministic).
//   loss_finish_kernel  one workgroup per task: the <= batch * max_objs object slots (positive term, L1 codes), the
//                       partials of the first kernel, the final scalars.
// Head maps are read as channels-last pixel rows with a row stride (df3d_head_task), i.e. where the head's row kernels
// left them; heat-map targets come [B, C, H, W] as the reference's assigner writes them.
#include <string.h>

#include "common.h"

namespace df3d {

constexpr int LOSS_CHUNK = 1024;   // (sample, pixel) rows per workgroup of the dense pass

struct LossArgs {
  df3d_head_task task[DF3D_MAX_HEAD_TASKS];
  df3d_head_task gtask[DF3D_MAX_HEAD_TASKS];    // gradient maps (same layout as `task`), GRAD kernels only
  df3d_head_targets target[DF3D_MAX_HEAD_TASKS];
  float code_weights[DF3D_LOSS_MAX_CODES];
  int ntasks, batch, hw, max_objs, box_dim, ncodes, chunks;
  float weight;
};

__device__ __forceinline__ float clamped_sigmoid(float x) {
  const float s = 1.f / (1.f + expf(-x));
  return fminf(fmaxf(s, 1e-4f), 1.f - 1e-4f);
}

// sum over the 256 threads of a workgroup; result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double *s_red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();                                   // s_red may still be read from a previous call
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// clamp(sigmoid(x), 1e-4, 1 - 1e-4) and d/dx of it (0 where the clamp is active, like torch.clamp's backward)
__device__ __forceinline__ float clamped_sigmoid_grad(float x, float &dpdx) {
  const float s = 1.f / (1.f + expf(-x));
  const bool inside = s >= 1e-4f && s <= 1.f - 1e-4f;
  const float p = fminf(fmaxf(s, 1e-4f), 1.f - 1e-4f);
  dpdx = inside ? p * (1.f - p) : 0.f;
  return p;
}

// number of positives of task t (sum of the mask), by every thread of the workgroup
__device__ __forceinline__ double task_positives(const LossArgs &a, int t, double *s_red) {
  double n = 0.0;
  const int slots = a.batch * a.max_objs;
  for (int s = threadIdx.x; s < slots; s += 256) n += a.target[t].mask[s] ? 1.0 : 0.0;
  block_sum(n, s_red);
  __syncthreads();
  return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// GRAD: also d(sum of the tasks' losses) / d(heat-map logits) of the dense negative term, written (not added) to the
// gradient map: the training path (`CenterHeadLossFunction`) gets loss values AND the gradient of every head map from the
// two launches.
template <bool GRAD>
__global__ __launch_bounds__(256) void loss_neg_kernel(LossArgs a, double *__restrict__ partial) {
  __shared__ double s_red[4];
  const int t = blockIdx.y;
  const df3d_head_task &k = a.task[t];
  const float *__restrict__ gt = a.target[t].hm;
  const long long rows = (long long)a.batch * a.hw;
  const long long r0 = (long long)blockIdx.x * LOSS_CHUNK;
  float inv = 0.f;
  if (GRAD) {
    const double npos = task_positives(a, t, s_red);
    inv = (float)(1.0 / (npos > 0.0 ? npos : 1.0));
  }
  double acc = 0.0;
  for (int i = threadIdx.x; i < LOSS_CHUNK; i += 256) {
    const long long r = r0 + i;
    if (r >= rows) break;
    const int b = (int)(r / a.hw), pix = (int)(r - (long long)b * a.hw);
    const float *hm = k.hm + r * k.ld_hm;
    float s = 0.f;
    for (int c = 0; c < k.num_classes; ++c) {
      float dpdx = 0.f;
      const float p = GRAD ? clamped_sigmoid_grad(hm[c], dpdx) : clamped_sigmoid(hm[c]);
      const float g = 1.f - gt[((long long)b * k.num_classes + c) * a.hw + pix];
      const float g2 = g * g;
      const float l1p = logf(1.f - p);
      s += l1p * (p * p) * (g2 * g2);
      if (GRAD) {                                    // d[log(1-p) p^2]/dp = 2 p log(1-p) - p^2 / (1-p); loss = -(.)/npos
        const float dterm = (g2 * g2) * (2.f * p * l1p - (p * p) / (1.f - p));
        const_cast<float *>(a.gtask[t].hm)[r * a.gtask[t].ld_hm + c] = -inv * dterm * dpdx;
      }
    }
    acc += (double)s;
  }
  const double tot = block_sum(acc, s_red);
  if (threadIdx.x == 0) partial[(size_t)t * a.chunks + blockIdx.x] = tot;
}

template <bool GRAD>
__global__ __launch_bounds__(256) void loss_finish_kernel(LossArgs a, const double *__restrict__ partial,
                                                          float *__restrict__ out) {
  __shared__ double s_red[4];
  const int t = blockIdx.x;
  const df3d_head_task &gk = a.gtask[t];
  float inv_pos = 0.f, inv_l1 = 0.f;
  if (GRAD) {
    const double np = task_positives(a, t, s_red);
    inv_pos = (float)(1.0 / (np > 0.0 ? np : 1.0));
    inv_l1 = (float)(1.0 / (np + 1e-4));
  }
  const df3d_head_task &k = a.task[t];
  const df3d_head_targets &g = a.target[t];
  const int slots = a.batch * a.max_objs;
  double pos = 0.0, npos = 0.0, neg = 0.0, elem[DF3D_LOSS_MAX_CODES];
#pragma unroll
  for (int j = 0; j < DF3D_LOSS_MAX_CODES; ++j) elem[j] = 0.0;
  const bool vel = k.vel != nullptr;
  for (int s = threadIdx.x; s < slots; s += 256) {
    const float m = g.mask[s] ? 1.f : 0.f;
    const long long ind = g.ind[s];
    if (ind < 0 || ind >= a.hw) {                      // the reference's gather would fault; the slot contributes no
      npos += m;                                       // terms, but it counts as a positive like mask.sum() does -- the
      continue;                                        // same normaliser as the gradient kernels' task_positives()
    }
    const int b = s / a.max_objs;
    const long long r = (long long)b * a.hw + ind;
    const long long c = g.cat[s];
    if (m != 0.f && c >= 0 && c < k.num_classes) {
      float dpdx = 0.f;
      const float p = GRAD ? clamped_sigmoid_grad(k.hm[r * k.ld_hm + c], dpdx) : clamped_sigmoid(k.hm[r * k.ld_hm + c]);
      const float q = 1.f - p;
      const float lp = logf(p);
      pos += (double)(lp * (q * q));
      if (GRAD)                                      // d[log(p) (1-p)^2]/dp = (1-p)^2 / p - 2 (1-p) log(p)
        unsafeAtomicAdd(const_cast<float *>(gk.hm) + r * gk.ld_hm + c, -inv_pos * ((q * q) / p - 2.f * q * lp) * dpdx);
    }
    npos += m;
    // codes in the reference's concatenation order: reg(2) height(1) dim(3) [vel(2)] rot(2); without vel the
    // target keeps columns [0..5, -2, -1] of anno_box (center_head.py:229)
    float pred[DF3D_LOSS_MAX_CODES];
    pred[0] = k.reg[r * k.ld_reg], pred[1] = k.reg[r * k.ld_reg + 1];
    pred[2] = k.height[r * k.ld_height];
    pred[3] = k.dim[r * k.ld_dim], pred[4] = k.dim[r * k.ld_dim + 1], pred[5] = k.dim[r * k.ld_dim + 2];
    int n = 6;
    if (vel) pred[n++] = k.vel[r * k.ld_vel], pred[n++] = k.vel[r * k.ld_vel + 1];
    pred[n++] = k.rot[r * k.ld_rot], pred[n++] = k.rot[r * k.ld_rot + 1];
    const float *tb = g.box + (size_t)s * a.box_dim;
    float dl[DF3D_LOSS_MAX_CODES];
    for (int j = 0; j < a.ncodes; ++j) {
      const int col = (vel || j < 6) ? j : a.box_dim - 2 + (j - 6);
      const float d = pred[j] * m - tb[col] * m;
      elem[j] += (double)fabsf(d);
      dl[j] = a.weight * a.code_weights[j] * m * (float)((d > 0.f) - (d < 0.f)) * inv_l1;
    }
    if (GRAD && m != 0.f) {                            // the L1 term: loss += weight * sum_j cw_j |.| / (npos + 1e-4)
      float *gr = const_cast<float *>(gk.reg) + r * gk.ld_reg;
      unsafeAtomicAdd(gr, dl[0]), unsafeAtomicAdd(gr + 1, dl[1]);
      unsafeAtomicAdd(const_cast<float *>(gk.height) + r * gk.ld_height, dl[2]);
      float *gd = const_cast<float *>(gk.dim) + r * gk.ld_dim;
      unsafeAtomicAdd(gd, dl[3]), unsafeAtomicAdd(gd + 1, dl[4]), unsafeAtomicAdd(gd + 2, dl[5]);
      int q = 6;
      if (vel) {
        float *gv = const_cast<float *>(gk.vel) + r * gk.ld_vel;
        unsafeAtomicAdd(gv, dl[6]), unsafeAtomicAdd(gv + 1, dl[7]);
        q = 8;
      }
      float *go = const_cast<float *>(gk.rot) + r * gk.ld_rot;
      unsafeAtomicAdd(go, dl[q]), unsafeAtomicAdd(go + 1, dl[q + 1]);
    }
  }
  for (int i = threadIdx.x; i < a.chunks; i += 256) neg += partial[(size_t)t * a.chunks + i];
  pos = block_sum(pos, s_red);
  npos = block_sum(npos, s_red);
  neg = block_sum(neg, s_red);
  for (int j = 0; j < a.ncodes; ++j) elem[j] = block_sum(elem[j], s_red);
  if (threadIdx.x != 0) return;
  float *o = out + (size_t)t * DF3D_LOSS_FIELDS;
  const float hm_loss = npos == 0.0 ? (float)(-neg) : (float)(-(pos + neg) / npos);
  double loc = 0.0;
  for (int j = 0; j < DF3D_LOSS_MAX_CODES; ++j) {
    const float e = j < a.ncodes ? (float)(elem[j] / (npos + 1e-4)) : 0.f;
    o[4 + j] = e;
    if (j < a.ncodes) loc += (double)e * a.code_weights[j];
  }
  o[0] = hm_loss + a.weight * (float)loc;
  o[1] = hm_loss;
  o[2] = (float)loc;
  o[3] = (float)npos;
}

}  // namespace df3d

using namespace df3d;

extern "C" {

size_t df3d_centerhead_loss_workspace_bytes(int ntasks, int batch, int H, int W) {
  if (ntasks <= 0 || batch <= 0 || H <= 0 || W <= 0) return 0;
  const long long rows = (long long)batch * H * W;
  return (size_t)ntasks * (size_t)cdiv(rows, LOSS_CHUNK) * sizeof(double) + 256;
}

static int centerhead_loss_impl(const df3d_head_task *tasks, const df3d_head_task *grad_tasks, const df3d_head_targets *targets,
                                int ntasks, int batch, int H, int W, int max_objs, int box_dim, const float *code_weights,
                                int ncodes, float weight, float *out, void *workspace, size_t workspace_bytes, void *stream) {
  DF3D_CHECK_ARG(tasks && targets && out, "df3d_centerhead_loss: null argument");
  DF3D_CHECK_ARG(ntasks >= 1 && ntasks <= DF3D_MAX_HEAD_TASKS, "df3d_centerhead_loss: 1..%d tasks", DF3D_MAX_HEAD_TASKS);
  DF3D_CHECK_ARG(batch >= 1 && H >= 1 && W >= 1 && max_objs >= 0, "df3d_centerhead_loss: bad sizes");
  const bool vel = tasks[0].vel != nullptr;
  DF3D_CHECK_ARG(ncodes == (vel ? 10 : 8) && ncodes <= DF3D_LOSS_MAX_CODES,
                 "df3d_centerhead_loss: %d code weights for %s boxes", ncodes, vel ? "10-code (vel)" : "8-code");
  DF3D_CHECK_ARG(box_dim >= ncodes, "df3d_centerhead_loss: anno_box has %d columns, %d codes", box_dim, ncodes);
  DF3D_CHECK_ARG(code_weights, "df3d_centerhead_loss: code_weights (host array) missing");
  LossArgs a;
  memset(&a, 0, sizeof(a));
  for (int t = 0; t < ntasks; ++t) {
    a.task[t] = tasks[t];
    a.target[t] = targets[t];
    DF3D_CHECK_ARG(tasks[t].hm && tasks[t].reg && tasks[t].height && tasks[t].dim && tasks[t].rot,
                   "df3d_centerhead_loss: task %d misses a head map", t);
    DF3D_CHECK_ARG((tasks[t].vel != nullptr) == vel, "df3d_centerhead_loss: tasks disagree about the vel head");
    DF3D_CHECK_ARG(targets[t].hm && (max_objs == 0 || (targets[t].ind && targets[t].mask && targets[t].cat && targets[t].box)),
                   "df3d_centerhead_loss: task %d misses a target", t);
    DF3D_CHECK_ARG(tasks[t].num_classes >= 1, "df3d_centerhead_loss: task %d has no classes", t);
    if (grad_tasks) {
      a.gtask[t] = grad_tasks[t];
      DF3D_CHECK_ARG(grad_tasks[t].hm && grad_tasks[t].reg && grad_tasks[t].height && grad_tasks[t].dim && grad_tasks[t].rot &&
                         (grad_tasks[t].vel != nullptr) == vel,
                     "df3d_centerhead_loss_grad: task %d misses a gradient map", t);
    }
  }
  for (int j = 0; j < DF3D_LOSS_MAX_CODES; ++j) a.code_weights[j] = j < ncodes ? code_weights[j] : 0.f;
  a.ntasks = ntasks, a.batch = batch, a.hw = H * W, a.max_objs = max_objs, a.box_dim = box_dim, a.ncodes = ncodes;
  a.chunks = cdiv((long long)batch * H * W, LOSS_CHUNK);
  a.weight = weight;
  DF3D_CHECK_ARG(workspace && workspace_bytes >= df3d_centerhead_loss_workspace_bytes(ntasks, batch, H, W),
                 "df3d_centerhead_loss: workspace too small");
  double *partial = (double *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  hipStream_t st = (hipStream_t)stream;
  if (grad_tasks) {
    loss_neg_kernel<true><<<dim3(a.chunks, ntasks), 256, 0, st>>>(a, partial);
    loss_finish_kernel<true><<<ntasks, 256, 0, st>>>(a, partial, out);
  } else {
    loss_neg_kernel<false><<<dim3(a.chunks, ntasks), 256, 0, st>>>(a, partial);
    loss_finish_kernel<false><<<ntasks, 256, 0, st>>>(a, partial, out);
  }
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

int df3d_centerhead_loss(const df3d_head_task *tasks, const df3d_head_targets *targets, int ntasks, int batch, int H, int W,
                         int max_objs, int box_dim, const float *code_weights, int ncodes, float weight, float *out,
                         void *workspace, size_t workspace_bytes, void *stream) {
  return centerhead_loss_impl(tasks, nullptr, targets, ntasks, batch, H, W, max_objs, box_dim, code_weights, ncodes, weight, out,
                              workspace, workspace_bytes, stream);
}

int df3d_centerhead_loss_grad(const df3d_head_task *tasks, const df3d_head_task *grad_tasks, const df3d_head_targets *targets,
                              int ntasks, int batch, int H, int W, int max_objs, int box_dim, const float *code_weights,
                              int ncodes, float weight, float *out, void *workspace, size_t workspace_bytes, void *stream) {
  DF3D_CHECK_ARG(grad_tasks, "df3d_centerhead_loss_grad: null gradient maps");
  return centerhead_loss_impl(tasks, grad_tasks, targets, ntasks, batch, H, W, max_objs, box_dim, code_weights, ncodes, weight,
                              out, workspace, workspace_bytes, stream);
}

}  // extern "C"
