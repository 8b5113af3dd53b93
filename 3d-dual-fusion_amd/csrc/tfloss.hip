// Target assignment costs and losses of the TransFusion head on the device (the values
// BASELINE configs[2] / configs[3] all-reduce).
//
// Reference: TF/mmdet3d/models/dense_heads/transfusion_head.py:1048-1283 (get_targets / get_targets_single / loss),
// TF/mmdet3d/core/bbox/assigners/hungarian_assigner.py:14-48,100-160 (matching costs), TF/mmdet3d/ops/iou3d/src/
// iou3d_kernel.cu:56-258 (rotated BEV overlap of xyxyr boxes), TF/mmdet3d/core/bbox/structures/base_box3d.py:352-438
// (3-D IoU), TF/mmdet3d/core/utils/gaussian.py:5-86 (dense heat-map targets), transfusion_bbox_coder.py:24-39 (encode),
// :41-77 (decode), and mmdetection 2.10.0's FocalLossCost / FocalLoss / L1Loss / GaussianFocalLoss.
// The reference runs this as a Python loop per sample and per ground-truth box (hundreds of launches, a host round
// trip per sample for the cost matrix and one `.item()` for the heat-map normaliser).
//
// Here, for a whole batch:
//   overlap_xyxyr_kernel     boxes_overlap_bev_gpu of the reference's `iou3d_cuda` module ([N,5] x [M,5] -> areas).
//   match_cost_kernel        decode of every proposal + classification / BEV-centre / 3-D IoU cost against every
//                            ground-truth box of its sample: cost [B, P, Gmax] and iou [B, P, Gmax] in one launch --
//                            the only thing that travels to the host, where scipy's linear_sum_assignment runs as in
//                            the reference.
//   splat_kernel             all Gaussians of all samples into the zeroed target heat map [B, C, H, W]
//                            (radius rule + flushed fp64 Gaussian + integer atomicMax; values are >= 0).
//   gfocal_kernel / finish   Gaussian focal loss of the clamped-sigmoid dense heat map, its normaliser (#pixels == 1)
//                            and its gradient, fp64 partial sums in fixed order.
//   query_loss_kernel        from the matching: labels / weights / encoded box targets on the fly, focal + L1 loss per
//                            decoder layer, num_pos, matched IoUs, and the gradient with respect to every prediction.
// Polygon arithmetic is float, operation for operation the reference kernel's (FMA contraction off).
#include <float.h>

#include "common.h"

#pragma clang fp contract(off)

namespace df3d {

struct TPt {
  float x, y;
};

__device__ __forceinline__ float t_cross3(TPt p1, TPt p2, TPt p0) {
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

__device__ __forceinline__ bool t_rect_cross(TPt p1, TPt p2, TPt q1, TPt q2) {
  return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
         fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

__device__ __forceinline__ bool t_intersection(TPt p1, TPt p0, TPt q1, TPt q0, TPt &ans) {
  if (!t_rect_cross(p0, p1, q0, q1)) return false;
  const float s1 = t_cross3(q0, p1, p0), s2 = t_cross3(p1, q1, p0);
  const float s3 = t_cross3(p0, q1, q0), s4 = t_cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
  const float s5 = t_cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > 1e-8f) {
    ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    const float D = a0 * b1 - a1 * b0;
    ans.x = (b0 * c1 - b1 * c0) / D;
    ans.y = (a1 * c0 - a0 * c1) / D;
  }
  return true;
}

// box = [x1, y1, x2, y2, angle]; the corner turned by -angle about the centre (iou3d_kernel.cu:106-114,139-168)
__device__ __forceinline__ void t_corners(const float *box, TPt *c) {
  const float cx = (box[0] + box[2]) / 2, cy = (box[1] + box[3]) / 2, ca = cosf(box[4]), sa = sinf(box[4]);
  const float px[4] = {box[0], box[2], box[2], box[0]}, py[4] = {box[1], box[1], box[3], box[3]};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    c[k].x = (px[k] - cx) * ca + (py[k] - cy) * sa + cx;
    c[k].y = -(px[k] - cx) * sa + (py[k] - cy) * ca + cy;
  }
  c[4] = c[0];
}

__device__ __forceinline__ bool t_in_box(const float *box, TPt p) {       // iou3d_kernel.cu:56-79
  const float margin = 1e-5f;
  const float cx = (box[0] + box[2]) / 2, cy = (box[1] + box[3]) / 2;
  const float ca = cosf(-box[4]), sa = sinf(-box[4]);
  const float rx = (p.x - cx) * ca + (p.y - cy) * sa + cx;
  const float ry = -(p.x - cx) * sa + (p.y - cy) * ca + cy;
  return rx > box[0] - margin && rx < box[2] + margin && ry > box[1] - margin && ry < box[3] + margin;
}

__device__ float t_box_overlap(const float *a, const float *b) {          // iou3d_kernel.cu:122-230
  TPt ca[5], cb[5], pts[16], ctr = {0.f, 0.f};
  int cnt = 0;
  t_corners(a, ca);
  t_corners(b, cb);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      TPt x;
      if (t_intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], x)) {
        pts[cnt] = x;
        ctr.x += x.x;
        ctr.y += x.y;
        cnt++;
      }
    }
  for (int k = 0; k < 4; ++k) {
    if (t_in_box(a, cb[k])) {
      ctr.x += cb[k].x;
      ctr.y += cb[k].y;
      pts[cnt++] = cb[k];
    }
    if (t_in_box(b, ca[k])) {
      ctr.x += ca[k].x;
      ctr.y += ca[k].y;
      pts[cnt++] = ca[k];
    }
  }
  if (cnt < 3) return 0.f;                 // the reference's fan over fewer than three vertices is empty as well
  ctr.x /= cnt;
  ctr.y /= cnt;
  float ang[16];
  for (int i = 0; i < cnt; ++i) ang[i] = atan2f(pts[i].y - ctr.y, pts[i].x - ctr.x);
  for (int j = 0; j < cnt - 1; ++j)        // the reference's bubble sort (same swaps: the keys do not change)
    for (int i = 0; i < cnt - j - 1; ++i)
      if (ang[i] > ang[i + 1]) {
        const TPt t = pts[i];
        pts[i] = pts[i + 1];
        pts[i + 1] = t;
        const float u = ang[i];
        ang[i] = ang[i + 1];
        ang[i + 1] = u;
      }
  float area = 0.f;
  for (int k = 0; k < cnt - 1; ++k) {
    const float ax = pts[k].x - pts[0].x, ay = pts[k].y - pts[0].y;
    const float bx = pts[k + 1].x - pts[0].x, by = pts[k + 1].y - pts[0].y;
    area += ax * by - ay * bx;
  }
  return fabsf(area) / 2.0f;
}

// overlap can only be non-zero when the centres are closer than the two half diagonals (+ margin)
__device__ __forceinline__ bool t_far_apart(const float *a, const float *b) {
  const float aw = a[2] - a[0], ah = a[3] - a[1], bw = b[2] - b[0], bh = b[3] - b[1];
  const float ra = 0.5f * sqrtf(aw * aw + ah * ah), rb = 0.5f * sqrtf(bw * bw + bh * bh);
  const float dx = (a[0] + a[2]) / 2 - (b[0] + b[2]) / 2, dy = (a[1] + a[3]) / 2 - (b[1] + b[3]) / 2;
  const float r = ra + rb + 0.01f;
  return dx * dx + dy * dy > r * r * 1.01f;
}

__global__ __launch_bounds__(256) void overlap_xyxyr_kernel(const float *__restrict__ A, int na, const float *__restrict__ B,
                                                            int nb, float *__restrict__ out) {
  const int j = blockIdx.x * 16 + (threadIdx.x & 15), i = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (i >= na || j >= nb) return;
  float a[5], b[5];
#pragma unroll
  for (int e = 0; e < 5; ++e) {
    a[e] = A[(size_t)i * 5 + e];
    b[e] = B[(size_t)j * 5 + e];
  }
  out[(size_t)i * nb + j] = t_far_apart(a, b) ? 0.f : t_box_overlap(a, b);
}

// ---------------------------------------------------------------------------------------------- matching costs
struct MatchArgs {
  const float *rows;        // [B, P, ld]: center 0:2, height 2, dim 3:6, rot 6:8, (vel 8:10,) class logits at col_cls
  const float *gt;          // [G_total, gt_dim] (x, y, z_bottom, w, l, h, yaw, ...)
  const int32_t *gt_labels; // [G_total]
  const int32_t *gt_off;    // [B + 1]
  float *cost, *iou;        // [B, P, gmax]
  float *boxes;             // [B, P, 7] decoded (x, y, z_bottom, w, l, h, yaw) or NULL
  int B, P, ld, col_cls, num_classes, gt_dim, gmax;
  float osf, vs0, vs1, pc0, pc1;          // decode
  float start0, start1, ext0, ext1;       // BBoxBEVL1Cost normalisation
  float w_cls, alpha, gamma, eps, w_reg, w_iou;
};

// torch.pow with exponent 2 is x * x exactly; other exponents go through powf
__device__ __forceinline__ float pow_e(float x, float e) { return e == 2.f ? x * x : (e == 1.f ? x : powf(x, e)); }

__device__ __forceinline__ void decode_box(const float *r, const MatchArgs &a, float *box) {
  box[0] = r[0] * a.osf * a.vs0 + a.pc0;
  box[1] = r[1] * a.osf * a.vs1 + a.pc1;
  box[3] = expf(r[3]);
  box[4] = expf(r[4]);
  box[5] = expf(r[5]);
  box[2] = r[2] - box[5] * 0.5f;
  box[6] = atan2f(r[6], r[7]);
}

__device__ __forceinline__ void to_xyxyr(const float *box, float *q) {
  const float hw = box[3] / 2, hl = box[4] / 2;
  q[0] = box[0] - hw;
  q[1] = box[1] - hl;
  q[2] = box[0] + hw;
  q[3] = box[1] + hl;
  q[4] = box[6];
}

__device__ __forceinline__ float iou3d_lidar(const float *p, const float *g) {
  float pq[5], gq[5];
  to_xyxyr(p, pq);
  to_xyxyr(g, gq);
  const float bev = t_far_apart(pq, gq) ? 0.f : t_box_overlap(pq, gq);
  const float oh = fmaxf(fminf(p[2] + p[5], g[2] + g[5]) - fmaxf(p[2], g[2]), 0.f);
  const float o3 = bev * oh;
  const float v1 = p[3] * p[4] * p[5], v2 = g[3] * g[4] * g[5];
  return o3 / fmaxf(v1 + v2 - o3, 1e-8f);
}

__global__ __launch_bounds__(256) void match_cost_kernel(MatchArgs a) {
  const int b = blockIdx.z, p = blockIdx.y * 16 + (threadIdx.x >> 4), gi = blockIdx.x * 16 + (threadIdx.x & 15);
  if (p >= a.P) return;
  const int g0 = a.gt_off[b], ng = a.gt_off[b + 1] - g0;
  const float *r = a.rows + ((size_t)b * a.P + p) * a.ld;
  float box[7];
  decode_box(r, a, box);
  if (a.boxes && gi == 0) {
#pragma unroll
    for (int e = 0; e < 7; ++e) a.boxes[((size_t)b * a.P + p) * 7 + e] = box[e];
  }
  if (gi >= a.gmax) return;
  const size_t o = ((size_t)b * a.P + p) * a.gmax + gi;
  if (gi >= ng) {
    a.cost[o] = 0.f;
    a.iou[o] = 0.f;
    return;
  }
  const float *g = a.gt + (size_t)(g0 + gi) * a.gt_dim;
  const int lab = a.gt_labels[g0 + gi];
  float cls = 0.f;
  if (lab >= 0 && lab < a.num_classes) {
    const float s = 1.f / (1.f + expf(-r[a.col_cls + lab]));
    const float neg = -logf(1.f - s + a.eps) * (1.f - a.alpha) * pow_e(s, a.gamma);
    const float pos = -logf(s + a.eps) * a.alpha * pow_e(1.f - s, a.gamma);
    cls = (pos - neg) * a.w_cls;
  }
  const float reg = (fabsf((box[0] - a.start0) / a.ext0 - (g[0] - a.start0) / a.ext0) +
                     fabsf((box[1] - a.start1) / a.ext1 - (g[1] - a.start1) / a.ext1)) * a.w_reg;
  const float iou = iou3d_lidar(box, g);
  a.cost[o] = cls + reg + (-iou * a.w_iou);
  a.iou[o] = iou;
}

// ---------------------------------------------------------------------------------------------- dense targets
struct SplatArgs {
  const float *gt;
  const int32_t *gt_labels, *gt_off;
  float *heatmap;             // [B, C, H, W], zeroed by the entry point
  int B, C, H, W, gt_dim, min_radius;
  float vs0, vs1, osf, pc0, pc1;
  float k1a, k1b;             // (1 - o), (1 + o) as the fp32 values torch multiplies / divides by
  float k2;                   // (1 - o)
  float a3x4, b3, c3;         // 4 * (4 o), -2 o, (o - 1)
};

// radius rule of gaussian.py:60-86 element for element in fp32 (torch scalar-tensor arithmetic), then int() and
// max(min_radius, .) (transfusion_head.py:1193-1194)
__device__ __forceinline__ int splat_radius(float height, float width, const SplatArgs &a) {
  const float b1 = height + width;
  const float c1 = width * height * a.k1a / a.k1b;
  const float r1 = (b1 + sqrtf(b1 * b1 - 4.f * c1)) / 2.f;
  const float b2 = 2.f * (height + width);
  const float c2 = a.k2 * width * height;
  const float r2 = (b2 + sqrtf(b2 * b2 - 16.f * c2)) / 2.f;
  const float b3 = a.b3 * (height + width);
  const float c3 = a.c3 * width * height;
  const float r3 = (b3 + sqrtf(b3 * b3 - a.a3x4 * c3)) / 2.f;
  const float r = fminf(fminf(r1, r2), r3);
  const int ri = (int)r;
  return ri > a.min_radius ? ri : a.min_radius;
}

__global__ __launch_bounds__(256) void splat_kernel(SplatArgs a, int total) {
  const int gi = blockIdx.x;
  if (gi >= total) return;
  int b = 0;
  while (b + 1 < a.B && gi >= a.gt_off[b + 1]) ++b;
  const float *g = a.gt + (size_t)gi * a.gt_dim;
  const int lab = a.gt_labels[gi];
  if (lab < 0 || lab >= a.C) return;
  const float width = g[3] / a.vs0 / a.osf, length = g[4] / a.vs1 / a.osf;
  if (!(width > 0.f && length > 0.f)) return;
  const int radius = splat_radius(length, width, a);
  const int cx = (int)((g[0] - a.pc0) / a.vs0 / a.osf), cy = (int)((g[1] - a.pc1) / a.vs1 / a.osf);
  const int x0 = max(cx - radius, 0), x1 = min(cx + radius + 1, a.W), y0 = max(cy - radius, 0), y1 = min(cy + radius + 1, a.H);
  if (x1 <= x0 || y1 <= y0) return;
  const double sigma = (double)(2 * radius + 1) / 6.0, den = 2.0 * sigma * sigma;
  const int w = x1 - x0, n = w * (y1 - y0);
  int *plane = (int *)(a.heatmap + ((size_t)b * a.C + lab) * a.H * a.W);
  for (int i = threadIdx.x; i < n; i += 256) {
    const int y = y0 + i / w, x = x0 + i % w;
    const double dx = (double)(x - cx), dy = (double)(y - cy);
    double v = exp(-(dx * dx + dy * dy) / den);
    if (v < DBL_EPSILON) v = 0.0;                       // h[h < eps * h.max()] = 0, the patch's peak is exp(0) = 1
    atomicMax(plane + (size_t)y * a.W + x, __float_as_int((float)v));
  }
}

// ---------------------------------------------------------------------------------------------- Gaussian focal loss
constexpr int GF_CHUNK = 2048;

__device__ __forceinline__ double block_sum256(double v, double *s_red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// logits element (b, c, pix) at b * sb + c * sc + pix * sp; target / grad contiguous [B, C, HW]
template <bool GRAD>
__global__ __launch_bounds__(256) void gfocal_kernel(const float *__restrict__ logits, long long sb, long long sc, long long sp,
                                                     const float *__restrict__ target, long long n, int C, int HW, float alpha,
                                                     float gamma, float *__restrict__ grad, double *__restrict__ partial) {
  __shared__ double s_red[4];
  const long long i0 = (long long)blockIdx.x * GF_CHUNK;
  double acc = 0.0, ones = 0.0;
  for (int k = threadIdx.x; k < GF_CHUNK; k += 256) {
    const long long i = i0 + k;
    if (i >= n) break;
    const int pix = (int)(i % HW);
    const long long bc = i / HW;
    const int c = (int)(bc % C), b = (int)(bc / C);
    const float x = logits[b * sb + c * sc + pix * sp], t = target[i];
    const float s = 1.f / (1.f + expf(-x));
    const float p = fminf(fmaxf(s, 1e-4f), 1.f - 1e-4f);
    const float lp = logf(p + 1e-12f), l1p = logf(1.f - p + 1e-12f);
    const float negw = pow_e(1.f - t, gamma);
    const bool one = t == 1.f;
    const float pos = one ? -lp * pow_e(1.f - p, alpha) : 0.f;
    const float neg = -l1p * pow_e(p, alpha) * negw;
    acc += (double)(pos + neg);
    ones += one ? 1.0 : 0.0;
    if (GRAD) {
      const bool inside = s >= 1e-4f && s <= 1.f - 1e-4f;
      const float dpdx = inside ? p * (1.f - p) : 0.f;
      const float dpos = one ? -pow_e(1.f - p, alpha) / (p + 1e-12f) + alpha * pow_e(1.f - p, alpha - 1.f) * lp : 0.f;
      const float dneg = (pow_e(p, alpha) / (1.f - p + 1e-12f) - alpha * pow_e(p, alpha - 1.f) * l1p) * negw;
      grad[i] = (dpos + dneg) * dpdx;
    }
  }
  const double tot = block_sum256(acc, s_red);
  const double cnt = block_sum256(ones, s_red);
  if (threadIdx.x == 0) {
    partial[2 * (size_t)blockIdx.x] = tot;
    partial[2 * (size_t)blockIdx.x + 1] = cnt;
  }
}

// out[0] = loss_weight * sum / max(#ones, 1), out[1] = #ones, out[2] = loss_weight / max(#ones, 1) (gradient scale)
__global__ __launch_bounds__(256) void gfocal_finish_kernel(const double *__restrict__ partial, int chunks, float loss_weight,
                                                            float *__restrict__ out) {
  __shared__ double s_red[4];
  double s = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < chunks; i += 256) {
    s += partial[2 * (size_t)i];
    c += partial[2 * (size_t)i + 1];
  }
  const double S = block_sum256(s, s_red), Cn = block_sum256(c, s_red);
  if (threadIdx.x == 0) {
    const double avg = Cn > 1.0 ? Cn : 1.0;
    out[0] = (float)((double)loss_weight * S / avg);
    out[1] = (float)Cn;
    out[2] = (float)((double)loss_weight / avg);
  }
}

// ---------------------------------------------------------------------------------------------- query losses
struct QueryArgs {
  const float *rows;          // [B, P_all, ld] as MatchArgs::rows
  const int32_t *assigned;    // [B, P_all]: index into gt (global) or -1
  const float *iou;           // [B, P_all, gmax] from match_cost_kernel
  const float *gt;
  const int32_t *gt_labels, *gt_off;
  float *grad;                // [B, P_all, ld] or NULL: d(sum of the layer losses)/d rows
  float *out;                 // [2 * layers + 2]: per layer (loss_cls, loss_bbox), then num_pos, matched_ious
  int B, P_all, K, layers, ld, col_cls, num_classes, code, gt_dim, gmax;
  float enc_dx, enc_dy, pc0, pc1;             // encode: (x - pc0) / enc_dx
  float alpha, gamma, w_cls, w_bbox, pos_weight;
  float code_weights[DF3D_LOSS_MAX_CODES];
};

__global__ __launch_bounds__(1024) void query_loss_kernel(QueryArgs a) {
  __shared__ double s_red[16];
  __shared__ double s_np, s_miou;
  auto bsum = [&](double v) -> double {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += s_red[w];
    return t;
  };
  const int rows = a.B * a.P_all;
  // positives (all layers, all samples) and matched IoUs: mean over samples of sum(iou[pos]) / max(#pos, 1)
  double np = 0.0;
  for (int r = threadIdx.x; r < rows; r += 1024) np += a.assigned[r] >= 0 ? 1.0 : 0.0;
  np = bsum(np);
  double miou = 0.0;
  for (int b = 0; b < a.B; ++b) {
    double s = 0.0, c = 0.0;
    for (int p = threadIdx.x; p < a.P_all; p += 1024) {
      const int g = a.assigned[b * a.P_all + p];
      if (g >= 0) {
        const float v = a.iou[((size_t)b * a.P_all + p) * a.gmax + (g - a.gt_off[b])];
        s += (double)fminf(fmaxf(v, 0.f), 1.f);
        c += 1.0;
      }
    }
    s = bsum(s);
    c = bsum(c);
    miou += s / (c > 1.0 ? c : 1.0);
  }
  if (threadIdx.x == 0) {
    s_np = np;
    s_miou = miou / (a.B > 0 ? a.B : 1);
  }
  __syncthreads();
  const float inv = (float)(1.0 / (s_np > 1.0 ? s_np : 1.0));
  const float tiny = FLT_MIN;
  for (int l = 0; l < a.layers; ++l) {
    double lc = 0.0, lb = 0.0;
    for (int i = threadIdx.x; i < a.B * a.K; i += 1024) {
      const int b = i / a.K, p = l * a.K + i % a.K;
      const size_t r = (size_t)b * a.P_all + p;
      const float *row = a.rows + r * a.ld;
      float *grow = a.grad ? a.grad + r * a.ld : nullptr;
      const int g = a.assigned[r];
      const int lab = g >= 0 ? a.gt_labels[g] : a.num_classes;
      const float lw = g >= 0 ? (a.pos_weight <= 0.f ? 1.f : (float)(long long)a.pos_weight) : 1.f;   // label_weights is a LONG tensor
      float s = 0.f;
      for (int c = 0; c < a.num_classes; ++c) {
        const float x = row[a.col_cls + c];
        const float pr = 1.f / (1.f + expf(-x));
        float e, de;
        if (c == lab) {                                // -alpha (1-p)^gamma log p
          const float lg = logf(fmaxf(pr, tiny)), q = pow_e(1.f - pr, a.gamma);
          e = -a.alpha * q * lg;
          de = -a.alpha * (q / fmaxf(pr, tiny) - a.gamma * pow_e(1.f - pr, a.gamma - 1.f) * lg);
        } else {                                       // -(1-alpha) p^gamma log(1-p)
          const float lg = logf(fmaxf(1.f - pr, tiny)), q = pow_e(pr, a.gamma);
          e = -(1.f - a.alpha) * q * lg;
          de = -(1.f - a.alpha) * (a.gamma * pow_e(pr, a.gamma - 1.f) * lg - q / fmaxf(1.f - pr, tiny));
        }
        s += e * lw;
        if (grow) grow[a.col_cls + c] = a.w_cls * inv * lw * de * pr * (1.f - pr);
      }
      lc += (double)s;
      float t[DF3D_LOSS_MAX_CODES];
      if (g >= 0) {
        const float *gb = a.gt + (size_t)g * a.gt_dim;
        t[0] = (gb[0] - a.pc0) / a.enc_dx;
        t[1] = (gb[1] - a.pc1) / a.enc_dy;
        t[2] = gb[2] + gb[5] * 0.5f;
        t[3] = logf(gb[3]);
        t[4] = logf(gb[4]);
        t[5] = logf(gb[5]);
        t[6] = sinf(gb[6]);
        t[7] = cosf(gb[6]);
        if (a.code == 10) {
          t[8] = gb[7];
          t[9] = gb[8];
        }
      }
      float sb = 0.f;
      for (int c = 0; c < a.code; ++c) {
        const float wgt = g >= 0 ? a.code_weights[c] : 0.f;
        const float d = row[c] - (g >= 0 ? t[c] : 0.f);
        sb += fabsf(d) * wgt;
        if (grow) grow[c] = a.w_bbox * inv * wgt * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
      }
      lb += (double)sb;
    }
    lc = bsum(lc);
    lb = bsum(lb);
    if (threadIdx.x == 0) {
      a.out[2 * l] = (float)((double)a.w_cls * lc * (double)inv);
      a.out[2 * l + 1] = (float)((double)a.w_bbox * lb * (double)inv);
    }
  }
  if (threadIdx.x == 0) {
    a.out[2 * a.layers] = (float)s_np;
    a.out[2 * a.layers + 1] = (float)s_miou;
  }
}

}  // namespace df3d

using namespace df3d;

extern "C" int df3d_boxes_overlap_bev_xyxyr(const float *boxes_a, int na, const float *boxes_b, int nb, float *out,
                                            void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(na >= 0 && nb >= 0, "boxes_overlap_bev_xyxyr: bad sizes");
  if (na == 0 || nb == 0) return DF3D_OK;
  DF3D_CHECK_ARG(boxes_a && boxes_b && out, "boxes_overlap_bev_xyxyr: null argument");
  hipLaunchKernelGGL(overlap_xyxyr_kernel, dim3(cdiv(nb, 16), cdiv(na, 16)), dim3(256), 0, stream, boxes_a, na, boxes_b, nb,
                     out);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_tf_match_cost(const float *rows, int batch, int proposals, int ld, int col_cls, int num_classes,
                                  const float *gt, const int32_t *gt_labels, const int32_t *gt_off, int gt_dim, int gmax,
                                  const df3d_tf_match_cfg *cfg, float *cost, float *iou, float *boxes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(batch >= 0 && proposals >= 0 && gmax >= 0, "tf_match_cost: bad sizes");
  if (batch == 0 || proposals == 0) return DF3D_OK;
  DF3D_CHECK_ARG(rows && gt_off && cfg && (gmax == 0 || (gt && gt_labels && cost && iou)), "tf_match_cost: null argument");
  DF3D_CHECK_ARG(ld >= 8 && col_cls >= 8 && col_cls + num_classes <= ld && gt_dim >= 7, "tf_match_cost: bad row layout");
  MatchArgs a;
  a.rows = rows, a.gt = gt, a.gt_labels = gt_labels, a.gt_off = gt_off, a.cost = cost, a.iou = iou, a.boxes = boxes;
  a.B = batch, a.P = proposals, a.ld = ld, a.col_cls = col_cls, a.num_classes = num_classes, a.gt_dim = gt_dim, a.gmax = gmax;
  a.osf = cfg->out_size_factor, a.vs0 = cfg->voxel_size[0], a.vs1 = cfg->voxel_size[1];
  a.pc0 = cfg->pc_range[0], a.pc1 = cfg->pc_range[1];
  a.start0 = cfg->point_cloud_range[0], a.start1 = cfg->point_cloud_range[1];
  a.ext0 = cfg->point_cloud_range[3] - cfg->point_cloud_range[0];
  a.ext1 = cfg->point_cloud_range[4] - cfg->point_cloud_range[1];
  a.w_cls = cfg->cls_weight, a.alpha = cfg->cls_alpha, a.gamma = cfg->cls_gamma, a.eps = cfg->cls_eps;
  a.w_reg = cfg->reg_weight, a.w_iou = cfg->iou_weight;
  const int gx = gmax > 0 ? cdiv(gmax, 16) : 1;
  hipLaunchKernelGGL(match_cost_kernel, dim3(gx, cdiv(proposals, 16), batch), dim3(256), 0, stream, a);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_draw_heatmap_gaussian(const float *gt, const int32_t *gt_labels, const int32_t *gt_off, int gt_dim,
                                          int total, int batch, int num_classes, int height, int width,
                                          const df3d_tf_splat_cfg *cfg, float *heatmap, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(batch >= 0 && num_classes > 0 && height > 0 && width > 0 && total >= 0 && gt_dim >= 5,
                 "draw_heatmap_gaussian: bad sizes");
  if (batch == 0) return DF3D_OK;
  DF3D_CHECK_ARG(heatmap && cfg && gt_off && (total == 0 || (gt && gt_labels)), "draw_heatmap_gaussian: null argument");
  DF3D_HIP(hipMemsetAsync(heatmap, 0, (size_t)batch * num_classes * height * width * sizeof(float), stream));
  if (total == 0) return DF3D_OK;
  SplatArgs a;
  a.gt = gt, a.gt_labels = gt_labels, a.gt_off = gt_off, a.heatmap = heatmap;
  a.B = batch, a.C = num_classes, a.H = height, a.W = width, a.gt_dim = gt_dim, a.min_radius = cfg->min_radius;
  a.vs0 = cfg->voxel_size[0], a.vs1 = cfg->voxel_size[1], a.osf = cfg->out_size_factor;
  a.pc0 = cfg->point_cloud_range[0], a.pc1 = cfg->point_cloud_range[1];
  const double o = cfg->gaussian_overlap;          // Python evaluates these in double, torch rounds them to fp32
  a.k1a = (float)(1.0 - o), a.k1b = (float)(1.0 + o), a.k2 = (float)(1.0 - o);
  a.a3x4 = (float)(4.0 * (4.0 * o)), a.b3 = (float)(-2.0 * o), a.c3 = (float)(o - 1.0);
  hipLaunchKernelGGL(splat_kernel, dim3(total), dim3(256), 0, stream, a, total);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" size_t df3d_gaussian_focal_loss_workspace_bytes(long long n) {
  return n <= 0 ? 0 : (size_t)cdiv(n, GF_CHUNK) * 2 * sizeof(double);
}

extern "C" int df3d_gaussian_focal_loss(const float *logits, long long stride_b, long long stride_c, long long stride_pix,
                                        const float *target, int batch, int num_classes, int hw, float alpha, float gamma,
                                        float loss_weight, float *grad, float *out, void *workspace, size_t workspace_bytes,
                                        void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const long long n = (long long)batch * num_classes * hw;
  DF3D_CHECK_ARG(batch > 0 && num_classes > 0 && hw > 0, "gaussian_focal_loss: bad sizes");
  DF3D_CHECK_ARG(logits && target && out && workspace, "gaussian_focal_loss: null argument");
  DF3D_CHECK_ARG(workspace_bytes >= df3d_gaussian_focal_loss_workspace_bytes(n), "gaussian_focal_loss: workspace too small");
  const int chunks = cdiv(n, GF_CHUNK);
  double *partial = (double *)workspace;
  if (grad)
    hipLaunchKernelGGL(gfocal_kernel<true>, dim3(chunks), dim3(256), 0, stream, logits, stride_b, stride_c, stride_pix,
                       target, n, num_classes, hw, alpha, gamma, grad, partial);
  else
    hipLaunchKernelGGL(gfocal_kernel<false>, dim3(chunks), dim3(256), 0, stream, logits, stride_b, stride_c, stride_pix,
                       target, n, num_classes, hw, alpha, gamma, grad, partial);
  DF3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(gfocal_finish_kernel, dim3(1), dim3(256), 0, stream, partial, chunks, loss_weight, out);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_tf_query_loss(const float *rows, const int32_t *assigned, const float *iou, int batch, int proposals_all,
                                  int proposals, int ld, int col_cls, int num_classes, int code_size, const float *gt,
                                  const int32_t *gt_labels, const int32_t *gt_off, int gt_dim, int gmax,
                                  const df3d_tf_loss_cfg *cfg, float *grad, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(batch > 0 && proposals > 0 && proposals_all > 0 && proposals_all % proposals == 0, "tf_query_loss: bad sizes");
  DF3D_CHECK_ARG(rows && assigned && gt_off && cfg && out, "tf_query_loss: null argument");
  DF3D_CHECK_ARG((code_size == 8 || code_size == 10) && code_size <= DF3D_LOSS_MAX_CODES && col_cls >= code_size &&
                     col_cls + num_classes <= ld && (gmax == 0 || (gt && gt_labels && iou && gt_dim >= code_size - 1)),
                 "tf_query_loss: bad row layout");
  QueryArgs a;
  a.rows = rows, a.assigned = assigned, a.iou = iou, a.gt = gt, a.gt_labels = gt_labels, a.gt_off = gt_off;
  a.grad = grad, a.out = out;
  a.B = batch, a.P_all = proposals_all, a.K = proposals, a.layers = proposals_all / proposals, a.ld = ld, a.col_cls = col_cls;
  a.num_classes = num_classes, a.code = code_size, a.gt_dim = gt_dim, a.gmax = gmax;
  a.enc_dx = cfg->encode_step[0], a.enc_dy = cfg->encode_step[1], a.pc0 = cfg->pc_range[0], a.pc1 = cfg->pc_range[1];
  a.alpha = cfg->cls_alpha, a.gamma = cfg->cls_gamma, a.w_cls = cfg->cls_loss_weight, a.w_bbox = cfg->bbox_loss_weight;
  a.pos_weight = cfg->pos_weight;
  for (int c = 0; c < DF3D_LOSS_MAX_CODES; ++c) a.code_weights[c] = c < code_size ? cfg->code_weights[c] : 0.f;
  hipLaunchKernelGGL(query_loss_kernel, dim3(1), dim3(1024), 0, stream, a);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}
