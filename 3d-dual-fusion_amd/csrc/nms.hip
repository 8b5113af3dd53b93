// Detection tail, part 1: rotated BEV overlap / IoU and greedy NMS.
//
// Reference: CP/det3d/ops/iou3d_nms/src/iou3d_nms_kernel.cu:36-245 (box_overlap / iou_bev), :262-306 (nms_kernel:
// 64x64 bit-matrix tiles), :309-366 (axis-aligned variant) and iou3d_nms.cpp:88-188 (the matrix is copied to the
// HOST and reduced by a serial loop there), called per (task, sample) from Python
// (CP/det3d/core/bbox/box_torch_ops.py:248-279).  Here: the upper triangle of the bit matrix for ALL lists of a
// batch in one launch (with a conservative centre-distance rejection in front of the polygon clipping), and the
// greedy reduction on the device by one wavefront per list -- the 64 boxes of a tile are resolved with scalar
// bit arithmetic on wave-uniform values (v_readlane), the suppression words of later tiles are OR-ed lane-parallel.
// No host round trip; lists are independent, so B x tasks lists cost one launch pair.
//
// The polygon arithmetic is float, operation for operation the reference's (contraction into FMAs disabled), so that
// results differ from its CPU path (iou3d_cpu.cpp) only through the last ulps of cosf / sinf / atan2f.
#include "common.h"

#pragma clang fp contract(off)

namespace df3d {

struct Pt {
  float x, y;
};

__device__ __forceinline__ float cross3(Pt p1, Pt p2, Pt p0) {
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

__device__ __forceinline__ bool rect_cross(Pt p1, Pt p2, Pt q1, Pt q2) {
  return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
         fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

__device__ __forceinline__ bool in_box2d(const float *box, Pt p) {
  const float margin = 1e-2f;
  const float ca = cosf(-box[6]), sa = sinf(-box[6]);
  const float rx = (p.x - box[0]) * ca + (p.y - box[1]) * (-sa);
  const float ry = (p.x - box[0]) * sa + (p.y - box[1]) * ca;
  return fabsf(rx) < box[3] / 2 + margin && fabsf(ry) < box[4] / 2 + margin;
}

__device__ __forceinline__ bool intersection(Pt p1, Pt p0, Pt q1, Pt q0, Pt &ans) {
  if (!rect_cross(p0, p1, q0, q1)) return false;
  const float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0);
  const float s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
  const float s5 = cross3(q1, p1, p0);
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

__device__ __forceinline__ void corners(const float *box, Pt *c) {
  const float hx = box[3] / 2, hy = box[4] / 2, ca = cosf(box[6]), sa = sinf(box[6]);
  const float x1 = box[0] - hx, y1 = box[1] - hy, x2 = box[0] + hx, y2 = box[1] + hy;
  const float px[4] = {x1, x2, x2, x1}, py[4] = {y1, y1, y2, y2};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    c[k].x = (px[k] - box[0]) * ca + (py[k] - box[1]) * (-sa) + box[0];
    c[k].y = (px[k] - box[0]) * sa + (py[k] - box[1]) * ca + box[1];
  }
  c[4] = c[0];
}

__device__ float box_overlap(const float *a, const float *b) {
  Pt ca[5], cb[5], pts[16], ctr = {0.f, 0.f};
  int cnt = 0;
  corners(a, ca);
  corners(b, cb);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      Pt x;
      if (intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], x)) {
        pts[cnt] = x;
        ctr.x += x.x;
        ctr.y += x.y;
        cnt++;
      }
    }
  for (int k = 0; k < 4; ++k) {
    if (in_box2d(a, cb[k])) {
      ctr.x += cb[k].x;
      ctr.y += cb[k].y;
      pts[cnt++] = cb[k];
    }
    if (in_box2d(b, ca[k])) {
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
        Pt t = pts[i];
        pts[i] = pts[i + 1];
        pts[i + 1] = t;
        float u = ang[i];
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

__device__ __forceinline__ float iou_rotated(const float *a, const float *b) {
  const float sa = a[3] * a[4], sb = b[3] * b[4], so = box_overlap(a, b);
  return so / fmaxf(sa + sb - so, 1e-8f);
}

__device__ __forceinline__ float iou_normal(const float *a, const float *b) {
  const float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  const float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f), inter = w * h;
  return inter / fmaxf(a[3] * a[4] + b[3] * b[4] - inter, 1e-8f);
}

// no overlap is possible when the centres are further apart than the two half diagonals (+ the 1e-2 corner margin):
// the reference then finds no crossing and no contained corner and returns exactly 0
__device__ __forceinline__ bool far_apart(const float *a, const float *b) {
  const float ra = 0.5f * sqrtf(a[3] * a[3] + a[4] * a[4]), rb = 0.5f * sqrtf(b[3] * b[3] + b[4] * b[4]);
  const float dx = a[0] - b[0], dy = a[1] - b[1], r = ra + rb + 0.1f;
  return dx * dx + dy * dy > r * r * 1.01f;
}

__global__ __launch_bounds__(256) void boxes_pairwise_kernel(const float *__restrict__ A, int na,
                                                             const float *__restrict__ B, int nb, int mode,
                                                             float *__restrict__ out) {
  const int j = blockIdx.x * 16 + (threadIdx.x & 15), i = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (i >= na || j >= nb) return;
  float a[7], b[7];
#pragma unroll
  for (int e = 0; e < 7; ++e) {
    a[e] = A[(size_t)i * 7 + e];
    b[e] = B[(size_t)j * 7 + e];
  }
  float v = 0.f;
  if (!far_apart(a, b)) v = mode ? iou_rotated(a, b) : box_overlap(a, b);
  out[(size_t)i * nb + j] = v;
}

// bit (i, j) = IoU(box i, box j) > thresh, j > i, 64 x 64 tiles; grid (col tile, row tile, list).
// Rotated mode in two phases, because the polygon clipping is ~100x the cost of the rejection test and only a few
// pairs of a tile need it: (1) every lane scans its row with the centre-distance test and records the surviving
// columns as a bit word; (2) the surviving (row, column) pairs of the whole tile are listed in LDS and clipped one
// pair per lane -- all 64 lanes busy, instead of the whole wave waiting whenever one lane met a close pair.
__global__ __launch_bounds__(64) void nms_mask_kernel(const float *__restrict__ boxes, const int32_t *__restrict__ counts,
                                                      int cap, int cbmax, float thresh, int mode,
                                                      unsigned long long *__restrict__ mask) {
  const int cb = blockIdx.x, rb = blockIdx.y, s = blockIdx.z;
  if (cb < rb) return;
  int n = counts ? counts[s] : cap;
  n = n < cap ? n : cap;
  if (rb * 64 >= n || cb * 64 >= n) return;
  const float *bx = boxes + (size_t)s * cap * 7;
  __shared__ float colb[64 * 7], rowb[64 * 7];
  __shared__ unsigned short pairs[64 * 64];
  __shared__ unsigned long long bitsL[64];
  const int t = threadIdx.x;
  const int csize = min(n - cb * 64, 64), rsize = min(n - rb * 64, 64);
  for (int e = t; e < csize * 7; e += 64) colb[e] = bx[(size_t)cb * 64 * 7 + e];
  for (int e = t; e < rsize * 7; e += 64) rowb[e] = bx[(size_t)rb * 64 * 7 + e];
  bitsL[t] = 0ull;
  __syncthreads();
  const int i = rb * 64 + t;
  const bool live = t < rsize;
  const float *a = rowb + t * 7;
  unsigned long long bits = 0ull, near = 0ull;
  const int start = rb == cb ? t + 1 : 0;
  if (live) {
    for (int j = start; j < csize; ++j) {
      const float *b = colb + j * 7;
      if (mode == DF3D_NMS_CIRCLE) {               // centre distance (CP/det3d/core/utils/circle_nms_jit.py:21-26)
        const float dx = a[0] - b[0], dy = a[1] - b[1];
        if (dx * dx + dy * dy <= thresh) bits |= 1ull << j;
      } else if (mode == DF3D_NMS_NORMAL) {
        if (iou_normal(a, b) > thresh) bits |= 1ull << j;
      } else if (thresh < 0.f || !far_apart(a, b)) {
        near |= 1ull << j;
      }
    }
  }
  if (mode == DF3D_NMS_ROTATED) {
    // exclusive prefix of the per-lane pair counts (wave64 scan by shuffles)
    const int mine = __popcll(near);
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d);
      if (t >= d) incl += v;
    }
    const int total = __shfl(incl, 63);
    int pos = incl - mine;
    unsigned long long m = near;
    while (m) {
      const int j = __ffsll((long long)m) - 1;
      m &= m - 1;
      pairs[pos++] = (unsigned short)((t << 6) | j);
    }
    __syncthreads();
    for (int p = t; p < total; p += 64) {
      const int r = pairs[p] >> 6, j = pairs[p] & 63;
      if (iou_rotated(rowb + r * 7, colb + j * 7) > thresh) atomicOr(&bitsL[r], 1ull << j);
    }
    __syncthreads();
    bits = bitsL[t];
  }
  if (live) mask[((size_t)s * cap + i) * cbmax + cb] = bits;
}

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l) {
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, l), hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}

// greedy reduction (iou3d_nms.cpp:118-133), one wavefront per list
__global__ __launch_bounds__(64) void nms_reduce_kernel(const unsigned long long *__restrict__ mask,
                                                        const int32_t *__restrict__ counts, int cap, int cbmax,
                                                        int max_keep, int32_t *__restrict__ keep,
                                                        int32_t *__restrict__ num_keep) {
  const int s = blockIdx.x, lane = threadIdx.x;
  int n = counts ? counts[s] : cap;
  n = n < cap ? n : cap;
  const int nb = (n + 63) / 64;
  const unsigned long long *m = mask + (size_t)s * cap * cbmax;
  int32_t *kp = keep + (size_t)s * cap;
  unsigned long long remv = 0ull;              // lane j: suppression word of column tile j
  int nk = 0;
  for (int b = 0; b < nb; ++b) {
    const int i = b * 64 + lane;
    const unsigned long long intra = i < n ? m[(size_t)i * cbmax + b] : 0ull;
    unsigned long long r = readlane64(remv, b);
    const int valid = min(n - b * 64, 64);
    if (valid < 64) r |= ~0ull << valid;
    unsigned long long kept = 0ull;
    for (int l = 0; l < valid; ++l) {
      if (!((r >> l) & 1ull)) {
        kept |= 1ull << l;
        r |= readlane64(intra, l);
      }
    }
    if ((kept >> lane) & 1ull) {
      const int pos = nk + __popcll(kept & ((1ull << lane) - 1ull));
      if (max_keep <= 0 || pos < max_keep) kp[pos] = i;
    }
    nk += __popcll(kept);
    if (max_keep > 0 && nk >= max_keep) break;
    if (lane > b && lane < nb) {               // later tiles: OR the rows of the boxes kept in this tile
      unsigned long long k2 = kept, acc = 0ull;
      const unsigned long long *row = m + (size_t)b * 64 * cbmax + lane;
      while (k2) {
        const int l = __ffsll((long long)k2) - 1;
        k2 &= k2 - 1;
        acc |= row[(size_t)l * cbmax];
      }
      remv |= acc;
    }
  }
  if (lane == 0) num_keep[s] = (max_keep > 0 && nk > max_keep) ? max_keep : nk;
}

}  // namespace df3d

using namespace df3d;

extern "C" int df3d_boxes_bev_pairwise(const float *boxes_a, int na, const float *boxes_b, int nb, int mode, float *out,
                                       void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(na >= 0 && nb >= 0 && (mode == 0 || mode == 1), "boxes_bev_pairwise: bad arguments");
  if (na == 0 || nb == 0) return DF3D_OK;
  DF3D_CHECK_ARG(boxes_a && boxes_b && out, "boxes_bev_pairwise: null argument");
  hipLaunchKernelGGL(boxes_pairwise_kernel, dim3(cdiv(nb, 16), cdiv(na, 16)), dim3(256), 0, stream, boxes_a, na, boxes_b,
                     nb, mode, out);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" size_t df3d_nms_bev_workspace_bytes(int lists, int cap) {
  if (lists <= 0 || cap <= 0) return 0;
  return (size_t)lists * cap * ((cap + 63) / 64) * sizeof(unsigned long long);
}

extern "C" int df3d_nms_bev(const float *boxes, const int32_t *counts, int lists, int cap, float thresh, int mode,
                            int max_keep, int32_t *keep, int32_t *num_keep, void *workspace, size_t workspace_bytes,
                            void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(lists >= 0 && cap >= 0, "nms_bev: bad sizes");
  if (lists == 0) return DF3D_OK;
  DF3D_CHECK_ARG(num_keep, "nms_bev: null num_keep");
  if (cap == 0) {
    DF3D_HIP(hipMemsetAsync(num_keep, 0, (size_t)lists * sizeof(int32_t), stream));
    return DF3D_OK;
  }
  DF3D_CHECK_ARG(boxes && keep && workspace, "nms_bev: null argument");
  DF3D_CHECK_ARG(cap <= 4096, "nms_bev: at most 4096 boxes per list (got %d)", cap);
  DF3D_CHECK_ARG(mode == DF3D_NMS_NORMAL || mode == DF3D_NMS_ROTATED || mode == DF3D_NMS_CIRCLE, "nms_bev: mode %d", mode);
  DF3D_CHECK_ARG(workspace_bytes >= df3d_nms_bev_workspace_bytes(lists, cap), "nms_bev: workspace too small");
  const int cb = (cap + 63) / 64;
  hipLaunchKernelGGL(nms_mask_kernel, dim3(cb, cb, lists), dim3(64), 0, stream, boxes, counts, cap, cb, thresh, mode,
                     (unsigned long long *)workspace);
  hipLaunchKernelGGL(nms_reduce_kernel, dim3(lists), dim3(64), 0, stream, (const unsigned long long *)workspace, counts,
                     cap, cb, max_keep, keep, num_keep);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}
