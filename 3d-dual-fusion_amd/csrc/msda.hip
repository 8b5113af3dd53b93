// Multi-scale deformable attention forward for gfx950.
//
// Semantics = the reference kernel ms_deformable_im2col_gpu_kernel
// (CP/det3d/models/model_utils/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299, bilinear
// helper :33-84): h_im = loc_h*H - 0.5, w_im = loc_w*W - 0.5; a sample contributes iff
// -1 < h_im < H and -1 < w_im < W; corners outside the map read 0;
// out[b,q,m,:] = sum_{l,p} w[b,q,m,l,p] * bilinear(value[b, level l, m, :]).
//
// The reference runs one thread per output CHANNEL (16 lanes repeat the same index math and
// share one 64-byte fetch) in 1024-thread blocks, chunked over the batch by im2col_step.
// Here a lane owns VEC=4 channels (one 16-byte load per corner), so D=16 needs 4 lanes per
// (query, head): a wave covers 16 (q, m) pairs = two whole queries of the 8-head layout, its
// 64 x 16 B stores are one contiguous 1 KiB segment, and the four lanes of a (q, m) group
// split the sampling points between them: each lane fetches one point's (x, y, w) and the
// group exchanges them with wave shuffles instead of re-reading them 16 times.
// Algorithmic bytes: min(N*S*M*D, N*Lq*M*L*P*4*D)*4 + N*Lq*M*L*P*3*4 +
// N*Lq*M*D*4.  Bound: HBM/L2 gather bandwidth (10 flop per fetched element).
#include "common.h"

namespace df3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MsdaArgs {
  const float *value;
  const int64_t *shapes, *lstart;
  const float *loc, *aw;
  float *out;
  int N, S, M, D, Lq, L, P;
};

// Fast path: D % 4 == 0 and (D/4) a power of two <= 16 (D = 4, 8, 16, 32, 64).
template <int LPG /*lanes per (q,m) group = D/4*/>
__global__ __launch_bounds__(256) void msda_vec4_kernel(MsdaArgs a) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over N*Lq*M*LPG
  const long long total = (long long)a.N * a.Lq * a.M * LPG;
  const bool live = gid < total;
  const long long qm = live ? gid / LPG : (total - 1) / LPG;  // (b*Lq + q)*M + m
  const int sub = (int)(gid % LPG);
  const int m = (int)(qm % a.M);
  const int b = (int)(qm / ((long long)a.M * a.Lq));
  const int LP = a.L * a.P;
  const float *loc = a.loc + (size_t)qm * LP * 2;
  const float *aw = a.aw + (size_t)qm * LP;
  const int lane = threadIdx.x & 63;
  const int gbase = lane & ~(LPG - 1);
  const int qstride = a.M * a.D;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int lp0 = 0; lp0 < LP; lp0 += LPG) {
    // lane `sub` of the group fetches point lp0+sub
    float mx = 0.f, my = 0.f, mw = 0.f;
    int lp = lp0 + sub;
    if (lp < LP) {
      mx = loc[lp * 2];
      my = loc[lp * 2 + 1];
      mw = aw[lp];
    }
    const int cnt = (LP - lp0) < LPG ? (LP - lp0) : LPG;
    for (int i = 0; i < cnt; ++i) {
      float lx = __shfl(mx, gbase + i, 64);
      float ly = __shfl(my, gbase + i, 64);
      float w = __shfl(mw, gbase + i, 64);
      int l = (lp0 + i) / a.P;
      int H = (int)a.shapes[l * 2], W = (int)a.shapes[l * 2 + 1];
      const float *vbase = a.value + ((size_t)b * a.S + (size_t)a.lstart[l]) * qstride + m * a.D + sub * 4;
      float h_im = ly * (float)H - 0.5f;
      float w_im = lx * (float)W - 0.5f;
      if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
        int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        int h_high = h_low + 1, w_high = w_low + 1;
        float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
        float hh = 1.f - lh, hw = 1.f - lw;
        f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 v1 = z, v2 = z, v3 = z, v4 = z;
        const size_t hs = (size_t)W * qstride;
        if (h_low >= 0 && w_low >= 0) v1 = *(const f32x4 *)(vbase + h_low * hs + (size_t)w_low * qstride);
        if (h_low >= 0 && w_high <= W - 1) v2 = *(const f32x4 *)(vbase + h_low * hs + (size_t)w_high * qstride);
        if (h_high <= H - 1 && w_low >= 0) v3 = *(const f32x4 *)(vbase + h_high * hs + (size_t)w_low * qstride);
        if (h_high <= H - 1 && w_high <= W - 1) v4 = *(const f32x4 *)(vbase + h_high * hs + (size_t)w_high * qstride);
        float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        acc += (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4) * w;
      }
    }
  }
  if (live) *(f32x4 *)(a.out + (size_t)qm * a.D + sub * 4) = acc;
}

// General path: one thread per output channel (any D).
__global__ __launch_bounds__(256) void msda_scalar_kernel(MsdaArgs a) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)a.N * a.Lq * a.M * a.D;
  if (gid >= total) return;
  const int c = (int)(gid % a.D);
  const long long qm = gid / a.D;
  const int m = (int)(qm % a.M);
  const int b = (int)(qm / ((long long)a.M * a.Lq));
  const int LP = a.L * a.P;
  const float *loc = a.loc + (size_t)qm * LP * 2;
  const float *aw = a.aw + (size_t)qm * LP;
  const int qstride = a.M * a.D;
  float acc = 0.f;
  for (int l = 0; l < a.L; ++l) {
    int H = (int)a.shapes[l * 2], W = (int)a.shapes[l * 2 + 1];
    const float *vbase = a.value + ((size_t)b * a.S + (size_t)a.lstart[l]) * qstride + m * a.D + c;
    for (int p = 0; p < a.P; ++p) {
      float lx = loc[(l * a.P + p) * 2], ly = loc[(l * a.P + p) * 2 + 1], w = aw[l * a.P + p];
      float h_im = ly * (float)H - 0.5f;
      float w_im = lx * (float)W - 0.5f;
      if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
        int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        int h_high = h_low + 1, w_high = w_low + 1;
        float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
        float hh = 1.f - lh, hw = 1.f - lw;
        float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
        const size_t hs = (size_t)W * qstride;
        if (h_low >= 0 && w_low >= 0) v1 = vbase[h_low * hs + (size_t)w_low * qstride];
        if (h_low >= 0 && w_high <= W - 1) v2 = vbase[h_low * hs + (size_t)w_high * qstride];
        if (h_high <= H - 1 && w_low >= 0) v3 = vbase[h_high * hs + (size_t)w_low * qstride];
        if (h_high <= H - 1 && w_high <= W - 1) v4 = vbase[h_high * hs + (size_t)w_high * qstride];
        float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        acc += (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4) * w;
      }
    }
  }
  a.out[gid] = acc;
}


// ---------------------------------------------------------------------------------------------------------
// Backward.  Reference: ms_deformable_col2im_gpu_kernel* and
// ms_deform_attn_col2im_bilinear (ms_deform_im2col_cuda.cuh:87-232, 301-921): per (b, q, m, c) and sampling point
//   grad_value[corner]   += w_corner * attn * g                (atomics; four corners, zero outside the map)
//   grad_attn[b,q,m,l,p] += g * bilinear(value)                (sum over the D channels)
//   grad_loc[..,0]       += W * (dval/dw_im) * attn * g        (x), grad_loc[..,1] += H * (dval/dh_im) * attn * g (y)
// with dval/dh_im = -hw*v1 - lw*v2 + hw*v3 + lw*v4 and dval/dw_im = -hh*v1 + hh*v2 - lh*v3 + lh*v4.
// The reference has seven kernel variants that differ in how the channel sums are reduced (shared memory trees,
// block-size specialisations, global atomics for > 1024 channels).  Here the forward's layout is reused: a lane owns
// four channels, the D/4 lanes of a (q, m) group reduce the two location gradients and the weight gradient with
// wave shuffles (no LDS, no atomics for them), and the value gradient uses the hardware fp32 atomic add.
struct MsdaBwdArgs {
  const float *value;
  const int64_t *shapes, *lstart;
  const float *loc, *aw, *gout;
  float *gvalue, *gloc, *gaw;
  int N, S, M, D, Lq, L, P;
};

__device__ __forceinline__ void atomic_add4(float *p, f32x4 v) {
  unsafeAtomicAdd(p, v[0]);
  unsafeAtomicAdd(p + 1, v[1]);
  unsafeAtomicAdd(p + 2, v[2]);
  unsafeAtomicAdd(p + 3, v[3]);
}

__device__ __forceinline__ float hsum4(f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }

template <int LPG>
__global__ __launch_bounds__(256) void msda_bwd_vec4_kernel(MsdaBwdArgs a) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)a.N * a.Lq * a.M * LPG;
  const bool live = gid < total;
  const long long qm = live ? gid / LPG : (total - 1) / LPG;
  const int sub = (int)(gid % LPG);
  const int m = (int)(qm % a.M);
  const int b = (int)(qm / ((long long)a.M * a.Lq));
  const int LP = a.L * a.P;
  const float *loc = a.loc + (size_t)qm * LP * 2;
  const float *aw = a.aw + (size_t)qm * LP;
  const int lane = threadIdx.x & 63;
  const int gbase = lane & ~(LPG - 1);
  const int qstride = a.M * a.D;
  f32x4 g = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (live) g = *(const f32x4 *)(a.gout + (size_t)qm * a.D + sub * 4);
  // A (query, head) whose upstream gradient is all zero contributes nothing: every product below carries g.  The padded
  // rows of the zero-padded per-camera query lists are exactly that (nobody reads their outputs), and they all sample around
  // reference point (0, 0): ~8 k rows per image adding zeros to the same few cache lines of grad_value (the atomics of one line
  // serialise in the L2: 3.4 ms per call at the TransFusion training shape, 0.12 ms for the forward).
  int nz = (g[0] != 0.f) | (g[1] != 0.f) | (g[2] != 0.f) | (g[3] != 0.f);
#pragma unroll
  for (int d = 1; d < LPG; d <<= 1) nz |= __shfl_xor(nz, d, 64);
  const bool work = live && nz;
  for (int lp0 = 0; lp0 < LP; lp0 += LPG) {
    float mx = 0.f, my = 0.f, mw = 0.f;
    const int lp = lp0 + sub;
    if (lp < LP) {
      mx = loc[lp * 2];
      my = loc[lp * 2 + 1];
      mw = aw[lp];
    }
    float keep_x = 0.f, keep_y = 0.f, keep_w = 0.f;       // gradients of the point this lane fetched
    const int cnt = (LP - lp0) < LPG ? (LP - lp0) : LPG;
    for (int i = 0; i < cnt; ++i) {
      const float lx = __shfl(mx, gbase + i, 64), ly = __shfl(my, gbase + i, 64), w = __shfl(mw, gbase + i, 64);
      const int l = (lp0 + i) / a.P;
      const int H = (int)a.shapes[l * 2], W = (int)a.shapes[l * 2 + 1];
      const size_t off = ((size_t)b * a.S + (size_t)a.lstart[l]) * qstride + m * a.D + sub * 4;
      const float h_im = ly * (float)H - 0.5f, w_im = lx * (float)W - 0.5f;
      float px = 0.f, py = 0.f, pw = 0.f;
      if (work && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const int h_high = h_low + 1, w_high = w_low + 1;
        const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
        const float hh = 1.f - lh, hw = 1.f - lw;
        const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 v1 = z, v2 = z, v3 = z, v4 = z;
        const size_t hs = (size_t)W * qstride;
        const f32x4 tg = g * w;                              // top_grad * attention weight
        const size_t o1 = off + h_low * hs + (size_t)w_low * qstride, o2 = off + h_low * hs + (size_t)w_high * qstride;
        const size_t o3 = off + h_high * hs + (size_t)w_low * qstride, o4 = off + h_high * hs + (size_t)w_high * qstride;
        if (h_low >= 0 && w_low >= 0) {
          v1 = *(const f32x4 *)(a.value + o1);
          atomic_add4(a.gvalue + o1, tg * (hh * hw));
        }
        if (h_low >= 0 && w_high <= W - 1) {
          v2 = *(const f32x4 *)(a.value + o2);
          atomic_add4(a.gvalue + o2, tg * (hh * lw));
        }
        if (h_high <= H - 1 && w_low >= 0) {
          v3 = *(const f32x4 *)(a.value + o3);
          atomic_add4(a.gvalue + o3, tg * (lh * hw));
        }
        if (h_high <= H - 1 && w_high <= W - 1) {
          v4 = *(const f32x4 *)(a.value + o4);
          atomic_add4(a.gvalue + o4, tg * (lh * lw));
        }
        const f32x4 val = (hh * hw) * v1 + (hh * lw) * v2 + (lh * hw) * v3 + (lh * lw) * v4;
        const f32x4 dh = hw * (v3 - v1) + lw * (v4 - v2), dw = hh * (v2 - v1) + lh * (v4 - v3);
        pw = hsum4(g * val);
        px = (float)W * hsum4(dw * tg);
        py = (float)H * hsum4(dh * tg);
      }
#pragma unroll
      for (int d = 1; d < LPG; d <<= 1) {                    // sum over the lanes (channel quads) of the group
        px += __shfl_xor(px, d, 64);
        py += __shfl_xor(py, d, 64);
        pw += __shfl_xor(pw, d, 64);
      }
      if (sub == i) {
        keep_x = px;
        keep_y = py;
        keep_w = pw;
      }
    }
    if (live && lp < LP) {
      a.gloc[((size_t)qm * LP + lp) * 2] = keep_x;
      a.gloc[((size_t)qm * LP + lp) * 2 + 1] = keep_y;
      a.gaw[(size_t)qm * LP + lp] = keep_w;
    }
  }
}

// any D: one thread per channel, global atomics for all three gradients (the reference's fallback,
// ms_deform_im2col_cuda.cuh:818-921)
__global__ __launch_bounds__(256) void msda_bwd_scalar_kernel(MsdaBwdArgs a) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)a.N * a.Lq * a.M * a.D;
  if (gid >= total) return;
  const int c = (int)(gid % a.D);
  const long long qm = gid / a.D;
  const int m = (int)(qm % a.M);
  const int b = (int)(qm / ((long long)a.M * a.Lq));
  const int LP = a.L * a.P;
  const int qstride = a.M * a.D;
  const float g = a.gout[gid];
  for (int lp = 0; lp < LP; ++lp) {
    const int l = lp / a.P;
    const int H = (int)a.shapes[l * 2], W = (int)a.shapes[l * 2 + 1];
    const float lx = a.loc[((size_t)qm * LP + lp) * 2], ly = a.loc[((size_t)qm * LP + lp) * 2 + 1];
    const float w = a.aw[(size_t)qm * LP + lp];
    const float h_im = ly * (float)H - 0.5f, w_im = lx * (float)W - 0.5f;
    if (!(h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W)) continue;
    const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im), h_high = h_low + 1, w_high = w_low + 1;
    const float lh = h_im - (float)h_low, lw = w_im - (float)w_low, hh = 1.f - lh, hw = 1.f - lw;
    const size_t off = ((size_t)b * a.S + (size_t)a.lstart[l]) * qstride + m * a.D + c, hs = (size_t)W * qstride;
    float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
    const float tg = g * w;
    if (h_low >= 0 && w_low >= 0) {
      v1 = a.value[off + h_low * hs + (size_t)w_low * qstride];
      unsafeAtomicAdd(a.gvalue + off + h_low * hs + (size_t)w_low * qstride, tg * hh * hw);
    }
    if (h_low >= 0 && w_high <= W - 1) {
      v2 = a.value[off + h_low * hs + (size_t)w_high * qstride];
      unsafeAtomicAdd(a.gvalue + off + h_low * hs + (size_t)w_high * qstride, tg * hh * lw);
    }
    if (h_high <= H - 1 && w_low >= 0) {
      v3 = a.value[off + h_high * hs + (size_t)w_low * qstride];
      unsafeAtomicAdd(a.gvalue + off + h_high * hs + (size_t)w_low * qstride, tg * lh * hw);
    }
    if (h_high <= H - 1 && w_high <= W - 1) {
      v4 = a.value[off + h_high * hs + (size_t)w_high * qstride];
      unsafeAtomicAdd(a.gvalue + off + h_high * hs + (size_t)w_high * qstride, tg * lh * lw);
    }
    unsafeAtomicAdd(a.gaw + (size_t)qm * LP + lp, g * (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4));
    unsafeAtomicAdd(a.gloc + ((size_t)qm * LP + lp) * 2, (float)W * tg * (hh * (v2 - v1) + lh * (v4 - v3)));
    unsafeAtomicAdd(a.gloc + ((size_t)qm * LP + lp) * 2 + 1, (float)H * tg * (hw * (v3 - v1) + lw * (v4 - v2)));
  }
}

}  // namespace df3d

using namespace df3d;

extern "C" int df3d_ms_deform_attn_forward(const float *value, const int64_t *spatial_shapes,
                                           const int64_t *level_start_index, const float *sampling_loc,
                                           const float *attn_weight, int N, int S, int M, int D, int Lq, int L, int P,
                                           float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(value && spatial_shapes && level_start_index && sampling_loc && attn_weight && out,
                 "ms_deform_attn_forward: null argument");
  DF3D_CHECK_ARG(N >= 0 && S > 0 && M > 0 && D > 0 && Lq >= 0 && L > 0 && P > 0, "ms_deform_attn_forward: bad sizes");
  if (N == 0 || Lq == 0) return DF3D_OK;
  MsdaArgs a = {value, spatial_shapes, level_start_index, sampling_loc, attn_weight, out, N, S, M, D, Lq, L, P};
  int lpg = (D % 4 == 0) ? D / 4 : 0;
  long long total;
  switch (lpg) {
#define DF3D_MSDA_CASE(G)                                                                                     \
  case G:                                                                                                     \
    total = (long long)N * Lq * M * G;                                                                        \
    hipLaunchKernelGGL(msda_vec4_kernel<G>, dim3(cdiv(total, 256)), dim3(256), 0, stream, a);                 \
    break;
    DF3D_MSDA_CASE(1)
    DF3D_MSDA_CASE(2)
    DF3D_MSDA_CASE(4)
    DF3D_MSDA_CASE(8)
    DF3D_MSDA_CASE(16)
#undef DF3D_MSDA_CASE
    default:
      total = (long long)N * Lq * M * D;
      hipLaunchKernelGGL(msda_scalar_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, a);
  }
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_ms_deform_attn_backward(const float *value, const int64_t *spatial_shapes,
                                            const int64_t *level_start_index, const float *sampling_loc,
                                            const float *attn_weight, const float *grad_output, int N, int S, int M, int D,
                                            int Lq, int L, int P, float *grad_value, float *grad_sampling_loc,
                                            float *grad_attn_weight, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(N >= 0 && S > 0 && M > 0 && D > 0 && Lq >= 0 && L > 0 && P > 0, "ms_deform_attn_backward: bad sizes");
  if (N == 0) return DF3D_OK;
  DF3D_CHECK_ARG(value && spatial_shapes && level_start_index && grad_value, "ms_deform_attn_backward: null argument");
  DF3D_HIP(hipMemsetAsync(grad_value, 0, (size_t)N * S * M * D * sizeof(float), stream));
  if (Lq == 0) return DF3D_OK;
  DF3D_CHECK_ARG(sampling_loc && attn_weight && grad_output && grad_sampling_loc && grad_attn_weight,
                 "ms_deform_attn_backward: null argument");
  MsdaBwdArgs a = {value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                   grad_value, grad_sampling_loc, grad_attn_weight, N, S, M, D, Lq, L, P};
  const int lpg = (D % 4 == 0) ? D / 4 : 0;
  long long total;
  switch (lpg) {
#define DF3D_MSDA_BWD_CASE(G)                                                                                 \
  case G:                                                                                                     \
    total = (long long)N * Lq * M * G;                                                                        \
    hipLaunchKernelGGL(msda_bwd_vec4_kernel<G>, dim3(cdiv(total, 256)), dim3(256), 0, stream, a);             \
    break;
    DF3D_MSDA_BWD_CASE(1)
    DF3D_MSDA_BWD_CASE(2)
    DF3D_MSDA_BWD_CASE(4)
    DF3D_MSDA_BWD_CASE(8)
    DF3D_MSDA_BWD_CASE(16)
#undef DF3D_MSDA_BWD_CASE
    default: {
      const size_t nlp = (size_t)N * Lq * M * L * P;
      DF3D_HIP(hipMemsetAsync(grad_sampling_loc, 0, nlp * 2 * sizeof(float), stream));
      DF3D_HIP(hipMemsetAsync(grad_attn_weight, 0, nlp * sizeof(float), stream));
      total = (long long)N * Lq * M * D;
      hipLaunchKernelGGL(msda_bwd_scalar_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, a);
    }
  }
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}
