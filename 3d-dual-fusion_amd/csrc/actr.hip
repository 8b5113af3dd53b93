// Fused element-wise stages of the dual-query encoder layer (ACTR) for gfx950.
//
// The reference layer (CP/det3d/models/model_utils/actr_transformer.py:399-426,
// ops/modules/ms_deform_attn.py:129-166, attentions.py:111-117) issues ~25 small element-wise
// kernels per layer around its GEMMs (position adds, softmax, location arithmetic, residual adds,
// LayerNorms, the bidirectional gate).  On [B*6*Q, 128] fp32 rows each of them is a 15 MB pass that is
// launch/latency bound.  Here they are four kernels, one wave per query row, 16-byte accesses:
//   actr_prep        A = q + pos (offset query), Bw = q + qi + 2 pos (weight query)        3 reads, 2 writes
//   msda_fused       softmax over the L*P logits + sampling-location arithmetic inside the sampling kernel
//   add_layernorm    out = LayerNorm(x + y)                                                2 reads, 1 write
//   bigate_sum       g = q + qi; s1 = sigmoid(g.wb + bb); s2 = sigmoid(g.wa + ba);
//                    q' = q + qi*s1; qi' = qi + q*s2                                       2 reads, 2 writes
// All HBM/L2-bandwidth bound; algorithmic bytes = rows * C * 4 * (reads + writes).
#include "common.h"

namespace df3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void actr_prep_kernel(const float *__restrict__ q, const float *__restrict__ qi,
                                                        const float *__restrict__ pos, size_t n4,
                                                        float *__restrict__ A, float *__restrict__ Bw) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  f32x4 a = ((const f32x4 *)q)[i], b = ((const f32x4 *)qi)[i], p = ((const f32x4 *)pos)[i];
  f32x4 lq = a + p;
  ((f32x4 *)A)[i] = lq;
  ((f32x4 *)Bw)[i] = lq + (b + p);          // (q + pos) + (qi + pos): the reference's association
}

// one wave per row; C <= 1024, C % 4 == 0
__global__ __launch_bounds__(256) void add_layernorm_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, float eps, long long rows,
                                                            int C, float *__restrict__ out) {
  long long row = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int nv = C / 4;
  f32x4 v[4];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int c4 = lane + 64 * k;
    v[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (c4 < nv) {
      f32x4 a = ((const f32x4 *)(x + row * C))[c4];
      if (y) a += ((const f32x4 *)(y + row * C))[c4];
      v[k] = a;
      s += a[0] + a[1] + a[2] + a[3];
    }
  }
  float mean = wave_sum(s) / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int c4 = lane + 64 * k;
    if (c4 < nv) {
      f32x4 d = v[k] - mean;
      ss += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
    }
  }
  float rstd = rsqrtf(wave_sum(ss) / (float)C + eps);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int c4 = lane + 64 * k;
    if (c4 < nv) {
      f32x4 g = ((const f32x4 *)gamma)[c4], b = ((const f32x4 *)beta)[c4];
      ((f32x4 *)(out + row * C))[c4] = (v[k] - mean) * rstd * g + b;
    }
  }
}

__global__ __launch_bounds__(256) void bigate_sum_kernel(const float *__restrict__ q, const float *__restrict__ qi,
                                                         const float *__restrict__ wb, const float *__restrict__ bb,
                                                         const float *__restrict__ wa, const float *__restrict__ ba,
                                                         long long rows, int C, float *__restrict__ qo,
                                                         float *__restrict__ qio) {
  long long row = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int nv = C / 4;
  f32x4 a[4], b[4];
  float d1 = 0.f, d2 = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int c4 = lane + 64 * k;
    a[k] = b[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (c4 < nv) {
      a[k] = ((const f32x4 *)(q + row * C))[c4];
      b[k] = ((const f32x4 *)(qi + row * C))[c4];
      f32x4 g = a[k] + b[k];
      f32x4 w1 = ((const f32x4 *)wb)[c4], w2 = ((const f32x4 *)wa)[c4];
      d1 += g[0] * w1[0] + g[1] * w1[1] + g[2] * w1[2] + g[3] * w1[3];
      d2 += g[0] * w2[0] + g[1] * w2[1] + g[2] * w2[2] + g[3] * w2[3];
    }
  }
  float s1 = 1.f / (1.f + __expf(-(wave_sum(d1) + bb[0])));
  float s2 = 1.f / (1.f + __expf(-(wave_sum(d2) + ba[0])));
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int c4 = lane + 64 * k;
    if (c4 < nv) {
      ((f32x4 *)(qo + row * C))[c4] = a[k] + b[k] * s1;
      ((f32x4 *)(qio + row * C))[c4] = b[k] + a[k] * s2;
    }
  }
}

struct MsdaFusedArgs {
  const float *value;
  const int64_t *shapes, *lstart;
  const float *ref;      // [N, Lq, 2] (x, y) in [0,1]
  const float *off;      // [N, Lq, M, L, P, 2] raw sampling offsets (pixels)
  const float *logit;    // [N, Lq, M, L*P] raw attention logits
  float *out;
  int N, S, M, D, Lq, L, P;
  long long vstride;     // floats between consecutive pixels of `value` (>= M*D)
};

// msda_vec4_kernel + in-kernel softmax over the L*P logits of a (query, head) and
// loc = ref + off / (W_l, H_l) (ms_deform_attn.py:149-166, reference_points with 2 coordinates).
template <int LPG>
__global__ __launch_bounds__(256) void msda_fused_kernel(MsdaFusedArgs a) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)a.N * a.Lq * a.M * LPG;
  const bool live = gid < total;
  const long long qm = live ? gid / LPG : (total - 1) / LPG;
  const int sub = (int)(gid % LPG);
  const int m = (int)(qm % a.M);
  const long long bq = qm / a.M;                 // b*Lq + q
  const int b = (int)(bq / a.Lq);
  const int LP = a.L * a.P;
  const float *off = a.off + (size_t)qm * LP * 2;
  const float *lg = a.logit + (size_t)qm * LP;
  const float rx = a.ref[bq * 2], ry = a.ref[bq * 2 + 1];
  const int lane = threadIdx.x & 63;
  const int gbase = lane & ~(LPG - 1);
  // softmax statistics over the group's logits
  float mx = -3.0e38f;
  for (int lp = sub; lp < LP; lp += LPG) mx = fmaxf(mx, lg[lp]);
#pragma unroll
  for (int o = LPG >> 1; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sm = 0.f;
  for (int lp = sub; lp < LP; lp += LPG) sm += __expf(lg[lp] - mx);
#pragma unroll
  for (int o = LPG >> 1; o >= 1; o >>= 1) sm += __shfl_xor(sm, o, 64);
  const float inv = 1.f / sm;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int lp0 = 0; lp0 < LP; lp0 += LPG) {
    float mxo = 0.f, myo = 0.f, mw = 0.f;
    int lp = lp0 + sub;
    if (lp < LP) {
      mxo = off[lp * 2];
      myo = off[lp * 2 + 1];
      mw = __expf(lg[lp] - mx) * inv;
    }
    const int cnt = (LP - lp0) < LPG ? (LP - lp0) : LPG;
    for (int i = 0; i < cnt; ++i) {
      float ox = __shfl(mxo, gbase + i, 64);
      float oy = __shfl(myo, gbase + i, 64);
      float w = __shfl(mw, gbase + i, 64);
      int l = (lp0 + i) / a.P;
      int H = (int)a.shapes[l * 2], W = (int)a.shapes[l * 2 + 1];
      float lx = rx + ox / (float)W;
      float ly = ry + oy / (float)H;
      const float *vbase = a.value + ((size_t)b * a.S + (size_t)a.lstart[l]) * a.vstride + m * a.D + sub * 4;
      float h_im = ly * (float)H - 0.5f;
      float w_im = lx * (float)W - 0.5f;
      if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
        int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        int h_high = h_low + 1, w_high = w_low + 1;
        float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
        float hh = 1.f - lh, hw = 1.f - lw;
        f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 v1 = z, v2 = z, v3 = z, v4 = z;
        const size_t hs = (size_t)W * a.vstride;
        if (h_low >= 0 && w_low >= 0) v1 = *(const f32x4 *)(vbase + h_low * hs + (size_t)w_low * a.vstride);
        if (h_low >= 0 && w_high <= W - 1) v2 = *(const f32x4 *)(vbase + h_low * hs + (size_t)w_high * a.vstride);
        if (h_high <= H - 1 && w_low >= 0) v3 = *(const f32x4 *)(vbase + h_high * hs + (size_t)w_low * a.vstride);
        if (h_high <= H - 1 && w_high <= W - 1) v4 = *(const f32x4 *)(vbase + h_high * hs + (size_t)w_high * a.vstride);
        float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        acc += (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4) * w;
      }
    }
  }
  if (live) *(f32x4 *)(a.out + (size_t)qm * a.D + sub * 4) = acc;
}

}  // namespace df3d

using namespace df3d;

extern "C" int df3d_actr_prep(const float *q, const float *qi, const float *pos, long long rows, int C, float *A,
                              float *Bw, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(q && qi && pos && A && Bw && C % 4 == 0, "actr_prep: bad arguments");
  size_t n4 = (size_t)rows * C / 4;
  if (n4 == 0) return DF3D_OK;
  hipLaunchKernelGGL(actr_prep_kernel, dim3(cdiv((long long)n4, 256)), dim3(256), 0, stream, q, qi, pos, n4, A, Bw);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_add_layernorm(const float *x, const float *y, const float *gamma, const float *beta, float eps,
                                  long long rows, int C, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(x && gamma && beta && out, "add_layernorm: null argument");
  DF3D_CHECK_ARG(C % 4 == 0 && C <= 1024, "add_layernorm: C must be a multiple of 4 and <= 1024 (got %d)", C);
  if (rows == 0) return DF3D_OK;
  hipLaunchKernelGGL(add_layernorm_kernel, dim3(cdiv(rows * 64, 256)), dim3(256), 0, stream, x, y, gamma, beta, eps,
                     rows, C, out);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_bigate_sum(const float *q, const float *qi, const float *wb, const float *bb, const float *wa,
                               const float *ba, long long rows, int C, float *q_out, float *qi_out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(q && qi && wb && bb && wa && ba && q_out && qi_out, "bigate_sum: null argument");
  DF3D_CHECK_ARG(C % 4 == 0 && C <= 1024, "bigate_sum: C must be a multiple of 4 and <= 1024 (got %d)", C);
  if (rows == 0) return DF3D_OK;
  hipLaunchKernelGGL(bigate_sum_kernel, dim3(cdiv(rows * 64, 256)), dim3(256), 0, stream, q, qi, wb, bb, wa, ba, rows,
                     C, q_out, qi_out);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_ms_deform_attn_fused(const float *value, long long value_stride, const int64_t *spatial_shapes,
                                         const int64_t *level_start_index, const float *ref_xy,
                                         const float *offsets, const float *logits, int N, int S, int M, int D, int Lq,
                                         int L, int P, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(value && spatial_shapes && level_start_index && ref_xy && offsets && logits && out,
                 "ms_deform_attn_fused: null argument");
  DF3D_CHECK_ARG(D % 4 == 0 && value_stride >= (long long)M * D && value_stride % 4 == 0,
                 "ms_deform_attn_fused: D %% 4 != 0 or bad value stride");
  if (N == 0 || Lq == 0) return DF3D_OK;
  MsdaFusedArgs a = {value, spatial_shapes, level_start_index, ref_xy, offsets, logits, out, N, S, M, D, Lq, L, P,
                     value_stride};
  long long total;
  switch (D / 4) {
#define DF3D_CASE(G)                                                                                \
  case G:                                                                                           \
    total = (long long)N * Lq * M * G;                                                              \
    hipLaunchKernelGGL(msda_fused_kernel<G>, dim3(cdiv(total, 256)), dim3(256), 0, stream, a);      \
    break;
    DF3D_CASE(1)
    DF3D_CASE(2)
    DF3D_CASE(4)
    DF3D_CASE(8)
    DF3D_CASE(16)
#undef DF3D_CASE
    default:
      set_error("ms_deform_attn_fused: head dim %d unsupported (need D/4 in {1,2,4,8,16})", D);
      return DF3D_EINVAL;
  }
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}
