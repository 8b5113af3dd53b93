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

DF3D_SPLIT_OVERFLOW_TU(actr)

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

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

// LPR lanes per row (16 / 32 / 64: the smallest that covers C / 4 vectors, so a 64-channel row keeps a quarter wave busy and a
// wave normalises four rows -- one wave per row left 48 of 64 lanes idle there: 113 us for 262 k rows); C <= 1024, C % 4 == 0
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int LPR>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, float eps, long long rows,
                                                            int C, float *__restrict__ out,
                                                            unsigned *__restrict__ out_split = nullptr) {
  constexpr int RPW = 64 / LPR, NK = LPR == 64 ? 4 : 1;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63, sub = lane % LPR;
  const long long row = wave * RPW + lane / LPR;
  const bool live = row < rows;
  const int nv = C / 4;
  f32x4 v[NK];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    int c4 = sub + LPR * k;
    v[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (live && c4 < nv) {
      f32x4 a = ((const f32x4 *)(x + row * C))[c4];
      if (y) a += ((const f32x4 *)(y + row * C))[c4];
      v[k] = a;
      s += a[0] + a[1] + a[2] + a[3];
    }
  }
  float mean = group_sum<LPR>(s) / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    int c4 = sub + LPR * k;
    if (c4 < nv) {
      f32x4 d = v[k] - mean;
      ss += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
    }
  }
  float rstd = rsqrtf(group_sum<LPR>(ss) / (float)C + eps);
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    int c4 = sub + LPR * k;
    if (live && c4 < nv) {
      f32x4 g = ((const f32x4 *)gamma)[c4], b = ((const f32x4 *)beta)[c4];
      const f32x4 o = (v[k] - mean) * rstd * g + b;
      ((f32x4 *)(out + row * C))[c4] = o;
      if (out_split) {       // split rows for the next conv / linear: per 8 channels 16 B of bf16 hi, 16 B of bf16 lo
        unsigned h0, l0, h1, l1;
        split_pair(o[0], o[1], h0, l0);
        split_pair(o[2], o[3], h1, l1);
        unsigned *p = out_split + (size_t)row * C + (c4 >> 1) * 8 + (c4 & 1) * 2;      // u32 units: 8 per 8 channels
        p[0] = h0, p[1] = h1, p[4] = l0, p[5] = l1;
      }
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
  const float *pscale;   // optional [N, S]: value(p) = pscale[p] * value[p] + ibias[n]
  const float *ibias;    // optional [N, M*D] rows, `bstride` floats apart
  long long bstride;
};

// msda_vec4_kernel + in-kernel softmax over the L*P logits of a (query, head) and
// loc = ref + off / (W_l, H_l) (ms_deform_attn.py:149-166, reference_points with 2 coordinates).
// VBF16: `value` holds bf16 rows (vstride counts bf16 elements); everything else stays fp32
template <int LPG, bool VBF16 = false>
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
  f32x4 cb = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (a.ibias) cb = *(const f32x4 *)(a.ibias + (size_t)b * a.bstride + m * a.D + sub * 4);
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
      const size_t pix0 = (size_t)b * a.S + (size_t)a.lstart[l];
      const size_t voff = pix0 * a.vstride + m * a.D + sub * 4;          // in elements of the value type
      const float *vbase = a.value + voff;
      const unsigned short *vbase16 = (const unsigned short *)a.value + voff;
      auto corner = [&](size_t p) -> f32x4 {
        if constexpr (VBF16) {
          const u32x2 t = *(const u32x2 *)(vbase16 + p * a.vstride);
          return (f32x4){__uint_as_float(t[0] << 16), __uint_as_float(t[0] & 0xffff0000u), __uint_as_float(t[1] << 16),
                         __uint_as_float(t[1] & 0xffff0000u)};
        } else {
          return *(const f32x4 *)(vbase + p * a.vstride);
        }
      };
      const float *sbase = a.pscale ? a.pscale + pix0 : nullptr;
      float h_im = ly * (float)H - 0.5f;
      float w_im = lx * (float)W - 0.5f;
      if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
        int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        int h_high = h_low + 1, w_high = w_low + 1;
        float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
        float hh = 1.f - lh, hw = 1.f - lw;
        f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 v1 = z, v2 = z, v3 = z, v4 = z;
        float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        const bool in1 = h_low >= 0 && w_low >= 0, in2 = h_low >= 0 && w_high <= W - 1;
        const bool in3 = h_high <= H - 1 && w_low >= 0, in4 = h_high <= H - 1 && w_high <= W - 1;
        const size_t p1 = (size_t)h_low * W + w_low, p2 = (size_t)h_low * W + w_high;
        const size_t p3 = (size_t)h_high * W + w_low, p4 = (size_t)h_high * W + w_high;
        if (in1) v1 = corner(p1);
        if (in2) v2 = corner(p2);
        if (in3) v3 = corner(p3);
        if (in4) v4 = corner(p4);
        if (sbase) {
          // value(p) = s_p * raw_p + c: the scale rides on the corner weight, the constant on the in-bounds weight sum
          float ws = (in1 ? w1 : 0.f) + (in2 ? w2 : 0.f) + (in3 ? w3 : 0.f) + (in4 ? w4 : 0.f);
          w1 *= in1 ? sbase[p1] : 0.f;
          w2 *= in2 ? sbase[p2] : 0.f;
          w3 *= in3 ? sbase[p3] : 0.f;
          w4 *= in4 ? sbase[p4] : 0.f;
          acc += (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4 + ws * cb) * w;
        } else if (a.ibias) {
          float ws = (in1 ? w1 : 0.f) + (in2 ? w2 : 0.f) + (in3 ? w3 : 0.f) + (in4 ? w4 : 0.f);
          acc += (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4 + ws * cb) * w;
        } else {
          acc += (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4) * w;
        }
      }
    }
  }
  if (live) *(f32x4 *)(a.out + (size_t)qm * a.D + sub * 4) = acc;
}

// Per (image, channel) sums over pixels of a_p * u_cp and (a_p * u_cp)^2 for channel-first u [N, C(+), S]
// (`cstride` floats between channels, `nstride` between images); a may be NULL (a_p = 1).
// One block per (channel, image): the GroupNorm statistics of x = a*u + b follow in closed form.
__global__ __launch_bounds__(256) void scaled_moments_kernel(const float *__restrict__ u, long long nstride,
                                                             long long cstride, const float *__restrict__ a, int S,
                                                             int C, double *__restrict__ out) {
  const int c = blockIdx.x, n = blockIdx.y;
  const float *row = u + (size_t)n * nstride + (size_t)c * cstride;
  const float *ar = a ? a + (size_t)n * S : nullptr;
  // 16-byte loads of u over the aligned body of the row (rows start at arbitrary 4-byte offsets), scalar
  // head/tail; fp32 partials are folded into doubles every 8 vectors
  const size_t off = ((size_t)(uintptr_t)row >> 2) & 3;     // dwords past a 16-byte boundary
  const int head = (int)((4 - off) & 3) < S ? (int)((4 - off) & 3) : S;
  const int nb = (S - head) / 4;
  double d1 = 0.0, d2 = 0.0;
  float s1 = 0.f, s2 = 0.f;
  int it = 0;
  for (int i = threadIdx.x; i < nb; i += 256) {
    const int p = head + 4 * i;
    f32x4 v = *(const f32x4 *)(row + p);
    if (ar) {
      v[0] *= ar[p];
      v[1] *= ar[p + 1];
      v[2] *= ar[p + 2];
      v[3] *= ar[p + 3];
    }
    s1 += (v[0] + v[1]) + (v[2] + v[3]);
    s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    if (++it == 8) {
      d1 += s1; d2 += s2; s1 = s2 = 0.f; it = 0;
    }
  }
  {                                                      // scalar head (< 4 elements) and tail (< 4 elements)
    const int tail0 = head + 4 * nb;
    int p = -1;
    if ((int)threadIdx.x < head) p = threadIdx.x;
    else if ((int)threadIdx.x - head < S - tail0) p = tail0 + (int)threadIdx.x - head;
    if (p >= 0) {
      float v = row[p] * (ar ? ar[p] : 1.f);
      s1 += v;
      s2 += v * v;
    }
  }
  d1 += s1; d2 += s2;
  __shared__ double sh[2][256];
  sh[0][threadIdx.x] = d1;
  sh[1][threadIdx.x] = d2;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[((size_t)n * C + c) * 2] = sh[0][0];
    out[((size_t)n * C + c) * 2 + 1] = sh[1][0];
  }
}

// GroupNorm(x = a*u + b) followed by O linear outputs, folded per image:
//   y_c = a*u_c*s_c + t_c, s_c = rstd_g*gamma_c, t_c = (b_c - mean_g)*rstd_g*gamma_c + beta_c
//   out = a * (Wf u) + cf, Wf[o,c] = W[o,c]*s_c, cf[o] = sum_c W[o,c]*t_c + wb[o]
// One block per image, 256 threads; C <= 256.
__global__ __launch_bounds__(256) void gn_fold_kernel(const double *__restrict__ mom, const float *__restrict__ b,
                                                      const float *__restrict__ gamma, const float *__restrict__ beta,
                                                      float eps, int S, int C, int groups,
                                                      const float *__restrict__ W, const float *__restrict__ wb, int O,
                                                      float *__restrict__ Wf, float *__restrict__ cf) {
  const int n = blockIdx.x, tid = threadIdx.x;
  __shared__ double m1[256], m2[256];
  __shared__ float sc[256], tc[256];
  if (tid < C) {
    double su = mom[((size_t)n * C + tid) * 2], sq = mom[((size_t)n * C + tid) * 2 + 1];
    double bc = b ? (double)b[tid] : 0.0;
    m1[tid] = su + S * bc;                                // sum_p x_c
    m2[tid] = sq + 2.0 * bc * su + S * bc * bc;           // sum_p x_c^2
  }
  __syncthreads();
  const int cpg = C / groups;
  if (tid < C) {
    int g = tid / cpg;
    double s = 0.0, q = 0.0;
    for (int k = 0; k < cpg; ++k) { s += m1[g * cpg + k]; q += m2[g * cpg + k]; }
    double cnt = (double)cpg * S;
    double mean = s / cnt, var = q / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    float rstd = (float)(1.0 / sqrt(var + (double)eps));
    float bc = b ? b[tid] : 0.f;
    sc[tid] = rstd * gamma[tid];
    tc[tid] = (bc - (float)mean) * rstd * gamma[tid] + beta[tid];
  }
  __syncthreads();
  for (int o = tid; o < O; o += 256) {
    float acc = wb ? wb[o] : 0.f;
    for (int c = 0; c < C; ++c) {
      float w = W[(size_t)o * C + c];
      Wf[((size_t)n * O + o) * C + c] = w * sc[c];
      acc += w * tc[c];
    }
    cf[(size_t)n * O + o] = acc;
  }
}

// GroupNorm over [N, Q, C] rows (the reference normalises the Conv1d output [N, C, Q], actr.py:150-158: statistics per
// image n and group over Q x C/groups values) without the two transposes: pass 1 accumulates per-(n, group) sums
// with one 16-byte quad of channels per lane (a row is read as one contiguous 4C-byte segment), pass 2 normalises.
__global__ __launch_bounds__(256) void rows_gn_stats_kernel(const float *__restrict__ x, int Q, int C, int cpg,
                                                            int rows_per_block, double *__restrict__ stats) {
  const int n = blockIdx.y;
  const int nq = C / 4;                      // quads per row
  const int q4 = threadIdx.x % nq, rsub = threadIdx.x / nq, rstep = blockDim.x / nq;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(r0 + rows_per_block, Q);
  float s1 = 0.f, s2 = 0.f;
  if (rsub < rstep)
    for (int r = r0 + rsub; r < r1; r += rstep) {
      f32x4 v = *(const f32x4 *)(x + ((size_t)n * Q + r) * C + q4 * 4);
      s1 += (v[0] + v[1]) + (v[2] + v[3]);
      s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
  __shared__ float sh1[256], sh2[256];
  sh1[threadIdx.x] = s1;
  sh2[threadIdx.x] = s2;
  __syncthreads();
  const int groups = C / cpg, qpg = cpg / 4;  // quads per group
  if ((int)threadIdx.x < groups) {
    double a = 0.0, b = 0.0;
    for (int rs = 0; rs < rstep; ++rs)
      for (int k = 0; k < qpg; ++k) {
        a += sh1[rs * nq + threadIdx.x * qpg + k];
        b += sh2[rs * nq + threadIdx.x * qpg + k];
      }
    atomicAdd(&stats[((size_t)n * groups + threadIdx.x) * 2], a);
    atomicAdd(&stats[((size_t)n * groups + threadIdx.x) * 2 + 1], b);
  }
}

__global__ __launch_bounds__(256) void rows_gn_apply_kernel(const float *__restrict__ x, const double *__restrict__ stats,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, float eps, int Q, int C,
                                                            int cpg, size_t n4, float *__restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int nq = C / 4;
  const int q4 = (int)(i % nq);
  const size_t row = i / nq;
  const int n = (int)(row / Q);
  const int g = q4 * 4 / cpg, groups = C / cpg;
  const double cnt = (double)Q * cpg;
  const double m = stats[((size_t)n * groups + g) * 2] / cnt;
  double var = stats[((size_t)n * groups + g) * 2 + 1] / cnt - m * m;
  if (var < 0.0) var = 0.0;
  const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)eps));
  f32x4 v = ((const f32x4 *)x)[i];
  f32x4 ga = ((const f32x4 *)gamma)[q4], be = ((const f32x4 *)beta)[q4];
  ((f32x4 *)out)[i] = (v - mean) * rstd * ga + be;
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

static int add_layernorm_impl(const float *x, const float *y, const float *gamma, const float *beta, float eps,
                                  long long rows, int C, float *out, unsigned *out_split, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(x && gamma && beta && out, "add_layernorm: null argument");
  DF3D_CHECK_ARG(C % 4 == 0 && C <= 1024, "add_layernorm: C must be a multiple of 4 and <= 1024 (got %d)", C);
  if (rows == 0) return DF3D_OK;
  if (C <= 64)
    hipLaunchKernelGGL(add_layernorm_kernel<16>, dim3(cdiv(cdiv(rows, 4) * 64, 256)), dim3(256), 0, stream, x, y, gamma, beta,
                       eps, rows, C, out, out_split);
  else if (C <= 128)
    hipLaunchKernelGGL(add_layernorm_kernel<32>, dim3(cdiv(cdiv(rows, 2) * 64, 256)), dim3(256), 0, stream, x, y, gamma, beta,
                       eps, rows, C, out, out_split);
  else
    hipLaunchKernelGGL(add_layernorm_kernel<64>, dim3(cdiv(rows * 64, 256)), dim3(256), 0, stream, x, y, gamma, beta, eps,
                       rows, C, out, out_split);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_add_layernorm(const float *x, const float *y, const float *gamma, const float *beta, float eps,
                                  long long rows, int C, float *out, void *stream_) {
  return add_layernorm_impl(x, y, gamma, beta, eps, rows, C, out, nullptr, stream_);
}

extern "C" int df3d_add_layernorm_split(const float *x, const float *y, const float *gamma, const float *beta, float eps,
                                        long long rows, int C, float *out, void *out_split, void *stream_) {
  DF3D_CHECK_ARG(out_split && C % 8 == 0, "add_layernorm_split: split rows need a multiple of 8 channels (got %d)", C);
  return add_layernorm_impl(x, y, gamma, beta, eps, rows, C, out, (unsigned *)out_split, stream_);
}

// ---- training: ReLU + dropout of the feed-forward hidden rows in one pass, in place ------------------------------------------------
// forward_ffn (CP/det3d/models/model_utils/actr_transformer.py:309-311, 388-395) runs linear2(dropout(relu(linear1(src)))): on the
// [rows x 1024] hidden tensor of a training step (150 k - 240 k rows: 0.6 - 1 GB) the library's relu + dropout read and write it
// twice and keep a mask, their backward twice more.  Here h <- relu(h) . keep / (1 - p) in ONE in-place pass; the kept-and-active
// elements are exactly the non-zeros of the result, so the backward (g <- g / (1 - p) where h != 0) needs no mask tensor.
// keep = a counter-based hash of (element index, the call's 64-bit seed): the same mask whatever the launch shape.
// (rd_keep: common.h)
__global__ __launch_bounds__(256) void relu_dropout_kernel(float *__restrict__ h, unsigned long long n, unsigned thr, float scale,
                                                           unsigned s0, unsigned s1) {
  const unsigned long long i4 = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 + 3 < n) {
    f32x4 v = *(f32x4 *)(h + i4);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f && (thr == 0u || rd_keep(i4 + j, s0, s1, thr)) ? v[j] * scale : 0.f;
    *(f32x4 *)(h + i4) = v;
  } else {
    for (unsigned long long i = i4; i < n; ++i) {
      const float v = h[i];
      h[i] = v > 0.f && (thr == 0u || rd_keep(i, s0, s1, thr)) ? v * scale : 0.f;
    }
  }
}
__global__ __launch_bounds__(256) void relu_dropout_bwd_kernel(const float *__restrict__ h, const float *__restrict__ g,
                                                               unsigned long long n, float scale, float *__restrict__ out) {
  const unsigned long long i4 = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 + 3 < n) {
    const f32x4 hv = *(const f32x4 *)(h + i4), gv = *(const f32x4 *)(g + i4);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = hv[j] != 0.f ? gv[j] * scale : 0.f;
    *(f32x4 *)(out + i4) = o;
  } else {
    for (unsigned long long i = i4; i < n; ++i) out[i] = h[i] != 0.f ? g[i] * scale : 0.f;
  }
}

// bfloat16 rows (the bf16 mixed-precision mode): eight elements per lane
__device__ __forceinline__ float rd_bf16_to_f32(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ unsigned short rd_f32_to_bf16(float f) {       // round to nearest even (finite inputs)
  const unsigned u = __float_as_uint(f);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
template <bool BWD>
__global__ __launch_bounds__(256) void relu_dropout_bf16_kernel(unsigned short *__restrict__ h, const unsigned short *__restrict__ g,
                                                                unsigned short *__restrict__ out, unsigned long long n,
                                                                unsigned thr, float scale, unsigned s0, unsigned s1) {
  const unsigned long long i8 = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i8 >= n) return;
  const int cnt = (int)min((unsigned long long)8, n - i8);
  unsigned short hv[8], gv[8], ov[8];
  if (cnt == 8) {
    *(u32x4 *)hv = *(const u32x4 *)(h + i8);
    if (BWD) *(u32x4 *)gv = *(const u32x4 *)(g + i8);
  } else {
    for (int j = 0; j < cnt; ++j) hv[j] = h[i8 + j], gv[j] = BWD ? g[i8 + j] : (unsigned short)0;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (BWD) {
      ov[j] = (hv[j] & 0x7fffu) ? rd_f32_to_bf16(rd_bf16_to_f32(gv[j]) * scale) : (unsigned short)0;
    } else {
      const float v = rd_bf16_to_f32(hv[j]);
      ov[j] = v > 0.f && (thr == 0u || rd_keep(i8 + j, s0, s1, thr)) ? rd_f32_to_bf16(v * scale) : (unsigned short)0;
    }
  }
  unsigned short *dst = BWD ? out : h;
  if (cnt == 8) *(u32x4 *)(dst + i8) = *(const u32x4 *)ov;
  else
    for (int j = 0; j < cnt; ++j) dst[i8 + j] = ov[j];
}

extern "C" int df3d_relu_dropout_bf16(void *h, long long n, float p, unsigned long long seed, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(n >= 0 && p >= 0.f && p < 1.f, "relu_dropout_bf16: p must be in [0, 1) (got %g)", (double)p);
  if (n == 0) return DF3D_OK;
  DF3D_CHECK_ARG(h && ((uintptr_t)h & 15) == 0, "relu_dropout_bf16: null or unaligned rows");
  const unsigned thr = (unsigned)((double)p * 16777216.0);
  hipLaunchKernelGGL(relu_dropout_bf16_kernel<false>, dim3(cdiv(cdiv(n, 8), 256)), dim3(256), 0, stream, (unsigned short *)h, nullptr,
                     nullptr, (unsigned long long)n, thr, thr ? (float)(1.0 / (1.0 - (double)thr / 16777216.0)) : 1.f, (unsigned)seed,
                     (unsigned)(seed >> 32));
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_relu_dropout_backward_bf16(const void *h, const void *grad, long long n, float p, void *grad_in, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(n >= 0 && p >= 0.f && p < 1.f, "relu_dropout_backward_bf16: p must be in [0, 1) (got %g)", (double)p);
  if (n == 0) return DF3D_OK;
  DF3D_CHECK_ARG(h && grad && grad_in && (((uintptr_t)h | (uintptr_t)grad | (uintptr_t)grad_in) & 15) == 0,
                 "relu_dropout_backward_bf16: null or unaligned rows");
  const unsigned thr = (unsigned)((double)p * 16777216.0);
  hipLaunchKernelGGL(relu_dropout_bf16_kernel<true>, dim3(cdiv(cdiv(n, 8), 256)), dim3(256), 0, stream, (unsigned short *)h,
                     (const unsigned short *)grad, (unsigned short *)grad_in, (unsigned long long)n, thr,
                     thr ? (float)(1.0 / (1.0 - (double)thr / 16777216.0)) : 1.f, 0u, 0u);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_relu_dropout(float *h, long long n, float p, unsigned long long seed, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(n >= 0 && p >= 0.f && p < 1.f, "relu_dropout: p must be in [0, 1) (got %g)", (double)p);
  if (n == 0) return DF3D_OK;
  DF3D_CHECK_ARG(h && ((uintptr_t)h & 15) == 0, "relu_dropout: null or unaligned rows");
  const unsigned thr = (unsigned)((double)p * 16777216.0);
  hipLaunchKernelGGL(relu_dropout_kernel, dim3(cdiv(cdiv(n, 4), 256)), dim3(256), 0, stream, h, (unsigned long long)n, thr,
                     thr ? (float)(1.0 / (1.0 - (double)thr / 16777216.0)) : 1.f, (unsigned)seed, (unsigned)(seed >> 32));
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_relu_dropout_backward(const float *h, const float *grad, long long n, float p, float *grad_in, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(n >= 0 && p >= 0.f && p < 1.f, "relu_dropout_backward: p must be in [0, 1) (got %g)", (double)p);
  if (n == 0) return DF3D_OK;
  DF3D_CHECK_ARG(h && grad && grad_in && (((uintptr_t)h | (uintptr_t)grad | (uintptr_t)grad_in) & 15) == 0,
                 "relu_dropout_backward: null or unaligned rows");
  const unsigned thr = (unsigned)((double)p * 16777216.0);
  hipLaunchKernelGGL(relu_dropout_bwd_kernel, dim3(cdiv(cdiv(n, 4), 256)), dim3(256), 0, stream, h, grad, (unsigned long long)n,
                     thr ? (float)(1.0 / (1.0 - (double)thr / 16777216.0)) : 1.f, grad_in);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

// ---- training: LayerNorm(x + dropout(y)) and its backward as one row kernel each way ------------------------------------------
// The residual steps of the encoder layers (actr_transformer.py:311-312, 330-331, 389-390, 395-396, 416-417: src = norm(src +
// dropout(src2))) ran as dropout (+ mask) + add + LayerNorm (7 passes over the rows) and, backwards, LayerNorm's two kernels +
// masked scale + the residual's gradient sum (8 passes).  Forward here: out and the normalised rows xhat (what the backward
// needs instead of the input) + 1 / sigma per row; backward: dx = (g gamma - mean(g gamma) - xhat mean(g gamma xhat)) / sigma for
// the residual branch, dy = dx . keep / (1 - p) with the mask recomputed from the hash, d gamma / d beta as per-workgroup
// column sums added once per workgroup.
template <int LPR>
__global__ __launch_bounds__(256) void dropout_add_layernorm_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                                    const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                    float eps, long long rows, int C, unsigned thr, float scale,
                                                                    unsigned s0, unsigned s1, float *__restrict__ out,
                                                                    float *__restrict__ xhat, float *__restrict__ rstd_out) {
  constexpr int RPW = 64 / LPR, NK = LPR == 64 ? 4 : 1;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63, sub = lane % LPR;
  const long long row = wave * RPW + lane / LPR;
  const bool live = row < rows;
  const int nv = C / 4;
  f32x4 v[NK];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int c4 = sub + LPR * k;
    v[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (live && c4 < nv) {
      f32x4 a = ((const f32x4 *)(x + row * C))[c4];
      f32x4 b = ((const f32x4 *)(y + row * C))[c4];
      if (thr) {
        const unsigned long long i0 = (unsigned long long)row * C + c4 * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = rd_keep(i0 + j, s0, s1, thr) ? b[j] * scale : 0.f;
      }
      a += b;
      v[k] = a;
      s += a[0] + a[1] + a[2] + a[3];
    }
  }
  const float mean = group_sum<LPR>(s) / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int c4 = sub + LPR * k;
    if (c4 < nv) {
      const f32x4 d = v[k] - mean;
      ss += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
    }
  }
  const float rstd = rsqrtf(group_sum<LPR>(ss) / (float)C + eps);
  if (live && sub == 0) rstd_out[row] = rstd;
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int c4 = sub + LPR * k;
    if (live && c4 < nv) {
      const f32x4 g = ((const f32x4 *)gamma)[c4], b = ((const f32x4 *)beta)[c4];
      const f32x4 xh = (v[k] - mean) * rstd;
      ((f32x4 *)(xhat + row * C))[c4] = xh;
      ((f32x4 *)(out + row * C))[c4] = xh * g + b;
    }
  }
}

template <int LPR>
__global__ __launch_bounds__(256) void dropout_add_layernorm_bwd_kernel(const float *__restrict__ grad, const float *__restrict__ xhat,
                                                                        const float *__restrict__ rstd,
                                                                        const float *__restrict__ gamma, long long rows, int C,
                                                                        unsigned thr, float scale, unsigned s0, unsigned s1,
                                                                        float *__restrict__ dx, float *__restrict__ dy,
                                                                        float *__restrict__ dgamma, float *__restrict__ dbeta) {
  constexpr int RPW = 64 / LPR, NK = LPR == 64 ? 4 : 1;
  __shared__ float sdg[1024], sdb[1024];
  for (int c = threadIdx.x; c < C; c += 256) sdg[c] = 0.f, sdb[c] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, sub = lane % LPR;
  const int nv = C / 4;
  const long long wave0 = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (long long)gridDim.x * 4;
  f32x4 gam[NK], ag[NK], ab[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int c4 = sub + LPR * k;
    gam[k] = c4 < nv ? ((const f32x4 *)gamma)[c4] : (f32x4){0.f, 0.f, 0.f, 0.f};
    ag[k] = ab[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (long long wave = wave0; wave * RPW < rows; wave += nwaves) {
    const long long row = wave * RPW + lane / LPR;
    const bool live = row < rows;
    f32x4 g[NK], xh[NK];
    float s1v = 0.f, s2v = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int c4 = sub + LPR * k;
      g[k] = xh[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (live && c4 < nv) {
        g[k] = ((const f32x4 *)(grad + row * C))[c4];
        xh[k] = ((const f32x4 *)(xhat + row * C))[c4];
        ab[k] += g[k];
        ag[k] += g[k] * xh[k];
        const f32x4 a = g[k] * gam[k];
        s1v += a[0] + a[1] + a[2] + a[3];
        s2v += a[0] * xh[k][0] + a[1] * xh[k][1] + a[2] * xh[k][2] + a[3] * xh[k][3];
      }
    }
    const float m1 = group_sum<LPR>(s1v) / (float)C, m2 = group_sum<LPR>(s2v) / (float)C;
    const float rs = live ? rstd[row] : 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int c4 = sub + LPR * k;
      if (live && c4 < nv) {
        const f32x4 d = (g[k] * gam[k] - m1 - xh[k] * m2) * rs;
        ((f32x4 *)(dx + row * C))[c4] = d;
        if (thr) {
          const unsigned long long i0 = (unsigned long long)row * C + c4 * 4;
          f32x4 e;
#pragma unroll
          for (int j = 0; j < 4; ++j) e[j] = rd_keep(i0 + j, s0, s1, thr) ? d[j] * scale : 0.f;
          ((f32x4 *)(dy + row * C))[c4] = e;
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int c4 = sub + LPR * k;
    if (c4 < nv)
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(&sdg[c4 * 4 + j], ag[k][j]), atomicAdd(&sdb[c4 * 4 + j], ab[k][j]);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) unsafeAtomicAdd(dgamma + c, sdg[c]), unsafeAtomicAdd(dbeta + c, sdb[c]);
}

static unsigned dropout_threshold(float p) { return (unsigned)((double)p * 16777216.0); }
static float dropout_scale(unsigned thr) { return thr ? (float)(1.0 / (1.0 - (double)thr / 16777216.0)) : 1.f; }

extern "C" int df3d_dropout_add_layernorm(const float *x, const float *y, const float *gamma, const float *beta, float eps, float p,
                                          unsigned long long seed, long long rows, int C, float *out, float *xhat, float *rstd,
                                          void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(rows >= 0 && C > 0 && C % 4 == 0 && C <= 1024, "dropout_add_layernorm: C must be a multiple of 4 and <= 1024 (got %d)", C);
  DF3D_CHECK_ARG(p >= 0.f && p < 1.f, "dropout_add_layernorm: p must be in [0, 1) (got %g)", (double)p);
  if (rows == 0) return DF3D_OK;
  DF3D_CHECK_ARG(x && y && gamma && beta && out && xhat && rstd, "dropout_add_layernorm: null argument");
  const unsigned thr = dropout_threshold(p), s0 = (unsigned)seed, s1 = (unsigned)(seed >> 32);
  const float sc = dropout_scale(thr);
#define DF3D_DAL(LPR, RPWV)                                                                                                   \
  hipLaunchKernelGGL(dropout_add_layernorm_kernel<LPR>, dim3(cdiv(cdiv(rows, RPWV) * 64, 256)), dim3(256), 0, stream, x, y, gamma, \
                     beta, eps, rows, C, thr, sc, s0, s1, out, xhat, rstd)
  if (C <= 64) DF3D_DAL(16, 4);
  else if (C <= 128) DF3D_DAL(32, 2);
  else DF3D_DAL(64, 1);
#undef DF3D_DAL
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

/* dgamma / dbeta [C] are ADDED to (zero them first) */
extern "C" int df3d_dropout_add_layernorm_backward(const float *grad, const float *xhat, const float *rstd, const float *gamma,
                                                   float p, unsigned long long seed, long long rows, int C, float *dx, float *dy,
                                                   float *dgamma, float *dbeta, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(rows >= 0 && C > 0 && C % 4 == 0 && C <= 1024, "dropout_add_layernorm_backward: C must be a multiple of 4 and <= 1024 (got %d)", C);
  DF3D_CHECK_ARG(p >= 0.f && p < 1.f, "dropout_add_layernorm_backward: p must be in [0, 1) (got %g)", (double)p);
  if (rows == 0) return DF3D_OK;
  const unsigned thr = dropout_threshold(p), s0 = (unsigned)seed, s1 = (unsigned)(seed >> 32);
  DF3D_CHECK_ARG(grad && xhat && rstd && gamma && dx && dgamma && dbeta && (dy || !thr), "dropout_add_layernorm_backward: null argument");
  const float sc = dropout_scale(thr);
#define DF3D_DALB(LPR, RPWV)                                                                                                       \
  hipLaunchKernelGGL(dropout_add_layernorm_bwd_kernel<LPR>, dim3((unsigned)std::min<long long>(2048, cdiv(cdiv(rows, RPWV), 4))), dim3(256), \
                     0, stream, grad, xhat, rstd, gamma, rows, C, thr, sc, s0, s1, dx, dy, dgamma, dbeta)
  if (C <= 64) DF3D_DALB(16, 4);
  else if (C <= 128) DF3D_DALB(32, 2);
  else DF3D_DALB(64, 1);
#undef DF3D_DALB
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

static int msda_fused_launch(const void *value, bool vbf16, long long value_stride, const int64_t *spatial_shapes,
                             const int64_t *level_start_index, const float *ref_xy, const float *offsets,
                             const float *logits, const float *pixel_scale, const float *image_bias, long long bias_stride,
                             int N, int S, int M, int D, int Lq, int L, int P, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(value && spatial_shapes && level_start_index && ref_xy && offsets && logits && out,
                 "ms_deform_attn_fused: null argument");
  DF3D_CHECK_ARG(D % 4 == 0 && value_stride >= (long long)M * D && value_stride % 4 == 0,
                 "ms_deform_attn_fused: D %% 4 != 0 or bad value stride");
  if (N == 0 || Lq == 0) return DF3D_OK;
  DF3D_CHECK_ARG(!pixel_scale || image_bias, "ms_deform_attn_fused: pixel_scale needs image_bias");
  DF3D_CHECK_ARG(!image_bias || bias_stride % 4 == 0, "ms_deform_attn_fused: bias stride must be a multiple of 4");
  MsdaFusedArgs a = {(const float *)value, spatial_shapes, level_start_index, ref_xy, offsets, logits, out, N, S, M, D, Lq, L, P,
                     value_stride, pixel_scale, image_bias, bias_stride};
  long long total;
  switch (D / 4) {
#define DF3D_CASE(G)                                                                                       \
  case G:                                                                                                  \
    total = (long long)N * Lq * M * G;                                                                     \
    if (vbf16) hipLaunchKernelGGL((msda_fused_kernel<G, true>), dim3(cdiv(total, 256)), dim3(256), 0, stream, a);  \
    else hipLaunchKernelGGL((msda_fused_kernel<G, false>), dim3(cdiv(total, 256)), dim3(256), 0, stream, a);       \
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

extern "C" int df3d_scaled_moments(const float *u, long long image_stride, long long channel_stride, const float *a,
                                   int N, int S, int C, double *moments, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(u && moments, "scaled_moments: null argument");
  if (N == 0 || C == 0) return DF3D_OK;
  hipLaunchKernelGGL(scaled_moments_kernel, dim3(C, N), dim3(256), 0, stream, u, image_stride, channel_stride, a, S, C,
                     moments);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_groupnorm_fold(const double *moments, const float *b, const float *gamma, const float *beta,
                                   float eps, int N, int S, int C, int groups, const float *W, const float *wb, int O,
                                   float *Wf, float *cf, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(moments && gamma && beta && W && Wf && cf, "groupnorm_fold: null argument");
  DF3D_CHECK_ARG(C <= 256 && groups > 0 && C % groups == 0, "groupnorm_fold: need C <= 256 and C %% groups == 0");
  if (N == 0) return DF3D_OK;
  hipLaunchKernelGGL(gn_fold_kernel, dim3(N), dim3(256), 0, stream, moments, b, gamma, beta, eps, S, C, groups, W, wb,
                     O, Wf, cf);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_rows_groupnorm(const float *x, int N, int Q, int C, int groups, const float *gamma,
                                   const float *beta, float eps, double *stats, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(x && gamma && beta && stats && out, "rows_groupnorm: null argument");
  DF3D_CHECK_ARG(groups > 0 && C % groups == 0 && (C / groups) % 4 == 0 && C <= 1024 && groups <= 256,
                 "rows_groupnorm: needs C %% groups == 0 and a group width that is a multiple of 4 (C=%d, groups=%d)",
                 C, groups);
  if (N == 0 || Q == 0) return DF3D_OK;
  DF3D_HIP(hipMemsetAsync(stats, 0, (size_t)N * groups * 2 * sizeof(double), stream));
  const int rows_per_block = 256;
  hipLaunchKernelGGL(rows_gn_stats_kernel, dim3(cdiv(Q, rows_per_block), N), dim3(256), 0, stream, x, Q, C, C / groups,
                     rows_per_block, stats);
  size_t n4 = (size_t)N * Q * C / 4;
  hipLaunchKernelGGL(rows_gn_apply_kernel, dim3(cdiv((long long)n4, 256)), dim3(256), 0, stream, x, stats, gamma, beta,
                     eps, Q, C, C / groups, n4, out);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_ms_deform_attn_fused(const float *value, long long value_stride, const int64_t *spatial_shapes,
                                         const int64_t *level_start_index, const float *ref_xy,
                                         const float *offsets, const float *logits, const float *pixel_scale,
                                         const float *image_bias, long long bias_stride, int N, int S, int M, int D,
                                         int Lq, int L, int P, float *out, void *stream_) {
  return msda_fused_launch(value, false, value_stride, spatial_shapes, level_start_index, ref_xy, offsets, logits,
                           pixel_scale, image_bias, bias_stride, N, S, M, D, Lq, L, P, out, stream_);
}

extern "C" int df3d_ms_deform_attn_fused_bf16(const void *value_bf16, long long value_stride,
                                              const int64_t *spatial_shapes, const int64_t *level_start_index,
                                              const float *ref_xy, const float *offsets, const float *logits,
                                              const float *pixel_scale, const float *image_bias, long long bias_stride,
                                              int N, int S, int M, int D, int Lq, int L, int P, float *out, void *stream_) {
  return msda_fused_launch(value_bf16, true, value_stride, spatial_shapes, level_start_index, ref_xy, offsets, logits,
                           pixel_scale, image_bias, bias_stride, N, S, M, D, Lq, L, P, out, stream_);
}
