// Cross-attention of a few hundred object queries over the whole BEV map (TransFusionHead decoder layer,
// TF/mmdet3d/models/dense_heads/transfusion_head.py:110-113 -> multi_head_attention_forward :255-505):
//   out[b, q, h*16:(h+1)*16] = softmax_k( scale * <Q[b,q,h], K[b,k,h]> ) . V[b,k,h]     200 queries x 32 400 keys x 8 heads
//
// The reference materialises the [B*8, 200, 32400] score tensor (bmm, softmax, bmm: 3 passes over 207 MB per sample);
// a fused-attention library kernel parallelises over QUERY blocks, which leaves 16 workgroups for 200 queries, each
// walking all 32 400 keys (measured ~1 ms).  Here the KEYS are split ("flash decoding"):
//   xattn_partial   grid (key chunks, heads, samples x query blocks); a wave owns up to 4 tiles of 16 queries (Q in
//                   registers) and walks the chunk in tiles of 16 keys:  S^T = K.Q^T (4 x v_mfma_f32_16x16x4_f32, the
//                   head dimension 16 is the contraction), running maximum per query (two cross-lane steps: with the
//                   transposed scores a lane holds 4 keys of ONE query), P^T = exp2(S^T - m), O^T += V^T.P^T (4 MFMA);
//                   two key tiles share one softmax step (one max / rescale per 32 keys).
//                   K / V rows are read once per wave straight from the [pixels, 2E] projection rows (64 B per head).
//                   Writes (m, l, O) per (chunk, query).
//   xattn_combine   log-sum-exp merge of the chunks.
// fp32 throughout; scale * log2(e) is folded into Q so the exponentials are single v_exp_f32.
#include <algorithm>
#include <cstring>

#include "common.h"

namespace df3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct XAttnArgs {
  const float *q, *k, *v;
  int ld_q, ld_k, ld_v;
  int batch, nq, nk, heads;
  float qscale;              // softmax scale * log2(e)
  int tiles_per_chunk, nchunks, qblocks;
  float *po;                 // [B][heads][nchunks][nq_pad][16]
  float *pml;                // [B][heads][nchunks][nq_pad][2]
  int nq_pad;
  // training (df3d_cross_attention_train): dropout of the probabilities (thr = p * 2^24, 0 = none; kept ones x dscale; element
  // index ((b * heads + h) * nq + q) * nk + key under the seed words s0 / s1) and the base-2 log-sum-exp per (b, h, q)
  unsigned thr, s0, s1;
  float dscale;
  float *lse;
};

constexpr int XQ = 4;        // query tiles per wave

struct KVTile {
  float4 kf;
  float vf[4];
};

__device__ __forceinline__ KVTile load_kv(const float *kb, const float *vb, int ld_k, int ld_v, int kt, int nk, int j, int g) {
  KVTile t;
  const int key0 = kt * 16;
  const int krow = min(key0 + j, nk - 1);                       // clamped rows are masked by the caller
  t.kf = *(const float4 *)(kb + (size_t)krow * ld_k + 4 * g);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int key = key0 + 4 * g + r;
    t.vf[r] = key < nk ? vb[(size_t)key * ld_v + j] : 0.f;
  }
  return t;
}

// NT query tiles of this wave, fully unrolled: NT independent MFMA chains per stage
template <int NT>
__device__ __forceinline__ void xattn_wave(const XAttnArgs &a, int b, int h, int chunk, int tile0, int t0, int t1, int j,
                                           int g) {
  f32x4 qf[NT], o[NT];
  float m[NT], ls[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int q = (tile0 + 4 * t) * 16 + j;
    qf[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (q < a.nq) {
      const float4 x = *(const float4 *)(a.q + ((size_t)b * a.nq + q) * a.ld_q + h * 16 + 4 * g);
      qf[t] = (f32x4){x.x * a.qscale, x.y * a.qscale, x.z * a.qscale, x.w * a.qscale};
    }
    o[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    m[t] = -INFINITY;
    ls[t] = 0.f;
  }
  const float *kb = a.k + (size_t)b * a.nk * a.ld_k + h * 16, *vb = a.v + (size_t)b * a.nk * a.ld_v + h * 16;
  // two key tiles (32 keys) per softmax step: one running-max update, one rescale and two cross-lane steps per 32 keys
  KVTile c0 = load_kv(kb, vb, a.ld_k, a.ld_v, t0, a.nk, j, g);
  KVTile c1 = load_kv(kb, vb, a.ld_k, a.ld_v, min(t0 + 1, t1 - 1), a.nk, j, g);
  for (int kt = t0; kt < t1; kt += 2) {
    const KVTile n0 = load_kv(kb, vb, a.ld_k, a.ld_v, min(kt + 2, t1 - 1), a.nk, j, g);      // in flight during the MFMAs
    const KVTile n1 = load_kv(kb, vb, a.ld_k, a.ld_v, min(kt + 3, t1 - 1), a.nk, j, g);
    const bool two = kt + 1 < t1;                               // wave-uniform: the second tile exists
    const bool full = two && (kt + 2) * 16 <= a.nk;             // no key of the pair is beyond nk
    const float k0[4] = {c0.kf.x, c0.kf.y, c0.kf.z, c0.kf.w}, k1[4] = {c1.kf.x, c1.kf.y, c1.kf.z, c1.kf.w};
    f32x4 s0[NT], s1[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) s0[t] = s1[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        s0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(k0[c], qf[t][c], s0[t], 0, 0, 0);
        s1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(k1[c], qf[t][c], s1[t], 0, 0, 0);
      }
    // lane (query j of tile t, keys 4g + r of either key tile)
    const int key0 = kt * 16 + 4 * g;
    float p0[NT][4], p1[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (!full) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (key0 + r >= a.nk) s0[t][r] = -INFINITY;
          if (!two || key0 + 16 + r >= a.nk) s1[t][r] = -INFINITY;
        }
      }
      float mx = fmaxf(fmaxf(fmaxf(s0[t][0], s0[t][1]), fmaxf(s0[t][2], s0[t][3])),
                       fmaxf(fmaxf(s1[t][0], s1[t][1]), fmaxf(s1[t][2], s1[t][3])));
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float mnew = fmaxf(m[t], mx);                       // finite: key 0 of the first tile exists
      const float alpha = __builtin_amdgcn_exp2f(m[t] - mnew);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p0[t][r] = __builtin_amdgcn_exp2f(s0[t][r] - mnew);
        p1[t][r] = __builtin_amdgcn_exp2f(s1[t][r] - mnew);
      }
      ls[t] = ls[t] * alpha + (((p0[t][0] + p0[t][1]) + (p0[t][2] + p0[t][3])) + ((p1[t][0] + p1[t][1]) + (p1[t][2] + p1[t][3])));
      o[t] *= alpha;
      m[t] = mnew;
    }
    if (a.thr) {                                                // training: dropout of the probabilities (not of the normaliser)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const unsigned long long e0 = (((unsigned long long)b * a.heads + h) * a.nq + (tile0 + 4 * t) * 16 + j) * a.nk + key0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p0[t][r] = rd_keep(e0 + r, a.s0, a.s1, a.thr) ? p0[t][r] * a.dscale : 0.f;
          p1[t][r] = rd_keep(e0 + 16 + r, a.s0, a.s1, a.thr) ? p1[t][r] * a.dscale : 0.f;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {                               // consecutive MFMAs hit different accumulators
#pragma unroll
      for (int t = 0; t < NT; ++t) o[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(c0.vf[r], p0[t][r], o[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) o[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1.vf[r], p1[t][r], o[t], 0, 0, 0);
    }
    c0 = n0;
    c1 = n1;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float l = ls[t];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const int q = (tile0 + 4 * t) * 16 + j;                     // < nq_pad
    const size_t slot = (((size_t)b * a.heads + h) * a.nchunks + chunk) * a.nq_pad + q;
    *(f32x4 *)(a.po + slot * 16 + 4 * g) = o[t];               // lane (query j, head-dim rows 4g + r)
    if (g == 0) {
      a.pml[slot * 2] = m[t];
      a.pml[slot * 2 + 1] = l;
    }
  }
}

__global__ __launch_bounds__(256) void xattn_partial_kernel(XAttnArgs a) {
  const int chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z / a.qblocks, qb = blockIdx.z - b * a.qblocks;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntiles = (a.nk + 15) >> 4;
  const int t0 = chunk * a.tiles_per_chunk, t1 = min(t0 + a.tiles_per_chunk, ntiles);
  const int nqt = (a.nq + 15) >> 4;
  const int tile0 = qb * (4 * XQ) + wave;                       // this wave's tiles: tile0 + 4 t
  if (tile0 >= nqt) return;
  const int mine = min(XQ, (nqt - tile0 + 3) / 4);              // wave-uniform
  const int j = lane & 15, g = lane >> 4;
  if (mine == 4) xattn_wave<4>(a, b, h, chunk, tile0, t0, t1, j, g);
  else if (mine == 3) xattn_wave<3>(a, b, h, chunk, tile0, t0, t1, j, g);
  else if (mine == 2) xattn_wave<2>(a, b, h, chunk, tile0, t0, t1, j, g);
  else xattn_wave<1>(a, b, h, chunk, tile0, t0, t1, j, g);
}

// one wave per (b, h, q): lane = (chunk phase cg = lane >> 4, head-dim d = lane & 15); every lane merges the chunks
// c = cg (mod 4) with a running (M, L, O), then the four phases are merged across the lane groups
__global__ __launch_bounds__(256) void xattn_combine_kernel(XAttnArgs a, float *__restrict__ out, int ld_out) {
  const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= (long long)a.batch * a.nq * a.heads) return;
  const int lane = threadIdx.x & 63, d = lane & 15, cg = lane >> 4;
  const int h = (int)(w % a.heads);
  const long long bq = w / a.heads;
  const int b = (int)(bq / a.nq), q = (int)(bq - (long long)b * a.nq);
  const size_t base = ((size_t)b * a.heads + h) * a.nchunks;
  float M = -INFINITY, L = 0.f, O = 0.f;
  for (int c = cg; c < a.nchunks; c += 4) {
    const size_t slot = (base + c) * a.nq_pad + q;
    const float2 ml = *(const float2 *)(a.pml + slot * 2);
    const float o = a.po[slot * 16 + d];
    const float Mn = fmaxf(M, ml.x);
    const float w0 = __builtin_amdgcn_exp2f(M - Mn), w1 = __builtin_amdgcn_exp2f(ml.x - Mn);
    L = L * w0 + ml.y * w1;
    O = O * w0 + o * w1;
    M = Mn;
  }
#pragma unroll
  for (int s = 16; s <= 32; s <<= 1) {
    const float M2 = __shfl_xor(M, s), L2 = __shfl_xor(L, s), O2 = __shfl_xor(O, s);
    const float Mn = fmaxf(M, M2);
    const bool none = Mn == -INFINITY;                           // both phases empty (fewer than 4 chunks)
    const float w0 = none ? 0.f : __builtin_amdgcn_exp2f(M - Mn), w1 = none ? 0.f : __builtin_amdgcn_exp2f(M2 - Mn);
    L = L * w0 + L2 * w1;
    O = O * w0 + O2 * w1;
    M = Mn;
  }
  if (cg == 0) out[((size_t)b * a.nq + q) * ld_out + h * 16 + d] = O / L;
  if (a.lse && lane == 0) a.lse[((size_t)b * a.heads + h) * a.nq + q] = M + __builtin_amdgcn_logf(L);   // v_log_f32 = log2
}

// ---- backward (round 6: the TransFusion decoder's cross-attention in a training step) ---------------------------------------------
// With s = scale <q, k>, P = softmax_k(s), Pd = P . keep / (1 - p), O = Pd V:
//   dV = Pd^T dO,  dPd = dO V^T,  dS = P . (dPd . keep / (1 - p) - delta),  delta_q = <dO_q, O_q>,  dQ = scale dS K,  dK = scale dS^T Q.
// The torch composition writes and re-reads the [B, heads, nq, nk] tensors (829 MB at 4 x 8 x 200 x 32 400) about ten times
// (softmax, dropout and its mask, two products, and their backward: ~4.1 ms per step); here they never exist: P is recomputed
// from the forward's base-2 log-sum-exp, 16 x 16 tiles on v_mfma_f32_16x16x4_f32 (the 16-wide head dimension is the contraction
// of the score products: 4 instructions each).  A contraction over KEYS (dQ) wants the probability tile as [key][query] in the
// accumulator layout (lane = column, rows 4 g + r), one over QUERIES (dK, dV) as [query][key]: each result then feeds the next
// product straight from its registers, the four steps of a product contracting over rows {4 g + r : g}.  Two kernels, one per
// orientation -- no tile is formed twice for the same purpose, and neither carries more than two accumulator tiles per
// gradient (a single kernel holding dQ of all 16 query tiles over a key walk compiled to 512 registers + scratch: 4.7 ms):
//   xattn_bwd_kv   a wave owns a chunk of key tiles of one (b, h); per key tile it walks ALL query tiles (Q x scale log2 e, dO,
//                  log-sum-exp, delta staged in LDS once per workgroup) and stores the finished dK / dV tile;
//   xattn_bwd_q    a wave owns two query tiles and a chunk of key tiles; K / V tiles stream from L2 (the four waves of a
//                  workgroup read the same ones); its dQ tiles are added to the output once.
struct XBwdArgs {
  const float *q, *k, *v, *dout, *lse, *delta;
  int ld_q, ld_k, ld_v, ld_do;
  int batch, nq, nk, heads;
  float scale, qscale;
  unsigned thr, s0, s1;
  float dscale;
  int tiles_per_chunk, nchunks;        // key chunks of the kernel being launched
  int qblocks;                         // xattn_bwd_q: workgroups of 8 query tiles
  float *dq, *dk, *dv;
  int ld_dq, ld_dk, ld_dv;
};

constexpr int XB_LD = 20;    // floats per staged query row: the column reads of four consecutive rows fall on four bank quarters

struct XKTile {
  float kf[4], vf[4], ks[4];
};
// K / V rows of key tile kt for lane (j, g): kf / vf = dims 4 g .. 4 g + 3 of key j, ks[r] = dim j of key 4 g + r (clamped rows
// beyond nk are masked by the callers)
__device__ __forceinline__ XKTile xb_load_kv(const float *kb, const float *vb, int ld_k, int ld_v, int kt, int nk, int j, int g,
                                             bool want_ks) {
  XKTile t;
  const int key0 = kt * 16;
  const float4 k4 = *(const float4 *)(kb + (size_t)min(key0 + j, nk - 1) * ld_k + 4 * g);
  const float4 v4 = *(const float4 *)(vb + (size_t)min(key0 + j, nk - 1) * ld_v + 4 * g);
  t.kf[0] = k4.x, t.kf[1] = k4.y, t.kf[2] = k4.z, t.kf[3] = k4.w;
  t.vf[0] = v4.x, t.vf[1] = v4.y, t.vf[2] = v4.z, t.vf[3] = v4.w;
#pragma unroll
  for (int r = 0; r < 4; ++r) t.ks[r] = want_ks ? kb[(size_t)min(key0 + 4 * g + r, nk - 1) * ld_k + j] : 0.f;
  return t;
}

__global__ __launch_bounds__(256) void xattn_bwd_kv_kernel(XBwdArgs a) {
  extern __shared__ float xb_smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int chunk = blockIdx.x * 4 + wave, h = blockIdx.y, b = blockIdx.z;
  const int j = lane & 15, g = lane >> 4;
  const int ntiles = (a.nk + 15) >> 4, nqt = (a.nq + 15) >> 4, nqp = nqt * 16;
  float *sQ = xb_smem, *sDO = sQ + nqp * XB_LD, *sL = sDO + nqp * XB_LD, *sD = sL + nqp;
  const float *qb = a.q + (size_t)b * a.nq * a.ld_q + h * 16, *dob = a.dout + (size_t)b * a.nq * a.ld_do + h * 16;
  for (int i = threadIdx.x; i < nqp * 16; i += 256) {
    const int row = i >> 4, c = i & 15;
    sQ[row * XB_LD + c] = row < a.nq ? qb[(size_t)row * a.ld_q + c] * a.qscale : 0.f;
    sDO[row * XB_LD + c] = row < a.nq ? dob[(size_t)row * a.ld_do + c] : 0.f;
  }
  for (int i = threadIdx.x; i < nqp; i += 256) {             // +inf on the padded rows: their probabilities vanish
    sL[i] = i < a.nq ? a.lse[((size_t)b * a.heads + h) * a.nq + i] : INFINITY;
    sD[i] = i < a.nq ? a.delta[((size_t)b * a.heads + h) * a.nq + i] : 0.f;
  }
  __syncthreads();
  if (chunk >= a.nchunks) return;
  const int t0 = chunk * a.tiles_per_chunk, t1 = min(t0 + a.tiles_per_chunk, ntiles);
  const float *kb = a.k + (size_t)b * a.nk * a.ld_k + h * 16, *vb = a.v + (size_t)b * a.nk * a.ld_v + h * 16;
  // dropout element index = ((b heads + h) nq + q) nk + key
  const unsigned long long e_g = (((unsigned long long)b * a.heads + h) * a.nq + 4 * g) * a.nk + j;
  XKTile cur = xb_load_kv(kb, vb, a.ld_k, a.ld_v, t0, a.nk, j, g, false);
  for (int kt = t0; kt < t1; ++kt) {
    const XKTile nxt = xb_load_kv(kb, vb, a.ld_k, a.ld_v, min(kt + 1, t1 - 1), a.nk, j, g, false);
    const int key0 = kt * 16;
    const bool key_j = key0 + j < a.nk;
    f32x4 dka = {0.f, 0.f, 0.f, 0.f}, dva = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int t = 0; t < nqt; ++t) {
      const int q0 = t * 16;
      const float4 qf4 = *(const float4 *)(sQ + (q0 + j) * XB_LD + 4 * g);
      const float4 df4 = *(const float4 *)(sDO + (q0 + j) * XB_LD + 4 * g);
      const float qf[4] = {qf4.x, qf4.y, qf4.z, qf4.w}, df[4] = {df4.x, df4.y, df4.z, df4.w};
      f32x4 sS = {0.f, 0.f, 0.f, 0.f}, dpS = {0.f, 0.f, 0.f, 0.f};       // [query 4 g + r][key j]
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        sS = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[c], cur.kf[c], sS, 0, 0, 0);
        dpS = __builtin_amdgcn_mfma_f32_16x16x4f32(df[c], cur.vf[c], dpS, 0, 0, 0);
      }
      float pd[4], ds[4], qs[4], dos[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qq = q0 + 4 * g + r;
        const float p = key_j ? __builtin_amdgcn_exp2f(sS[r] - sL[qq]) : 0.f;
        const float m = !a.thr ? 1.f : rd_keep(e_g + (unsigned long long)(q0 + r) * a.nk + key0, a.s0, a.s1, a.thr) ? a.dscale : 0.f;
        pd[r] = p * m;
        ds[r] = p * (dpS[r] * m - sD[qq]);
        qs[r] = sQ[qq * XB_LD + j];
        dos[r] = sDO[qq * XB_LD + j];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dva = __builtin_amdgcn_mfma_f32_16x16x4f32(dos[r], pd[r], dva, 0, 0, 0);
        dka = __builtin_amdgcn_mfma_f32_16x16x4f32(qs[r], ds[r], dka, 0, 0, 0);
      }
    }
    if (key_j) {                                                // lane (key j, head-dim rows 4 g + r)
      *(f32x4 *)(a.dv + ((size_t)b * a.nk + key0 + j) * a.ld_dv + h * 16 + 4 * g) = dva;
      // (the staged Q carries scale log2 e: dK = scale dS^T Q = ln 2 . dS^T (scale log2 e Q))
      *(f32x4 *)(a.dk + ((size_t)b * a.nk + key0 + j) * a.ld_dk + h * 16 + 4 * g) = dka * 0.6931471805599453f;
    }
    cur = nxt;
  }
}

__global__ __launch_bounds__(256) void xattn_bwd_q_kernel(XBwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z / a.qblocks, qblk = blockIdx.z - b * a.qblocks;
  const int j = lane & 15, g = lane >> 4;
  const int ntiles = (a.nk + 15) >> 4, nqt = (a.nq + 15) >> 4;
  const int tA = qblk * 8 + wave * 2;                           // this wave's query tiles: tA, tA + 1
  if (tA >= nqt) return;
  const int t0 = chunk * a.tiles_per_chunk, t1 = min(t0 + a.tiles_per_chunk, ntiles);
  const float *qb = a.q + (size_t)b * a.nq * a.ld_q + h * 16, *dob = a.dout + (size_t)b * a.nq * a.ld_do + h * 16;
  const float *kb = a.k + (size_t)b * a.nk * a.ld_k + h * 16, *vb = a.v + (size_t)b * a.nk * a.ld_v + h * 16;
  const float *lse = a.lse + ((size_t)b * a.heads + h) * a.nq, *del = a.delta + ((size_t)b * a.heads + h) * a.nq;
  float qf[2][4], df[2][4], lse_j[2], del_j[2];
  unsigned long long e_j[2];
  f32x4 dqa[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int q = (tA + u) * 16 + j;
    const bool ok = q < a.nq;
    const float4 q4 = *(const float4 *)(qb + (size_t)min(q, a.nq - 1) * a.ld_q + 4 * g);
    const float4 d4 = *(const float4 *)(dob + (size_t)min(q, a.nq - 1) * a.ld_do + 4 * g);
    qf[u][0] = q4.x * a.qscale, qf[u][1] = q4.y * a.qscale, qf[u][2] = q4.z * a.qscale, qf[u][3] = q4.w * a.qscale;
    df[u][0] = d4.x, df[u][1] = d4.y, df[u][2] = d4.z, df[u][3] = d4.w;
    lse_j[u] = ok ? lse[q] : INFINITY;
    del_j[u] = ok ? del[q] : 0.f;
    e_j[u] = (((unsigned long long)b * a.heads + h) * a.nq + q) * a.nk;
    dqa[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  XKTile cur = xb_load_kv(kb, vb, a.ld_k, a.ld_v, t0, a.nk, j, g, true);
  for (int kt = t0; kt < t1; ++kt) {
    const XKTile nxt = xb_load_kv(kb, vb, a.ld_k, a.ld_v, min(kt + 1, t1 - 1), a.nk, j, g, true);
    const int key0 = kt * 16 + 4 * g;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f32x4 sT = {0.f, 0.f, 0.f, 0.f}, dpT = {0.f, 0.f, 0.f, 0.f};       // [key 4 g + r][query j]
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        sT = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.kf[c], qf[u][c], sT, 0, 0, 0);
        dpT = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.vf[c], df[u][c], dpT, 0, 0, 0);
      }
      float dsT[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = key0 + r < a.nk ? __builtin_amdgcn_exp2f(sT[r] - lse_j[u]) : 0.f;
        const float m = !a.thr ? 1.f : rd_keep(e_j[u] + key0 + r, a.s0, a.s1, a.thr) ? a.dscale : 0.f;
        dsT[r] = p * (dpT[r] * m - del_j[u]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) dqa[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.ks[r], dsT[r], dqa[u], 0, 0, 0);
    }
    cur = nxt;
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {                                 // lane (query j, head-dim rows 4 g + r)
    const int q = (tA + u) * 16 + j;
    if (q < a.nq) {
      float *d = a.dq + ((size_t)b * a.nq + q) * a.ld_dq + h * 16 + 4 * g;
#pragma unroll
      for (int r = 0; r < 4; ++r) unsafeAtomicAdd(d + r, dqa[u][r] * a.scale);
    }
  }
}

static void xattn_plan(int batch, int heads, int nq, int nk, XAttnArgs &a) {
  const int ntiles = (nk + 15) / 16;
  a.qblocks = cdiv((nq + 15) / 16, 4 * XQ);
  a.nq_pad = a.qblocks * 4 * XQ * 16;
  static const char *env = getenv("DF3D_XATTN_WORKGROUPS");
  const int target = env ? atoi(env) : 768;
  int nchunks = target / (heads * batch * a.qblocks);
  if (nchunks < 1) nchunks = 1;
  if (nchunks > ntiles) nchunks = ntiles;
  a.tiles_per_chunk = cdiv(ntiles, nchunks);
  a.nchunks = cdiv(ntiles, a.tiles_per_chunk);
}

}  // namespace df3d

using namespace df3d;

static bool xattn_sizes_ok(int batch, int heads, int nq, int nk) {
  return batch > 0 && heads > 0 && nq > 0 && nk > 0 && (long long)batch * 64 <= 65535 && heads <= 65535;
}

extern "C" size_t df3d_cross_attention_workspace_bytes(int batch, int heads, int nq, int nk) {
  if (!xattn_sizes_ok(batch, heads, nq, nk)) return 0;
  XAttnArgs a;
  xattn_plan(batch, heads, nq, nk, a);
  return (size_t)batch * heads * a.nchunks * a.nq_pad * 18 * sizeof(float) + 256;
}

extern "C" int df3d_cross_attention(const float *q, int ld_q, const float *k, int ld_k, const float *v, int ld_v, int batch,
                                    int nq, int nk, int heads, int head_dim, float scale, float *out, int ld_out,
                                    void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(q && k && v && out && workspace, "cross_attention: null argument");
  DF3D_CHECK_ARG(head_dim == 16, "cross_attention: head dimension must be 16 (got %d)", head_dim);
  DF3D_CHECK_ARG(xattn_sizes_ok(batch, heads, nq, nk), "cross_attention: bad sizes");
  const int E = heads * 16;
  DF3D_CHECK_ARG(ld_q >= E && ld_k >= E && ld_v >= E && ld_out >= E && ld_q % 4 == 0 && ld_k % 4 == 0,
                 "cross_attention: row strides must cover %d columns (q / k strides multiples of 4)", E);
  DF3D_CHECK_ARG(((uintptr_t)q & 15) == 0 && ((uintptr_t)k & 15) == 0, "cross_attention: q / k must be 16-byte aligned");
  XAttnArgs a;
  memset(&a, 0, sizeof(a));
  xattn_plan(batch, heads, nq, nk, a);
  const size_t need = (size_t)batch * heads * a.nchunks * a.nq_pad * 18 * sizeof(float);
  DF3D_CHECK_ARG(workspace_bytes >= need, "cross_attention: workspace %zu < %zu bytes", workspace_bytes, need);
  a.q = q;
  a.k = k;
  a.v = v;
  a.ld_q = ld_q;
  a.ld_k = ld_k;
  a.ld_v = ld_v;
  a.batch = batch;
  a.nq = nq;
  a.nk = nk;
  a.heads = heads;
  a.qscale = scale * 1.4426950408889634f;
  a.po = (float *)workspace;
  a.pml = a.po + (size_t)batch * heads * a.nchunks * a.nq_pad * 16;
  hipLaunchKernelGGL(xattn_partial_kernel, dim3(a.nchunks, heads, batch * a.qblocks), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(xattn_combine_kernel, dim3(cdiv((long long)batch * nq * heads, 4)), dim3(256), 0, stream, a, out, ld_out);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

static unsigned xattn_threshold(float p) { return (unsigned)((double)p * 16777216.0); }

extern "C" int df3d_cross_attention_train(const float *q, int ld_q, const float *k, int ld_k, const float *v, int ld_v, int batch,
                                          int nq, int nk, int heads, int head_dim, float scale, float p, unsigned long long seed,
                                          float *out, int ld_out, float *lse, void *workspace, size_t workspace_bytes,
                                          void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(q && k && v && out && lse && workspace, "cross_attention_train: null argument");
  DF3D_CHECK_ARG(head_dim == 16, "cross_attention_train: head dimension must be 16 (got %d)", head_dim);
  DF3D_CHECK_ARG(xattn_sizes_ok(batch, heads, nq, nk), "cross_attention_train: bad sizes");
  DF3D_CHECK_ARG(p >= 0.f && p < 1.f, "cross_attention_train: p must be in [0, 1) (got %g)", (double)p);
  const int E = heads * 16;
  DF3D_CHECK_ARG(ld_q >= E && ld_k >= E && ld_v >= E && ld_out >= E && ld_q % 4 == 0 && ld_k % 4 == 0,
                 "cross_attention_train: row strides must cover %d columns (q / k strides multiples of 4)", E);
  DF3D_CHECK_ARG(((uintptr_t)q & 15) == 0 && ((uintptr_t)k & 15) == 0, "cross_attention_train: q / k must be 16-byte aligned");
  XAttnArgs a;
  memset(&a, 0, sizeof(a));
  xattn_plan(batch, heads, nq, nk, a);
  const size_t need = (size_t)batch * heads * a.nchunks * a.nq_pad * 18 * sizeof(float);
  DF3D_CHECK_ARG(workspace_bytes >= need, "cross_attention_train: workspace %zu < %zu bytes", workspace_bytes, need);
  a.q = q, a.k = k, a.v = v, a.ld_q = ld_q, a.ld_k = ld_k, a.ld_v = ld_v;
  a.batch = batch, a.nq = nq, a.nk = nk, a.heads = heads;
  a.qscale = scale * 1.4426950408889634f;
  a.po = (float *)workspace;
  a.pml = a.po + (size_t)batch * heads * a.nchunks * a.nq_pad * 16;
  a.thr = xattn_threshold(p), a.s0 = (unsigned)seed, a.s1 = (unsigned)(seed >> 32);
  a.dscale = a.thr ? (float)(1.0 / (1.0 - (double)a.thr / 16777216.0)) : 1.f;
  a.lse = lse;
  hipLaunchKernelGGL(xattn_partial_kernel, dim3(a.nchunks, heads, batch * a.qblocks), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(xattn_combine_kernel, dim3(cdiv((long long)batch * nq * heads, 4)), dim3(256), 0, stream, a, out, ld_out);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_cross_attention_backward(const float *q, int ld_q, const float *k, int ld_k, const float *v, int ld_v,
                                             const float *grad_out, int ld_do, const float *lse, const float *delta, int batch,
                                             int nq, int nk, int heads, int head_dim, float scale, float p,
                                             unsigned long long seed, float *dq, int ld_dq, float *dk, int ld_dk, float *dv,
                                             int ld_dv, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(q && k && v && grad_out && lse && delta && dq && dk && dv, "cross_attention_backward: null argument");
  DF3D_CHECK_ARG(head_dim == 16, "cross_attention_backward: head dimension must be 16 (got %d)", head_dim);
  DF3D_CHECK_ARG(xattn_sizes_ok(batch, heads, nq, nk) && nq <= 256, "cross_attention_backward: bad sizes (nq <= 256)");
  DF3D_CHECK_ARG(p >= 0.f && p < 1.f, "cross_attention_backward: p must be in [0, 1) (got %g)", (double)p);
  const int E = heads * 16;
  DF3D_CHECK_ARG(ld_q >= E && ld_k >= E && ld_v >= E && ld_do >= E && ld_dq >= E && ld_dk >= E && ld_dv >= E &&
                     (ld_q | ld_k | ld_v | ld_do | ld_dq | ld_dk | ld_dv) % 4 == 0,
                 "cross_attention_backward: row strides must cover %d columns and be multiples of 4", E);
  DF3D_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)grad_out | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) == 0,
                 "cross_attention_backward: operands must be 16-byte aligned");
  XBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.q = q, a.k = k, a.v = v, a.dout = grad_out, a.lse = lse, a.delta = delta;
  a.ld_q = ld_q, a.ld_k = ld_k, a.ld_v = ld_v, a.ld_do = ld_do;
  a.batch = batch, a.nq = nq, a.nk = nk, a.heads = heads;
  a.scale = scale, a.qscale = scale * 1.4426950408889634f;
  a.thr = xattn_threshold(p), a.s0 = (unsigned)seed, a.s1 = (unsigned)(seed >> 32);
  a.dscale = a.thr ? (float)(1.0 / (1.0 - (double)a.thr / 16777216.0)) : 1.f;
  const int ntiles = (nk + 15) / 16, nqt = (nq + 15) / 16, nqp = nqt * 16;
  a.dq = dq, a.dk = dk, a.dv = dv, a.ld_dq = ld_dq, a.ld_dk = ld_dk, a.ld_dv = ld_dv;
  static const char *env = getenv("DF3D_XATTN_BWD_WAVES");           // tuning aid: waves wanted per kernel
  const int want = env ? atoi(env) : 4096;
  auto chunks = [&](int per_bh) {
    const int n = std::max(1, std::min(want / std::max(1, heads * batch * per_bh), ntiles));
    a.tiles_per_chunk = cdiv(ntiles, n);
    a.nchunks = cdiv(ntiles, a.tiles_per_chunk);
  };
  DF3D_HIP(hipMemset2DAsync(dq, (size_t)ld_dq * sizeof(float), 0, (size_t)E * sizeof(float), (size_t)batch * nq, stream));
  chunks(1);
  hipLaunchKernelGGL(xattn_bwd_kv_kernel, dim3(cdiv(a.nchunks, 4), heads, batch), dim3(256),
                     (size_t)(2 * nqp * XB_LD + 2 * nqp) * sizeof(float), stream, a);
  a.qblocks = cdiv(nqt, 8);
  chunks(cdiv(nqt, 2));
  hipLaunchKernelGGL(xattn_bwd_q_kernel, dim3(a.nchunks, heads, batch * a.qblocks), dim3(256), 0, stream, a);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}
