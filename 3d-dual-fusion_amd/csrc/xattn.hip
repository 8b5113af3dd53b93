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
