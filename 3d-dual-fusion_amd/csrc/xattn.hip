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
//                   transposed scores a lane holds 4 keys of ONE query), P^T = exp2(S^T - m), O^T += V^T.P^T (4 MFMA).
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

__global__ __launch_bounds__(256) void xattn_partial_kernel(XAttnArgs a) {
  const int chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z / a.qblocks, qb = blockIdx.z - b * a.qblocks;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int ntiles = (a.nk + 15) >> 4;
  const int t0 = chunk * a.tiles_per_chunk, t1 = min(t0 + a.tiles_per_chunk, ntiles);
  const int nqt = (a.nq + 15) >> 4;
  f32x4 qf[XQ], o[XQ];
  float m[XQ], ls[XQ];
  int qtile[XQ];
#pragma unroll
  for (int t = 0; t < XQ; ++t) {
    qtile[t] = qb * (4 * XQ) + wave + 4 * t;
    const int q = qtile[t] * 16 + j;
    qf[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (qtile[t] < nqt && q < a.nq) {
      const float4 x = *(const float4 *)(a.q + ((size_t)b * a.nq + q) * a.ld_q + h * 16 + 4 * g);
      qf[t] = (f32x4){x.x * a.qscale, x.y * a.qscale, x.z * a.qscale, x.w * a.qscale};
    }
    o[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    m[t] = -INFINITY;
    ls[t] = 0.f;
  }
  if (qtile[0] >= nqt) return;                                  // whole wave idle (qtile grows with t)
  const float *kb = a.k + (size_t)b * a.nk * a.ld_k + h * 16, *vb = a.v + (size_t)b * a.nk * a.ld_v + h * 16;
  for (int kt = t0; kt < t1; ++kt) {
    const int key0 = kt * 16;
    const int krow = min(key0 + j, a.nk - 1);                   // clamped rows are masked below
    const float4 kf = *(const float4 *)(kb + (size_t)krow * a.ld_k + 4 * g);
    float vf[4];
    bool valid[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = key0 + 4 * g + r;
      valid[r] = key < a.nk;
      vf[r] = valid[r] ? vb[(size_t)key * a.ld_v + j] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < XQ; ++t) {
      if (qtile[t] >= nqt) break;                               // wave-uniform
      f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
      s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[t][0], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[t][1], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[t][2], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[t][3], s, 0, 0, 0);
      // lane (query j, keys 4g + r)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (!valid[r]) s[r] = -INFINITY;
      float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float mnew = fmaxf(m[t], mx);                       // finite: key 0 of every tile exists
      const float alpha = __builtin_amdgcn_exp2f(m[t] - mnew);
      float p[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(s[r] - mnew);
      ls[t] = ls[t] * alpha + ((p[0] + p[1]) + (p[2] + p[3]));
      o[t] *= alpha;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r], p[r], o[t], 0, 0, 0);
      m[t] = mnew;
    }
  }
#pragma unroll
  for (int t = 0; t < XQ; ++t) {
    if (qtile[t] >= nqt) break;
    float l = ls[t];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const int q = qtile[t] * 16 + j;                            // < nq_pad
    const size_t slot = (((size_t)b * a.heads + h) * a.nchunks + chunk) * a.nq_pad + q;
    *(f32x4 *)(a.po + slot * 16 + 4 * g) = o[t];               // lane (query j, head-dim rows 4g + r)
    if (g == 0) {
      a.pml[slot * 2] = m[t];
      a.pml[slot * 2 + 1] = l;
    }
  }
}

// thread = (b, q, h, d)
__global__ __launch_bounds__(256) void xattn_combine_kernel(XAttnArgs a, float *__restrict__ out, int ld_out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)a.batch * a.nq * a.heads * 16;
  if (i >= total) return;
  const int d = (int)(i & 15), h = (int)((i >> 4) % a.heads);
  const long long bq = (i >> 4) / a.heads;
  const int b = (int)(bq / a.nq), q = (int)(bq - (long long)b * a.nq);
  const size_t base = ((size_t)b * a.heads + h) * a.nchunks;
  float M = -INFINITY;
  for (int c = 0; c < a.nchunks; ++c) M = fmaxf(M, a.pml[((base + c) * a.nq_pad + q) * 2]);
  float L = 0.f, O = 0.f;
  for (int c = 0; c < a.nchunks; ++c) {
    const size_t slot = (base + c) * a.nq_pad + q;
    const float w = __builtin_amdgcn_exp2f(a.pml[slot * 2] - M);
    L += a.pml[slot * 2 + 1] * w;
    O += a.po[slot * 16 + d] * w;
  }
  out[((size_t)b * a.nq + q) * ld_out + h * 16 + d] = O / L;
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
  const long long total = (long long)batch * nq * heads * 16;
  hipLaunchKernelGGL(xattn_combine_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, a, out, ld_out);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}
