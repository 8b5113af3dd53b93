// One pre-norm transformer encoder layer over ball-query groups of 32 tokens x 64 channels as ONE kernel (round 3).
//
// Reference: the LocalTransformer of ACTRv2 (3-D local self-attention of the Voxel-RCNN tree) --
//   CP/det3d/models/model_utils/pointformer.py:10-44 (TransformerEncoderLayerPreNorm:
//       x1 = norm1(x);  x2 = x1 + out_proj(MHA(x1, x1, x1));  x3 = norm2(x2);  y = x3 + linear2(relu(linear1(x3)))),
//   :349-380 (the chunk of two such layers on [nsample, B * npoint, C] sequences), nn.MultiheadAttention with 4 heads of 16.
// Rounds 1-2 ran a layer as ~8 kernels over the [524 k, 64] activation rows (LayerNorm, in-projection, group attention,
// out-projection, add + LayerNorm, fused FFN): ~0.84 ms and ~3 GB of HBM traffic per layer where the rows in and out are
// 0.27 GB.  Here a WAVE owns a group (32 tokens = two MFMA row tiles) from the load of its rows to the store of the result:
//
//  * every matrix product runs on `v_mfma_f32_16x16x32_f16` with fp32 operands split into fp16 hi + lo of the scaled value
//    (csrc/common.h: activations x 2^5, weights x 2^7; three products, fp32 accumulate, the accumulator multiplied by 2^-12 --
//    2^-10 for the two activation x activation products K Q^T and V^T P^T; ~1e-6 of the output scale like the split-precision
//    convolutions.  Rounds 3-4 split into bf16 pairs: 16 significand bits);
//  * activations live in registers in ONE layout from start to end -- "token layout" T: lane (n, g) holds, of token
//    16 tt + n, the channels 16 ct + 4 g + r (ct = 0..3, r = 0..3).  That is the accumulator layout of a product computed
//    TRANSPOSED (weights as the A operand, D[out channel][token]), and -- because the contraction index of an MFMA may be
//    permuted freely as long as both operands agree -- it is also its B operand: k-slot (g, j) of k-step s <-> channel
//    16 (2 s + (j >> 2)) + 4 g + (j & 3), eight values the lane already holds.  The weights are packed once with that
//    permutation (dualfusion/ops.py lt_layer_pack).  So LayerNorm -> Q, K -> scores -> softmax -> P V -> out-projection ->
//    residual -> LayerNorm -> FFN chain through registers without a single transpose or LDS round trip:
//      Q^T, K^T  = W x1^T              (layout T; head h = channel tile h)
//      V         = x1 Wv^T             (the same x1 registers as A operand; accumulator lane = channel d, regs = tokens)
//      S^T       = K Q^T per head      (contraction over d = 16: slots [hi d | lo d] x [hi d | hi d], then [hi d | 0] x [lo d | 0];
//                                       operands are the lanes' own K^T / Q^T accumulators)
//      softmax over the 32 keys of a query = 8 values in the lane x the 4 lanes g: two cross-lane steps
//      O^T       = V^T P^T             (A = V accumulators: lane d, k-slots = the 8 tokens 4 g + r of both token tiles;
//                                       B = P^T = the S^T accumulators: same slots) -> layout T again
//      att^T, h^T, y^T = W . ^T        (transposed products; the hidden chunk's accumulators are the next B operand)
//  * the packed weights of the layer (128 KB) sit in LDS, fetched once per workgroup; fragments are lane-linear 16-byte
//    reads (conflict-free); the only cross-lane traffic is the LayerNorm / softmax reductions.
// Algorithmic traffic: rows in + rows out = 2 x 134 MB per layer at the Voxel-RCNN configuration (bound: HBM, 34 us at
// 8 TB/s; the matrix work is 440 MFMAs per group = 47 us chip-wide).
#include "common.h"

namespace df3d {

typedef float lt_f32x4 __attribute__((ext_vector_type(4)));

DF3D_SPLIT_OVERFLOW_TU(ltlayer)
typedef unsigned int lt_u32x4 __attribute__((ext_vector_type(4)));
#define LT_MFMA(A, B, C) DF3D_MFMA_F16(A, B, C)
#define LT_U DF3D_ACC_UNSCALE       /* weights x activations */
#define LT_AA DF3D_AA_UNSCALE       /* activations x activations */

// The machine scheduler otherwise hoists the LDS fragment reads of the whole (fully unrolled) layer to the top: 900 live
// registers.  A scheduling barrier after every stage keeps live ranges to what the stage needs.
#define LT_FENCE() __builtin_amdgcn_sched_barrier(0)

constexpr int LT_PAIRS = 64;                      // fragment pairs (hi, lo) of one layer: 24 in-proj, 8 out-proj, 16 + 16 FFN
constexpr int LT_WBYTES = LT_PAIRS * 2 * 64 * 16; // 128 KB
constexpr int LT_NVEC = 704;                      // b_qkv 192 | b_o 64 | b_1 128 | b_2 64 | g1 | be1 | g2 | be2 (64 each)
// gather mode (the first layer of a LocalTransformer chunk): 4 more pairs = the positional MLP's second linear [64, 32]
// (pairs 64..67 = out tile), and 192 more floats = its first linear (BatchNorm folded) w0 [32][3] | b0 [32] | b1 [64]
constexpr int LT_PE_PAIRS = 4;
constexpr int LT_PE_WBYTES = LT_PE_PAIRS * 2 * 64 * 16;
constexpr int LT_PE_NVEC = 192;

struct LtArgs {
  const float *x;          // [32][G][64] rows (sequence-first: row = token * G + group)
  const lt_u32x4 *w;       // packed fragments [pair][hi | lo][lane]
  const float *vec;        // LT_NVEC floats
  float *out;              // [32][G][64]
  int G;
  float eps1, eps2;
  long long ts, gs;        // floats between two tokens of a group / between two groups (input and output alike)
  // GATHER: token (t, grp) reads row sel[t * G + grp] of `x` ([rows, 64]) and adds the positional MLP of gxyz[t * G + grp]
  const long long *sel;
  const float *gxyz;       // [32 * G, 3]
  // SCATTER: besides nothing else, token (t, grp) writes its row to out[dst[t * G + grp]] when dst >= 0 (out = [rows, 64])
  const long long *dst;
};

struct LtOp {
  lt_u32x4 hi, lo;
};

// eight fp32 values (two float4 of the lane) -> one MFMA operand, hi and lo parts
// (range check without a branch per pair: split_pair_acc / split_range_flag, csrc/common.h)
__device__ __forceinline__ LtOp lt_split(lt_f32x4 a, lt_f32x4 b, float &amax) {
  unsigned h[4], l[4];
  split_pair_acc(a[0], a[1], h[0], l[0], amax);
  split_pair_acc(a[2], a[3], h[1], l[1], amax);
  split_pair_acc(b[0], b[1], h[2], l[2], amax);
  split_pair_acc(b[2], b[3], h[3], l[3], amax);
  LtOp o;
  o.hi = (lt_u32x4){h[0], h[1], h[2], h[3]};
  o.lo = (lt_u32x4){l[0], l[1], l[2], l[3]};
  return o;
}

__device__ __forceinline__ float lt_sum4(lt_f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }

// sum over the four lanes (n, g = 0..3) that hold one token
__device__ __forceinline__ float lt_token_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float lt_token_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

// LayerNorm over the 64 channels of the lane's token (layout T), two passes in registers
__device__ __forceinline__ void lt_layernorm(lt_f32x4 (&t)[4], const float *gamma, const float *beta, float eps, int g) {
  float s = (lt_sum4(t[0]) + lt_sum4(t[1])) + (lt_sum4(t[2]) + lt_sum4(t[3]));
  const float mean = lt_token_sum(s) * (1.f / 64.f);
  float q = 0.f;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    t[ct] -= mean;
    q += lt_sum4(t[ct] * t[ct]);
  }
  const float rstd = rsqrtf(lt_token_sum(q) * (1.f / 64.f) + eps);
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const lt_f32x4 ga = *(const lt_f32x4 *)(gamma + ct * 16 + 4 * g), be = *(const lt_f32x4 *)(beta + ct * 16 + 4 * g);
    t[ct] = t[ct] * rstd * ga + be;
  }
}

// acc[tt] += W_frag(pair) . act^T  (transposed product: weights are the A operand; result in layout T)
__device__ __forceinline__ void lt_gemm_t(const lt_u32x4 *Wl, int pair, int lane, const LtOp &b0, const LtOp &b1, lt_f32x4 &acc0,
                                          lt_f32x4 &acc1) {
  const lt_u32x4 wh = Wl[(pair * 2) * 64 + lane], wl = Wl[(pair * 2 + 1) * 64 + lane];
  acc0 = LT_MFMA(wl, b0.hi, acc0);
  acc1 = LT_MFMA(wl, b1.hi, acc1);
  acc0 = LT_MFMA(wh, b0.lo, acc0);
  acc1 = LT_MFMA(wh, b1.lo, acc1);
  acc0 = LT_MFMA(wh, b0.hi, acc0);
  acc1 = LT_MFMA(wh, b1.hi, acc1);
}

template <int NW, bool GATHER, bool SCATTER>
__global__ __launch_bounds__(NW * 64) void lt_layer_kernel(LtArgs a) {
  constexpr int WBYTES = LT_WBYTES + (GATHER ? LT_PE_WBYTES : 0);
  constexpr int NVEC = LT_NVEC + (GATHER ? LT_PE_NVEC : 0);
  extern __shared__ __align__(16) unsigned char lt_smem[];
  lt_u32x4 *Wl = (lt_u32x4 *)lt_smem;
  float *vl = (float *)(lt_smem + WBYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, n = lane & 15;
  {
    constexpr int TOT = WBYTES / 16, PER = TOT / (NW * 64);
    static_assert(TOT % (NW * 64) == 0, "weight image size");
#pragma unroll 1
    for (int b = 0; b < PER; b += 8) {              // eight 16-byte loads in flight per thread
      lt_u32x4 t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = a.w[tid + NW * 64 * ((b + i) < PER ? (b + i) : 0)];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (b + i < PER) Wl[tid + NW * 64 * (b + i)] = t[i];
    }
    for (int e = tid; e < NVEC; e += NW * 64) vl[e] = a.vec[e];
  }
  __syncthreads();
  const float *bqkv = vl, *bo = vl + 192, *b1 = vl + 256, *b2 = vl + 384;
  const float *g1 = vl + 448, *be1 = vl + 512, *g2 = vl + 576, *be2 = vl + 640;
  const lt_f32x4 zero4 = (lt_f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
  for (int grp = blockIdx.x * NW + wave; grp < a.G; grp += gridDim.x * NW) {
    // ---- rows of the group in layout T, LayerNorm 1 ------------------------------------------------------------
    lt_f32x4 xT[2][4];
    float amax = 0.f;
    if constexpr (GATHER) {
      // x = flat[sel] + pe(xyz): the gathered point row plus the positional MLP 3 -> 32 (ReLU) -> 64 of the grouped coordinate
      // (pointformer.py:287-290,362-364); the second linear is one more transposed product, its hidden operand = the 8 hidden
      // channels 16 (j >> 2) + 4 g + (j & 3) this lane computes itself
      const float *w0 = vl + LT_NVEC, *b0 = w0 + 96, *b1p = b0 + 32;
      LtOp ho[2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const size_t tok = (size_t)(tt * 16 + n) * a.G + grp;
        const float *row = a.x + (size_t)a.sel[tok] * 64 + 4 * g;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) xT[tt][ct] = *(const lt_f32x4 *)(row + ct * 16);
        const float px = a.gxyz[tok * 3], py = a.gxyz[tok * 3 + 1], pz = a.gxyz[tok * 3 + 2];
        lt_f32x4 h[2];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int ch = (j >> 2) * 16 + 4 * g + (j & 3);
          const float v = fmaf(w0[ch * 3 + 2], pz, fmaf(w0[ch * 3 + 1], py, fmaf(w0[ch * 3], px, b0[ch])));
          h[j >> 2][j & 3] = fmaxf(v, 0.f);
        }
        ho[tt] = lt_split(h[0], h[1], amax);
      }
#pragma unroll
      for (int ot = 0; ot < 4; ++ot) {
        lt_f32x4 a0 = zero4, a1 = zero4;
        lt_gemm_t(Wl, LT_PAIRS + ot, lane, ho[0], ho[1], a0, a1);
        const lt_f32x4 bb = *(const lt_f32x4 *)(b1p + ot * 16 + 4 * g);
        xT[0][ot] += a0 * LT_U + bb;
        xT[1][ot] += a1 * LT_U + bb;
      }
      LT_FENCE();
    } else {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const float *row = a.x + (size_t)(tt * 16 + n) * a.ts + (size_t)grp * a.gs + 4 * g;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) xT[tt][ct] = *(const lt_f32x4 *)(row + ct * 16);
      }
    }
    lt_layernorm(xT[0], g1, be1, a.eps1, g);
    lt_layernorm(xT[1], g1, be1, a.eps1, g);
    LtOp xo[2][2];                                  // x1 as MFMA operand: [token tile][k-step]
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int s = 0; s < 2; ++s) xo[tt][s] = lt_split(xT[tt][2 * s], xT[tt][2 * s + 1], amax);

    // ---- self-attention, one head (= one 16-channel tile) at a time ---------------------------------------------
    lt_f32x4 oT[2][4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      lt_f32x4 q[2] = {zero4, zero4}, k[2] = {zero4, zero4}, v[2] = {zero4, zero4};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        lt_gemm_t(Wl, h * 2 + s, lane, xo[0][s], xo[1][s], q[0], q[1]);
        lt_gemm_t(Wl, (4 + h) * 2 + s, lane, xo[0][s], xo[1][s], k[0], k[1]);
        // V = x1 Wv^T: activations as the A operand, the weight fragment (same format) as B
        const int pv = (8 + h) * 2 + s;
        const lt_u32x4 wh = Wl[(pv * 2) * 64 + lane], wl = Wl[(pv * 2 + 1) * 64 + lane];
        v[0] = LT_MFMA(xo[0][s].lo, wh, v[0]);
        v[1] = LT_MFMA(xo[1][s].lo, wh, v[1]);
        v[0] = LT_MFMA(xo[0][s].hi, wl, v[0]);
        v[1] = LT_MFMA(xo[1][s].hi, wl, v[1]);
        v[0] = LT_MFMA(xo[0][s].hi, wh, v[0]);
        v[1] = LT_MFMA(xo[1][s].hi, wh, v[1]);
        LT_FENCE();
      }
      const lt_f32x4 bq = *(const lt_f32x4 *)(bqkv + h * 16 + 4 * g), bk = *(const lt_f32x4 *)(bqkv + 64 + h * 16 + 4 * g);
      const float bv = bqkv[128 + h * 16 + n];
      // S^T[key][query] = sum_d K[key][d] Q[query][d] / 4
      lt_u32x4 ka1[2], ka2[2], qb1[2], qb2[2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const lt_f32x4 qq = (q[tt] * LT_U + bq) * 0.25f, kk = k[tt] * LT_U + bk;
        unsigned h01, l01, h23, l23;
        split_pair_acc(kk[0], kk[1], h01, l01, amax);
        split_pair_acc(kk[2], kk[3], h23, l23, amax);
        ka1[tt] = (lt_u32x4){h01, h23, l01, l23};
        ka2[tt] = (lt_u32x4){h01, h23, 0u, 0u};
        split_pair_acc(qq[0], qq[1], h01, l01, amax);
        split_pair_acc(qq[2], qq[3], h23, l23, amax);
        qb1[tt] = (lt_u32x4){h01, h23, h01, h23};
        qb2[tt] = (lt_u32x4){l01, l23, 0u, 0u};
        v[tt] = v[tt] * LT_U + bv;
      }
      lt_f32x4 sc[2][2];                            // [key tile][query tile]
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          sc[kt][qt] = LT_MFMA(ka1[kt], qb1[qt], zero4);
          sc[kt][qt] = LT_MFMA(ka2[kt], qb2[qt], sc[kt][qt]);
        }
      LT_FENCE();
      const LtOp va = lt_split(v[0], v[1], amax);         // V^T as A operand: lane = channel d, slots = tokens of both tiles
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        // softmax over the 32 keys of query 16 qt + n: 8 values here, the rest in the lanes n + 16 g'
        lt_f32x4 p0 = sc[0][qt] * LT_AA, p1 = sc[1][qt] * LT_AA;
        float m = fmaxf(fmaxf(fmaxf(p0[0], p0[1]), fmaxf(p0[2], p0[3])), fmaxf(fmaxf(p1[0], p1[1]), fmaxf(p1[2], p1[3])));
        m = lt_token_max(m);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p0[r] = __expf(p0[r] - m);
          p1[r] = __expf(p1[r] - m);
        }
        const float inv = 1.f / lt_token_sum(lt_sum4(p0) + lt_sum4(p1));
        const LtOp pb = lt_split(p0 * inv, p1 * inv, amax);
        lt_f32x4 o = LT_MFMA(va.lo, pb.hi, zero4);
        o = LT_MFMA(va.hi, pb.lo, o);
        oT[qt][h] = LT_MFMA(va.hi, pb.hi, o) * LT_AA;
        LT_FENCE();
      }
    }

    // ---- out-projection + residual (x1), LayerNorm 2 -------------------------------------------------------------
    {
      LtOp oo[2][2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int s = 0; s < 2; ++s) oo[tt][s] = lt_split(oT[tt][2 * s], oT[tt][2 * s + 1], amax);
#pragma unroll
      for (int ot = 0; ot < 4; ++ot) {
        lt_f32x4 a0 = zero4, a1 = zero4;
#pragma unroll
        for (int s = 0; s < 2; ++s) lt_gemm_t(Wl, 24 + ot * 2 + s, lane, oo[0][s], oo[1][s], a0, a1);
        const lt_f32x4 bb = *(const lt_f32x4 *)(bo + ot * 16 + 4 * g);
        xT[0][ot] += a0 * LT_U + bb;
        xT[1][ot] += a1 * LT_U + bb;
        LT_FENCE();
      }
    }
    lt_layernorm(xT[0], g2, be2, a.eps2, g);
    lt_layernorm(xT[1], g2, be2, a.eps2, g);
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int s = 0; s < 2; ++s) xo[tt][s] = lt_split(xT[tt][2 * s], xT[tt][2 * s + 1], amax);

    // ---- feed-forward 64 -> 128 -> 64 in two hidden chunks of 64, + residual (x3) ---------------------------------
    lt_f32x4 yT[2][4];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int ot = 0; ot < 4; ++ot) yT[tt][ot] = zero4;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      lt_f32x4 hT[2][4];
#pragma unroll
      for (int ht = 0; ht < 4; ++ht) {
        lt_f32x4 a0 = zero4, a1 = zero4;
#pragma unroll
        for (int s = 0; s < 2; ++s) lt_gemm_t(Wl, 32 + (c * 4 + ht) * 2 + s, lane, xo[0][s], xo[1][s], a0, a1);
        const lt_f32x4 bb = *(const lt_f32x4 *)(b1 + (c * 4 + ht) * 16 + 4 * g);
        a0 = a0 * LT_U + bb;
        a1 = a1 * LT_U + bb;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          a0[r] = fmaxf(a0[r], 0.f);
          a1[r] = fmaxf(a1[r], 0.f);
        }
        hT[0][ht] = a0;
        hT[1][ht] = a1;
        LT_FENCE();
      }
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {               // k-step 2 c + sl of the 128 hidden channels
        const LtOp h0 = lt_split(hT[0][2 * sl], hT[0][2 * sl + 1], amax), h1 = lt_split(hT[1][2 * sl], hT[1][2 * sl + 1], amax);
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) lt_gemm_t(Wl, 48 + ot * 4 + 2 * c + sl, lane, h0, h1, yT[0][ot], yT[1][ot]);
        LT_FENCE();
      }
    }
    split_range_flag(amax);
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      float *row;
      if constexpr (SCATTER) {
        // 'unique' + 'replace' aggregation (pointformer.py:315-347,371-372): only the winning occurrence of a point writes,
        // straight into the point's row of the query tensor
        const long long d = a.dst[(size_t)(tt * 16 + n) * a.G + grp];
        if (d < 0) continue;
        row = a.out + (size_t)d * 64 + 4 * g;
      } else {
        row = a.out + (size_t)(tt * 16 + n) * a.ts + (size_t)grp * a.gs + 4 * g;
      }
#pragma unroll
      for (int ot = 0; ot < 4; ++ot)
        *(lt_f32x4 *)(row + ot * 16) = xT[tt][ot] + yT[tt][ot] * LT_U + *(const lt_f32x4 *)(b2 + ot * 16 + 4 * g);
    }
  }
}

}  // namespace df3d

using namespace df3d;

extern "C" long long df3d_lt_layer_packed_bytes(void) { return LT_WBYTES; }
extern "C" int df3d_lt_layer_vector_floats(void) { return LT_NVEC; }
extern "C" long long df3d_lt_layer_pe_packed_bytes(void) { return LT_PE_WBYTES; }
extern "C" int df3d_lt_layer_pe_vector_floats(void) { return LT_PE_NVEC; }

template <bool GATHER, bool SCATTER>
static int lt_launch(const LtArgs &a, hipStream_t stream) {
  constexpr int NW = 8;
  const size_t lds = (size_t)LT_WBYTES + (GATHER ? LT_PE_WBYTES : 0) + (LT_NVEC + (GATHER ? LT_PE_NVEC : 0)) * sizeof(float);
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute((const void *)lt_layer_kernel<NW, GATHER, SCATTER>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS) failed: %s", hipGetErrorString(e));
      return DF3D_EHIP;
    }
    configured = true;
  }
  static int num_cu = 0;
  if (!num_cu) {
    hipDeviceProp_t p;
    num_cu = (hipGetDeviceProperties(&p, 0) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
  }
  int grid = cdiv(a.G, NW);
  if (grid > num_cu) grid = num_cu;                 // one workgroup per CU (its weights fill the LDS), waves walk the groups
  hipLaunchKernelGGL((lt_layer_kernel<NW, GATHER, SCATTER>), dim3(grid), dim3(NW * 64), lds, stream, a);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_lt_layer(const float *x, int L, int G, int C, int heads, int ffn, int group_major, const void *packed,
                             const float *vec, float eps1, float eps2, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(x && packed && vec && out, "lt_layer: null argument");
  DF3D_CHECK_ARG(L == 32 && C == 64 && heads == 4 && ffn == 128,
                 "lt_layer: built for 32 tokens x 64 channels, 4 heads, feed-forward 128 (got %d x %d, %d heads, %d)", L, C,
                 heads, ffn);
  DF3D_CHECK_ARG(G >= 0, "lt_layer: bad group count");
  if (G == 0) return DF3D_OK;
  LtArgs a = {x, (const lt_u32x4 *)packed, vec, out, G, eps1, eps2, group_major ? 64LL : (long long)G * 64,
              group_major ? 32LL * 64 : 64LL, nullptr, nullptr, nullptr};
  return lt_launch<false, false>(a, stream);
}

extern "C" int df3d_lt_layer_gather(const float *points, const long long *sel, const float *gxyz, int G, const void *packed,
                                    const float *vec, float eps1, float eps2, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(points && sel && gxyz && packed && vec && out && G >= 0, "lt_layer_gather: bad argument");
  if (G == 0) return DF3D_OK;
  LtArgs a = {points, (const lt_u32x4 *)packed, vec, out, G, eps1, eps2, (long long)G * 64, 64LL, sel, gxyz, nullptr};
  return lt_launch<true, false>(a, stream);
}

extern "C" int df3d_lt_layer_scatter(const float *x, int G, const void *packed, const float *vec, float eps1, float eps2,
                                     const long long *dst, float *points, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(x && dst && packed && vec && points && G >= 0, "lt_layer_scatter: bad argument");
  if (G == 0) return DF3D_OK;
  LtArgs a = {x, (const lt_u32x4 *)packed, vec, points, G, eps1, eps2, (long long)G * 64, 64LL, nullptr, nullptr, dst};
  return lt_launch<false, true>(a, stream);
}
