// Fused feed-forward block of the encoder layer for gfx950:
//     out = LayerNorm(x + W2 relu(W1 x + b1) + b2)          (actr_transformer.py:413-424, d_model 128, d_ffn 1024)
// as ONE kernel on the 16-bit matrix cores with split-precision operands (see common.h / spconv_split.hip: S x = hi + lo
// in fp16, products hi*hi + lo*hi + hi*lo, fp32 accumulate, ~1e-6 of the output scale against float64).  The reference (and the
// hipBLASLt path) runs two fp32 GEMMs at ~90-110 TFLOP/s plus ReLU, residual add and LayerNorm passes, and moves
// the [rows, 1024] hidden activation through HBM twice; here the hidden activation never leaves registers.
//
// A wave owns 16 rows.  The hidden layer is processed in chunks of 128 units:
//   phase 1 (4 steps, one per 32 input channels):   H^T chunk = W1_c x^T   -- W1 is the MFMA A operand, the
//            rows' x fragments the B operand, so the result lands as  lane (row, g) -> hidden 16t + 4g + {0..3}
//   that register layout IS an MFMA A operand (lane (row, g) holds 8 k-values) once two 16-unit tiles are paired,
//   so after bias + ReLU + hi/lo split the chunk feeds
//   phase 2 (4 steps, one per 32 hidden units):     Y += H_c W2_c^T        -- no shuffle, no LDS round trip.
// The packed weight stream [chunk][8 steps][16 KB] holds exactly the operand images of those steps in order
// (pack_ffn_kernel), staged through LDS by the whole workgroup (double buffered, one barrier per step) as in
// spconv_os_split_kernel.  Lane n owns output columns 8n..8n+7 (permuted W2 columns), so residual add,
// LayerNorm (16-lane reductions) and the 16-byte stores happen in registers.
//
// Algorithmic bytes: rows*128*4 read + rows*128*4 written + 1 MB of weights per workgroup from L2;
// 2*rows*128*1024*2 flops.  Bound: fp16 MFMA at 3 products per fp32 product.
#include <stdlib.h>

#include <string.h>

#include "common.h"

namespace df3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

DF3D_SPLIT_OVERFLOW_TU(ffn)

constexpr int FFN_WQ = 8 * 2 * 64;    // u32x4 per step tile (8 operand tiles x hi/lo x 64 lanes) = 16 KB
// Model width C = 128 (ACTR of the CenterPoint / TransFusion trees) or 64 (ACTRv2 and the LocalTransformer of the Voxel-RCNN
// tree): C / 32 phase-1 steps per chunk, C / 16 output column tiles (lane n owns C / 16 consecutive columns).  Every step
// tile of the stream keeps the 16 KB stride; the phase-2 tiles of C = 64 use the first half.

// stream tile (chunk c, step j): j < 4 -> W1 rows 128c..128c+127, input channels 32j..32j+31 (A operands);
//                                j >= 4 -> W2 columns of hidden units 128c + 32(j-4) .. +31 (B operands)
template <int C>
__global__ __launch_bounds__(256) void pack_ffn_kernel(const float *__restrict__ w1, const float *__restrict__ w2,
                                                       int H, u32x4 *__restrict__ out) {
  constexpr int KB = C / 32, CT = C / 16, SPC = KB + 4;      // steps per chunk
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)(H / 128) * SPC * FFN_WQ;
  if (i >= total) return;
  int lane = (int)(i & 63);
  int part = (int)((i >> 6) & 1);
  int t = (int)((i >> 7) & 7);
  int sj = (int)((i >> 10) % SPC);
  int c = (int)((i >> 10) / SPC);
  int n = lane & 15, g = lane >> 4;
  float wv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float w = 0.f;
    if (sj < KB) {
      int hid = c * 128 + t * 16 + n;            // A operand row m = n
      int ch = sj * 32 + g * 8 + e;
      w = w1[(size_t)hid * C + ch];
    } else if (t < CT) {
      int q = sj - KB;
      int col = n * CT + t;                      // lane n owns CT consecutive output columns
      int hid = c * 128 + (2 * q + (e >> 2)) * 16 + 4 * g + (e & 3);
      w = w2[(size_t)col * H + hid];
    }
    wv[e] = w;
  }
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    unsigned hi, lo;
    split_pair_w(wv[2 * e], wv[2 * e + 1], hi, lo);
    o[e] = part ? lo : hi;
  }
  out[i] = o;
}

struct FfnArgs {
  const float *x;          // [rows, 128]
  const u32x4 *w;          // packed stream
  const float *b1, *b2;    // [H], [128]
  const float *res;        // residual added before the LayerNorm (usually x), or NULL
  const float *ln_g, *ln_b;  // LayerNorm affine, or NULL (no normalisation)
  float eps;
  float *out;
  long long rows;
  int H;
  int dbg;     // tuning experiments (DF3D_FFN_DBG): 1 = no MFMAs, 2 = no weight staging
  int bf16;    // 1: 16-bit operands (activations and weights rounded to fp16 = the hi parts of their split, ONE product per
               //    pair: 11 significand bits where bf16 has 8), fp32 accumulate, fp32 bias / residual / LayerNorm -- the
               //    reduced-precision mode of BASELINE configs[2]
};

// NW waves per workgroup, RT 16-row tiles per wave.  RT = 2 halves the LDS fragment reads per MFMA (every operand
// fragment read from LDS feeds two row tiles) at the price of one resident wave per SIMD.
// up to four independent FFN jobs (same sizes, different weights / rows) in one launch: blockIdx.y picks the job.  The
// dual-query layer runs its image-query and LiDAR-query FFNs this way: twice the workgroups per launch fill the 256
// CUs better than two launches of ~1.2 waves each.
struct FfnJobs {
  FfnArgs s[4];
};

// NP = operand parts: 2 = split precision (hi + lo, three products), 1 = 16-bit (hi parts only, one product; the lo
// halves of the packed stream and of the activations are simply not read)
template <int NW, int RT, int NP = 2, int C = 128>
__global__ __launch_bounds__(NW * 64) void ffn_split_kernel(FfnJobs jobs) {
  constexpr int KB = C / 32, CT = C / 16, SPC = KB + 4;
  const FfnArgs &a = jobs.s[blockIdx.y];
  if ((long long)blockIdx.x * (16 * RT * NW) >= a.rows) return;
  constexpr int NT = NW * 64, WR = 16 * RT, TM = NW * WR;
  constexpr int WPT = FFN_WQ / NT;
  static_assert(FFN_WQ % NT == 0, "tile must divide over the workgroup");
  __shared__ u32x4 Wl[2][FFN_WQ];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;
  const long long wrow0 = (long long)blockIdx.x * TM + wave * WR;
  const int nchunks = a.H / 128;
  const int steps = nchunks * SPC;

  // x fragments of the wave's rows: lane (row n of tile rt, g) holds channels 32kb + 8g .. +7, split hi/lo
  u32x4 xh[RT][KB], xl[RT][KB];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const long long row_n = wrow0 + rt * 16 + n;
    const long long row_ld = row_n < a.rows ? row_n : a.rows - 1;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const float *p = a.x + row_ld * C + kb * 32 + g * 8;
      f32x4 v0 = *(const f32x4 *)p, v1 = *(const f32x4 *)(p + 4);
      split_pair(v0[0], v0[1], xh[rt][kb][0], xl[rt][kb][0]);
      split_pair(v0[2], v0[3], xh[rt][kb][1], xl[rt][kb][1]);
      split_pair(v1[0], v1[1], xh[rt][kb][2], xl[rt][kb][2]);
      split_pair(v1[2], v1[3], xh[rt][kb][3], xl[rt][kb][3]);
    }
  }

  // The stream tile of step s + 1 goes to LDS during step s.  Round 3: it was fetched TWO steps before that (two register
  // sets in turn; SPC is even, so the set of a step is a compile-time constant of the unrolled chunk) -- with one step of
  // ~1000 clocks between the load and its use every step began with the rest of an L2 round trip.
  static_assert(SPC % 2 == 0, "register set = step parity");
  u32x4 wreg[2][WPT];
  auto load_w = [&](int s, int set) {
    s = s < steps ? s : steps - 1;
    const u32x4 *src = a.w + (size_t)s * FFN_WQ;
#pragma unroll
    for (int i = 0; i < WPT; ++i) wreg[set][i] = src[tid + NT * i];
  };
  auto store_w = [&](int buf, int set) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) Wl[buf][tid + NT * i] = wreg[set][i];
  };

  f32x4 acc2[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc2[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  load_w(0, 0);
  store_w(0, 0);
  load_w(1, 1);
  load_w(2, 0);
  for (int c = 0; c < nchunks; ++c) {
    f32x4 acc1[RT][8];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int t = 0; t < 8; ++t) acc1[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // ---- phase 1: H^T chunk = W1_c x^T ----
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const int s = c * SPC + kb;
      __syncthreads();
      if (!(a.dbg & 2)) {
        store_w((s + 1) & 1, (kb + 1) & 1);
        load_w(s + 3, (kb + 1) & 1);
      }
      const u32x4 *wb = Wl[s & 1] + lane;
      if (a.dbg & 1) continue;
      // operand fragments of the next tile pair come from LDS while the MFMAs of the current pair run
      u32x4 fq[2][4];
#pragma unroll
      for (int q = 0; q < 4; q += (NP == 2 ? 1 : 2)) fq[0][q] = wb[q * 64];
#pragma unroll
      for (int t = 0; t < 8; t += 2) {
        const int cur = (t >> 1) & 1;
        if (t + 2 < 8) {
#pragma unroll
          for (int q = 0; q < 4; q += (NP == 2 ? 1 : 2)) fq[cur ^ 1][q] = wb[((t + 2) * 2 + q) * 64];
        }
        const u32x4 ah0 = fq[cur][0], al0 = fq[cur][1], ah1 = fq[cur][2], al1 = fq[cur][3];
        if constexpr (NP == 2) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            acc1[rt][t] = DF3D_MFMA_F16(ah0, xl[rt][kb], acc1[rt][t]);
            acc1[rt][t + 1] = DF3D_MFMA_F16(ah1, xl[rt][kb], acc1[rt][t + 1]);
          }
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            acc1[rt][t] = DF3D_MFMA_F16(al0, xh[rt][kb], acc1[rt][t]);
            acc1[rt][t + 1] = DF3D_MFMA_F16(al1, xh[rt][kb], acc1[rt][t + 1]);
          }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          acc1[rt][t] = DF3D_MFMA_F16(ah0, xh[rt][kb], acc1[rt][t]);
          acc1[rt][t + 1] = DF3D_MFMA_F16(ah1, xh[rt][kb], acc1[rt][t + 1]);
        }
      }
    }
    // ---- bias + ReLU + split: lane (row n, g) holds hidden 128c + 16t + 4g + {0..3} -> phase-2 A operands ----
    u32x4 hh[RT][4], hl[RT][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int t = 2 * q + half;
        const f32x4 b = *(const f32x4 *)(a.b1 + c * 128 + t * 16 + 4 * g);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          // (unchecked: a hidden value beyond fp16's range becomes inf / NaN in the output row, which the next split of
          // that row reports)
          const f32x4 hv = acc1[rt][t] * DF3D_ACC_UNSCALE + b;
          split_pair_nc(fmaxf(hv[0], 0.f), fmaxf(hv[1], 0.f), hh[rt][q][half * 2], hl[rt][q][half * 2]);
          split_pair_nc(fmaxf(hv[2], 0.f), fmaxf(hv[3], 0.f), hh[rt][q][half * 2 + 1], hl[rt][q][half * 2 + 1]);
        }
      }
    }
    // ---- phase 2: Y += H_c W2_c^T ----
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int s = c * SPC + KB + q;
      __syncthreads();
      if (!(a.dbg & 2)) {
        store_w((s + 1) & 1, (KB + q + 1) & 1);
        load_w(s + 3, (KB + q + 1) & 1);
      }
      const u32x4 *wb = Wl[s & 1] + lane;
      if (a.dbg & 1) continue;
      u32x4 fq[2][4];
#pragma unroll
      for (int k = 0; k < 4; k += (NP == 2 ? 1 : 2)) fq[0][k] = wb[k * 64];
#pragma unroll
      for (int ct = 0; ct < CT; ct += 2) {
        const int cur = (ct >> 1) & 1;
        if (ct + 2 < CT) {
#pragma unroll
          for (int k = 0; k < 4; k += (NP == 2 ? 1 : 2)) fq[cur ^ 1][k] = wb[((ct + 2) * 2 + k) * 64];
        }
        const u32x4 bh0 = fq[cur][0], bl0 = fq[cur][1], bh1 = fq[cur][2], bl1 = fq[cur][3];
        if constexpr (NP == 2) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            acc2[rt][ct] = DF3D_MFMA_F16(hl[rt][q], bh0, acc2[rt][ct]);
            acc2[rt][ct + 1] = DF3D_MFMA_F16(hl[rt][q], bh1, acc2[rt][ct + 1]);
          }
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            acc2[rt][ct] = DF3D_MFMA_F16(hh[rt][q], bl0, acc2[rt][ct]);
            acc2[rt][ct + 1] = DF3D_MFMA_F16(hh[rt][q], bl1, acc2[rt][ct + 1]);
          }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          acc2[rt][ct] = DF3D_MFMA_F16(hh[rt][q], bh0, acc2[rt][ct]);
          acc2[rt][ct + 1] = DF3D_MFMA_F16(hh[rt][q], bh1, acc2[rt][ct + 1]);
        }
      }
    }
  }

  // ---- epilogue: lane (n, g) holds rows 4g+r, columns CT*n .. CT*n + CT-1: + b2, + residual, LayerNorm over the row ----
  constexpr int NV = CT / 4;                     // 16-byte vectors per lane and row
  f32x4 bias[NV], gam[NV], bet[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    bias[v] = *(const f32x4 *)(a.b2 + n * CT + 4 * v);
    gam[v] = (f32x4){1.f, 1.f, 1.f, 1.f};
    bet[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.ln_g) {
      gam[v] = *(const f32x4 *)(a.ln_g + n * CT + 4 * v);
      bet[v] = *(const f32x4 *)(a.ln_b + n * CT + 4 * v);
    }
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const long long rbase = wrow0 + rt * 16 + 4 * g;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long long row = rbase + r;
      const bool live = row < a.rows;
      const long long rr = live ? row : a.rows - 1;
      f32x4 val[NV];
      float s = 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        val[v] = (f32x4){acc2[rt][4 * v][r], acc2[rt][4 * v + 1][r], acc2[rt][4 * v + 2][r], acc2[rt][4 * v + 3][r]} *
                     DF3D_ACC_UNSCALE + bias[v];
        if (a.res) val[v] += *(const f32x4 *)(a.res + rr * C + n * CT + 4 * v);
        s += val[v][0] + val[v][1] + val[v][2] + val[v][3];
      }
      if (a.ln_g) {
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);      // the 16 lanes of this row group
        const float mean = s * (1.f / C);
        float ss = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          val[v] = val[v] - mean;
          ss += val[v][0] * val[v][0] + val[v][1] * val[v][1] + val[v][2] * val[v][2] + val[v][3] * val[v][3];
        }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const float rstd = rsqrtf(ss * (1.f / C) + a.eps);
#pragma unroll
        for (int v = 0; v < NV; ++v) val[v] = val[v] * rstd * gam[v] + bet[v];
      }
      if (live) {
#pragma unroll
        for (int v = 0; v < NV; ++v) *(f32x4 *)(a.out + row * C + n * CT + 4 * v) = val[v];
      }
    }
  }
}

}  // namespace df3d

using namespace df3d;

extern "C" size_t df3d_ffn_packed_bytes(int d_model, int d_ffn) {
  if ((d_model != 128 && d_model != 64) || d_ffn <= 0 || d_ffn % 128 != 0) return 0;
  return (size_t)(d_ffn / 128) * (d_model / 32 + 4) * FFN_WQ * 16;
}

extern "C" int df3d_ffn_pack(const float *w1, const float *w2, int d_model, int d_ffn, void *packed, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(w1 && w2 && packed, "ffn_pack: null argument");
  DF3D_CHECK_ARG(df3d_ffn_packed_bytes(d_model, d_ffn) != 0,
                 "ffn_pack: the fused feed-forward kernel serves d_model 64 / 128 and d_ffn %% 128 == 0 (got %d, %d)",
                 d_model, d_ffn);
  size_t total = (size_t)(d_ffn / 128) * (d_model / 32 + 4) * FFN_WQ;
  if (d_model == 128)
    hipLaunchKernelGGL(pack_ffn_kernel<128>, dim3(cdiv((long long)total, 256)), dim3(256), 0, stream, w1, w2, d_ffn,
                       (u32x4 *)packed);
  else
    hipLaunchKernelGGL(pack_ffn_kernel<64>, dim3(cdiv((long long)total, 256)), dim3(256), 0, stream, w1, w2, d_ffn,
                       (u32x4 *)packed);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

static int ffn_launch(const FfnJobs &jobs, int njobs, long long max_rows, int d_model, hipStream_t stream) {
  if (d_model == 64) {                           // 64-wide rows: 8 waves x 16 rows, either precision
    static const int cfg64 = getenv("DF3D_FFN_CFG64") ? atoi(getenv("DF3D_FFN_CFG64")) : 0;
    if (!jobs.s[0].bf16 && cfg64 == 44) {
      hipLaunchKernelGGL((ffn_split_kernel<4, 4, 2, 64>), dim3(cdiv(max_rows, 256), njobs), dim3(256), 0, stream, jobs);
      DF3D_LAUNCH_CHECK();
      return DF3D_OK;
    }
    // (round 5, tools/ubench/ffn_probe64.py: 2 x 160 k rows 390 us with one row tile per wave, 327 us with two -- every
    // weight fragment read from LDS feeds two row tiles --, 370 us with four on one wave per SIMD: the exposed fragment reads
    // of a lone wave cost more than the halved LDS traffic saves)
    if (!jobs.s[0].bf16 && (cfg64 == 82 || (cfg64 == 0 && (long long)cdiv(max_rows, 256) * njobs >= 192))) {
      hipLaunchKernelGGL((ffn_split_kernel<8, 2, 2, 64>), dim3(cdiv(max_rows, 256), njobs), dim3(512), 0, stream, jobs);
      DF3D_LAUNCH_CHECK();
      return DF3D_OK;
    }
    if (jobs.s[0].bf16)
      hipLaunchKernelGGL((ffn_split_kernel<8, 1, 1, 64>), dim3(cdiv(max_rows, 128), njobs), dim3(512), 0, stream, jobs);
    else
      hipLaunchKernelGGL((ffn_split_kernel<8, 1, 2, 64>), dim3(cdiv(max_rows, 128), njobs), dim3(512), 0, stream, jobs);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
  }
  // tuning aid: NW*10 + RT (1081: bf16 mode with one row tile).  Default (round 3, tools/ubench/ffn_probe.py): two row tiles
  // per wave once their 256-row workgroups fill most of the CUs -- every operand fragment read from LDS then feeds two
  // MFMAs, and the LDS pipe (128 KB of fragment reads per step and CU with one row tile: 1024 clocks against 768 of MFMAs) is
  // what bounds this kernel: 2 x 31134 rows 122 -> 104 us
  static const int cfg_env = getenv("DF3D_FFN_CFG") ? atoi(getenv("DF3D_FFN_CFG")) : 0;
  // Round 4: a 256-row workgroup keeps the matrix and LDS pipes of a CU busy by itself; TWO of them on one CU (registers and
  // the 32 KB of LDS allow it) each run at half speed, and the launch takes twice as long as with one per CU.  That happened
  // (a) by count -- 260 workgroups on 256 CUs: 186 us instead of 103 (profiles/r04_start_pmc_by_kernel.json) -- and (b) by
  // placement, whenever another stream's kernels held some CUs at dispatch time and the dispatcher doubled up elsewhere
  // (178 us with 240 workgroups).  (a): more workgroups than CUs -> 128-row workgroups (two per CU share it evenly);
  // (b): the 256-row configuration reserves enough LDS that a second one does not fit beside it (DF3D_FFN_PAD=0: off).
  static int ncu = 0;
  if (!ncu) {
    hipDeviceProp_t prop;
    ncu = (hipGetDeviceProperties(&prop, 0) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  const long long wg82 = (long long)cdiv(max_rows, 256) * njobs;
  static const bool balance = !(getenv("DF3D_FFN_PAD") && getenv("DF3D_FFN_PAD")[0] == '0');
  const int cfg = cfg_env ? cfg_env : (wg82 >= 160 && !(balance && wg82 > ncu) ? 82 : 81);
  if (jobs.s[0].bf16) {                          // every job of a launch shares the precision mode
    // one product per operand pair: the kernel is bound by the 1 MB weight stream every workgroup pulls from L2, so many
    // rows take two row tiles per wave (half the stream per row); DF3D_FFN_CFG=81 keeps one
    if (max_rows >= 32 * 1024 && cfg != 81 + 1000)
      hipLaunchKernelGGL((ffn_split_kernel<8, 2, 1>), dim3(cdiv(max_rows, 256), njobs), dim3(512), 0, stream, jobs);
    else
      hipLaunchKernelGGL((ffn_split_kernel<8, 1, 1>), dim3(cdiv(max_rows, 128), njobs), dim3(512), 0, stream, jobs);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
  }
  if (cfg == 44) hipLaunchKernelGGL((ffn_split_kernel<4, 4>), dim3(cdiv(max_rows, 256), njobs), dim3(256), 0, stream, jobs);
  else if (cfg == 42) hipLaunchKernelGGL((ffn_split_kernel<4, 2>), dim3(cdiv(max_rows, 128), njobs), dim3(256), 0, stream, jobs);
  else if (cfg == 82) {
    size_t pad = 0;
    if (balance && wg82 <= ncu) {
      pad = 52 * 1024;                          // 32 KB static + 52 KB > half of the CU's 160 KB
      static bool attr = false;
      if (!attr) {
        attr = hipFuncSetAttribute((const void *)ffn_split_kernel<8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)pad) == hipSuccess;
        if (!attr) pad = 0;
      }
    }
    hipLaunchKernelGGL((ffn_split_kernel<8, 2>), dim3(cdiv(max_rows, 256), njobs), dim3(512), pad, stream, jobs);
  }
  else if (cfg == 41) hipLaunchKernelGGL((ffn_split_kernel<4, 1>), dim3(cdiv(max_rows, 64), njobs), dim3(256), 0, stream, jobs);
  else hipLaunchKernelGGL((ffn_split_kernel<8, 1>), dim3(cdiv(max_rows, 128), njobs), dim3(512), 0, stream, jobs);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

static int g_ffn_bf16 = 0;
// reduced-precision switch of the fused feed-forward kernel (process-wide, like DF3D_CONV_PRECISION for the convs):
// 0 = split precision (fp32-grade), 1 = bf16 operands with fp32 accumulate
extern "C" int df3d_ffn_set_precision(int bf16) {
  g_ffn_bf16 = bf16 ? 1 : 0;
  return DF3D_OK;
}

extern "C" int df3d_ffn_fused(const float *x, long long rows, int d_model, int d_ffn, const void *packed,
                              const float *b1, const float *b2, const float *residual, const float *ln_weight,
                              const float *ln_bias, float eps, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(x && packed && b1 && b2 && out, "ffn_fused: null argument");
  DF3D_CHECK_ARG(df3d_ffn_packed_bytes(d_model, d_ffn) != 0, "ffn_fused: unsupported sizes d_model=%d d_ffn=%d",
                 d_model, d_ffn);
  DF3D_CHECK_ARG((ln_weight == nullptr) == (ln_bias == nullptr), "ffn_fused: LayerNorm needs weight and bias");
  if (rows <= 0) return DF3D_OK;
  FfnJobs jobs;
  memset(&jobs, 0, sizeof(jobs));
  jobs.s[0] = {x, (const u32x4 *)packed, b1, b2, residual, ln_weight, ln_bias, eps, out, rows, d_ffn,
               getenv("DF3D_FFN_DBG") ? atoi(getenv("DF3D_FFN_DBG")) : 0, g_ffn_bf16};
  return ffn_launch(jobs, 1, rows, d_model, stream);
}

extern "C" int df3d_ffn_fused_jobs(const df3d_ffn_job *j, int njobs, int d_model, int d_ffn, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(j && njobs >= 1 && njobs <= 4, "ffn_fused_jobs: 1..4 jobs");
  DF3D_CHECK_ARG(df3d_ffn_packed_bytes(d_model, d_ffn) != 0, "ffn_fused_jobs: unsupported sizes d_model=%d d_ffn=%d",
                 d_model, d_ffn);
  FfnJobs jobs;
  memset(&jobs, 0, sizeof(jobs));
  long long max_rows = 0;
  static const int dbg = getenv("DF3D_FFN_DBG") ? atoi(getenv("DF3D_FFN_DBG")) : 0;      // tuning aid (see FfnArgs::dbg)
  for (int i = 0; i < njobs; ++i) {
    DF3D_CHECK_ARG(j[i].rows <= 0 || (j[i].x && j[i].packed && j[i].b1 && j[i].b2 && j[i].out),
                   "ffn_fused_jobs: job %d has a null argument", i);
    DF3D_CHECK_ARG((j[i].ln_weight == nullptr) == (j[i].ln_bias == nullptr),
                   "ffn_fused_jobs: LayerNorm needs weight and bias");
    jobs.s[i] = {j[i].x, (const u32x4 *)j[i].packed, j[i].b1, j[i].b2, j[i].residual, j[i].ln_weight, j[i].ln_bias,
                 j[i].eps, j[i].out, j[i].rows > 0 ? j[i].rows : 0, d_ffn, dbg, g_ffn_bf16};
    if (j[i].rows > max_rows) max_rows = j[i].rows;
  }
  if (max_rows <= 0) return DF3D_OK;
  return ffn_launch(jobs, njobs, max_rows, d_model, stream);
}
