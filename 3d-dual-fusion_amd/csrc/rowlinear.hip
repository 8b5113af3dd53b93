// Small dense layers over query rows on the matrix cores (round 3): y = W x + b for [rows, 128 | 256] fp32 rows and
// <= 128 outputs, fp32-grade (operands split into fp16 hi + lo ON LOAD -- common.h --, three MFMA products, fp32 accumulate).
//
// Reference: the query-side linears of the dual-query fusion encoder layer --
//   sampling_offsets / attention_weights of MSDeformAttn on the mixed queries (CP/det3d/models/model_utils/ops/modules/
//   ms_deform_attn.py:129-157 with q_method 'sum': offsets from query + pos, weights from (query + pos) + (i_query + pos),
//   actr_transformer.py:399-411), output_proj + dropout + residual + norm1 (ms_deform_attn.py:188, actr_transformer.py:
//   412-414) and i_input_proj's 1 x 1 Conv1d on the gathered image features (actr.py:96-104,166-170)
// -- which the reference (and rounds 1-2 here) run as library GEMMs: tall-skinny fp32 products that hipBLASLt serves at
// ~25 TFLOP/s (26 + 11 + 20 us per layer and 48 us for the image queries at 18.6 k rows), each behind an element-wise
// pass that writes its operand (q + pos ...) to HBM first.
//
// One kernel, three uses:
//   * operands formed on load: a0 = x0 (+ x2), a1 = a0 + (x1 + x2); column tiles < csplit multiply a0, the others a1 --
//     the two query mixtures of the sampler's linears never exist in memory (replaces df3d_actr_prep + two GEMMs);
//   * the outputs of the two column ranges go to two tensors (offsets [rows, 64], logits [rows, 32]);
//   * optional epilogue out = LayerNorm(res + y) over a full 128-column row (output_proj + norm1).
// A workgroup stages the whole packed filter bank in LDS once (<= 128 KB) and its four waves walk 16-row tiles; the A
// operand of a 32-channel block is 8 consecutive floats of the lane's row, split with the hardware converter.
#include "common.h"

namespace df3d {

typedef float rl_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int rl_u32x4 __attribute__((ext_vector_type(4)));

#define RL_MFMA(A, B, C) DF3D_MFMA_F16(A, B, C)

struct RowLinArgs {
  const float *x0, *x1, *x2;      // [rows, cin]; x1 / x2 may be null
  const rl_u32x4 *w;              // packed [kb][ct][hi|lo][lane] (column ct * 16 + n, channels kb * 32 + g * 8 ..): fp16 pairs of 2^7 w
  const float *bias;              // [ct * 16] or null
  float *out0, *out1;             // columns [0, n0) -> out0 (row stride ld0), [n0, n0 + n1) -> out1
  const float *ln_res, *ln_gamma, *ln_beta;   // LayerNorm(ln_res + y) over n0 == CT * 16 columns, or null
  long long rows;
  int csplit, ld0, n0, ld1, n1;
  float eps;
};

__device__ __forceinline__ void rl_split8(rl_f32x4 a, rl_f32x4 b, rl_u32x4 &hi, rl_u32x4 &lo) {
  // (unchecked: query rows beyond fp16's range leave as inf / NaN rows, which the next checked split reports)
  split_pair_nc(a[0], a[1], hi[0], lo[0]);
  split_pair_nc(a[2], a[3], hi[1], lo[1]);
  split_pair_nc(b[0], b[1], hi[2], lo[2]);
  split_pair_nc(b[2], b[3], hi[3], lo[3]);
}

template <int CIN, int CT>
__global__ __launch_bounds__(256, 2) void rows_linear_kernel(RowLinArgs a) {
  constexpr int KB = CIN / 32;
  constexpr int WQ = KB * CT * 2 * 64;
  extern __shared__ __align__(16) unsigned char rl_smem[];
  rl_u32x4 *Wl = (rl_u32x4 *)rl_smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;
  {                                                // filter bank -> LDS, eight 16-byte loads in flight per thread
    constexpr int PER = WQ / 256;
    static_assert(WQ % 256 == 0 && PER % 4 == 0, "filter bank size");
#pragma unroll 1
    for (int b = 0; b < PER; b += 8) {
      rl_u32x4 t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = a.w[tid + 256 * ((b + i) < PER ? (b + i) : 0)];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (b + i < PER) Wl[tid + 256 * (b + i)] = t[i];
    }
  }
  __syncthreads();
  const long long ntiles = (a.rows + 15) / 16;
  for (long long tile = (long long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long long)gridDim.x * 4) {
    long long row = tile * 16 + n;
    row = row < a.rows ? row : a.rows - 1;
    const float *p0 = a.x0 + row * CIN + g * 8;
    const float *p1 = a.x1 ? a.x1 + row * CIN + g * 8 : nullptr;
    const float *p2 = a.x2 ? a.x2 + row * CIN + g * 8 : nullptr;
    rl_f32x4 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ct] = (rl_f32x4){0.f, 0.f, 0.f, 0.f};
    // one 32-channel block at a time (the loads of the next block are issued before this block's matrix work; more
    // unrolling only spills: the first version held the whole row and the filter fragments in registers, 256 VGPRs + scratch)
    rl_f32x4 u0 = *(const rl_f32x4 *)p0, u1 = *(const rl_f32x4 *)(p0 + 4), v0 = u0, v1 = u0, w0 = u0, w1 = u0;
    if (p2) {
      w0 = *(const rl_f32x4 *)p2;
      w1 = *(const rl_f32x4 *)(p2 + 4);
    }
    if (p1) {
      v0 = *(const rl_f32x4 *)p1;
      v1 = *(const rl_f32x4 *)(p1 + 4);
    }
#pragma unroll 1
    for (int kb = 0; kb < KB; ++kb) {
      rl_f32x4 a00 = u0, a01 = u1, b0 = v0, b1 = v1;
      if (p2) {
        a00 += w0;
        a01 += w1;
        b0 += w0;
        b1 += w1;
      }
      if (kb + 1 < KB) {                             // next block's operands
        u0 = *(const rl_f32x4 *)(p0 + (kb + 1) * 32);
        u1 = *(const rl_f32x4 *)(p0 + (kb + 1) * 32 + 4);
        if (p2) {
          w0 = *(const rl_f32x4 *)(p2 + (kb + 1) * 32);
          w1 = *(const rl_f32x4 *)(p2 + (kb + 1) * 32 + 4);
        }
        if (p1) {
          v0 = *(const rl_f32x4 *)(p1 + (kb + 1) * 32);
          v1 = *(const rl_f32x4 *)(p1 + (kb + 1) * 32 + 4);
        }
      }
      rl_u32x4 h0, l0, h1, l1;
      rl_split8(a00, a01, h0, l0);
      h1 = h0, l1 = l0;
      if (p1) rl_split8(a00 + b0, a01 + b1, h1, l1);  // (x0 + x2) + (x1 + x2): the reference's association
      const rl_u32x4 *wb = Wl + (size_t)kb * CT * 128 + lane;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const rl_u32x4 bh = wb[(ct * 2) * 64], bl = wb[(ct * 2 + 1) * 64];
        const bool second = ct >= a.csplit;
        const rl_u32x4 ah = second ? h1 : h0, al = second ? l1 : l0;
        acc[ct] = RL_MFMA(al, bh, acc[ct]);
        acc[ct] = RL_MFMA(ah, bl, acc[ct]);
        acc[ct] = RL_MFMA(ah, bh, acc[ct]);
      }
    }
    // epilogue: lane (n, g) holds rows 4g .. 4g+3 of the tile, column ct * 16 + n
    float bia[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) bia[ct] = a.bias ? a.bias[ct * 16 + n] : 0.f;
    if (a.ln_res) {
      float ga[CT], be[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        ga[ct] = a.ln_gamma[ct * 16 + n];
        be[ct] = a.ln_beta[ct * 16 + n];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long ro = tile * 16 + 4 * g + r;
        const bool live = ro < a.rows;
        float y[CT], s = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          y[ct] = acc[ct][r] * DF3D_ACC_UNSCALE + bia[ct] + (live ? a.ln_res[ro * (CT * 16) + ct * 16 + n] : 0.f);
          s += y[ct];
        }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s / (float)(CT * 16);
        float q = 0.f;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) q += (y[ct] - mean) * (y[ct] - mean);
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) q += __shfl_xor(q, o, 64);
        const float inv = rsqrtf(q / (float)(CT * 16) + a.eps);
        if (live) {
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) a.out0[ro * a.ld0 + ct * 16 + n] = (y[ct] - mean) * inv * ga[ct] + be[ct];
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long ro = tile * 16 + 4 * g + r;
        if (ro >= a.rows) continue;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const int col = ct * 16 + n;
          const float y = acc[ct][r] * DF3D_ACC_UNSCALE + bia[ct];
          if (col < a.n0) a.out0[ro * a.ld0 + col] = y;
          else if (col < a.n0 + a.n1) a.out1[ro * a.ld1 + (col - a.n0)] = y;
        }
      }
    }
  }
}

template <int CIN, int CT>
static int launch_rows_linear(const RowLinArgs &a, hipStream_t stream) {
  constexpr size_t lds = (size_t)(CIN / 32) * CT * 2 * 64 * 16;
  static_assert(lds <= 128 * 1024, "filter bank must fit LDS");
  static bool configured = false;
  if (!configured && lds > 64 * 1024) {
    DF3D_HIP(hipFuncSetAttribute((const void *)rows_linear_kernel<CIN, CT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    configured = true;
  }
  const long long ntiles = (a.rows + 15) / 16;
  long long wgs = (ntiles + 3) / 4;
  if (wgs > 1024) wgs = 1024;
  hipLaunchKernelGGL((rows_linear_kernel<CIN, CT>), dim3((unsigned)wgs), dim3(256), lds, stream, a);
  return DF3D_OK;
}

}  // namespace df3d

using namespace df3d;

extern "C" size_t df3d_rows_linear_packed_bytes(int cin, int cout) {
  if ((cin != 128 && cin != 256) || cout <= 0 || cout > 128) return 0;
  const int ct = (cout + 15) / 16;
  if (ct != 2 && ct != 4 && ct != 6 && ct != 8) return 0;
  return (size_t)(cin / 32) * ct * 2 * 64 * 16;
}

extern "C" int df3d_rows_linear(const float *x0, const float *x1, const float *x2, long long rows, int cin, const void *packed,
                                int cout, int csplit_cols, const float *bias, float *out0, int ld0, int n0, float *out1, int ld1,
                                int n1, const float *ln_res, const float *ln_gamma, const float *ln_beta, float eps,
                                void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(df3d_rows_linear_packed_bytes(cin, cout) != 0, "rows_linear: cin %d -> cout %d is not served", cin, cout);
  DF3D_CHECK_ARG(rows >= 0 && x0 && packed && out0 && n0 >= 0 && n1 >= 0 && n0 + n1 <= ((cout + 15) / 16) * 16 && (n1 == 0 || out1),
                 "rows_linear: bad arguments");
  DF3D_CHECK_ARG(csplit_cols % 16 == 0 && (x1 || csplit_cols >= cout), "rows_linear: the operand switch must sit on a 16-column tile");
  if (rows == 0) return DF3D_OK;
  const int ct = (cout + 15) / 16;
  if (ln_res) DF3D_CHECK_ARG(ln_gamma && ln_beta && n0 == ct * 16 && n1 == 0 && ld0 >= n0, "rows_linear: LayerNorm needs the full row");
  RowLinArgs a = {x0, x1, x2, (const rl_u32x4 *)packed, bias, out0, out1, ln_res, ln_gamma, ln_beta, rows,
                  csplit_cols / 16, ld0, n0, ld1, n1, eps};
  int rc;
  if (cin == 128) {
    rc = ct == 2 ? launch_rows_linear<128, 2>(a, stream) : ct == 4 ? launch_rows_linear<128, 4>(a, stream)
       : ct == 6 ? launch_rows_linear<128, 6>(a, stream) : launch_rows_linear<128, 8>(a, stream);
  } else {
    rc = ct == 2 ? launch_rows_linear<256, 2>(a, stream) : ct == 4 ? launch_rows_linear<256, 4>(a, stream)
       : ct == 6 ? launch_rows_linear<256, 6>(a, stream) : launch_rows_linear<256, 8>(a, stream);
  }
  if (rc) return rc;
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}
