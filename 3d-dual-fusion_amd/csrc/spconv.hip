// Fused sparse convolution for gfx950: output-stationary implicit GEMM on fp32 MFMA.
//
// The reference runs, per conv, 27 x {gather kernel -> GEMM -> scatter-add kernel} with a
// D2H sync (TF/mmdet3d/ops/spconv/include/spconv/spconv_ops.h:260-361, reordering.cu.h:22-158)
// and BN / ReLU / residual as separate elementwise passes (CP/det3d/models/backbones/scn.py:
// 76-94).  That moves ~6x the algorithmic bytes.  Here ONE kernel per conv:
//
//   workgroup (4 waves) owns TM = 64*RT consecutive output rows and all COUT columns;
//   for every kernel offset k that has at least one neighbour in the tile
//     for every 32-wide slice of CIN
//        W[k][slice][:] is staged global -> LDS once per workgroup (double buffered, one
//          barrier per step), rows permuted/swizzled so the B-fragment reads are
//          conflict-free ds_read_b32;
//        each lane loads its A fragment (KS contiguous floats of the gathered input row,
//          16-byte vector loads straight from HBM/L2 - no LDS round trip for activations);
//        v_mfma_f32_16x16x4_f32 accumulates into registers (exact fp32, k-ordered fmaf chain);
//   epilogue: + bias, folded BatchNorm (scale, shift), + residual, ReLU, store.
//
// Each output row is written exactly once; each input row is read once per (output, offset)
// pair that uses it: algorithmic bytes = R*CIN*4 + N_out*COUT*4 + 4*K*N_out (nbr table)
// + K*CIN*COUT*4.  The reduction index order inside MFMA is a permutation
// of cin (lane group g owns cin [g*KS, (g+1)*KS) of the slice); fp32 accumulate.
// Roofline: HBM for C <= 32 (AI 3-14 flop/B), fp32 MFMA (157 TF) for C >= 64.
#include <stdlib.h>

#include "common.h"

namespace df3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
  const float *feat;
  const float *w;
  const int32_t *nbr;
  const float *bias, *scale, *shift, *residual;
  float *out;
  int n_in, n_out, K, cin, cout, relu;
};

template <int KS>
struct AFrag {
  float v[KS];
};

// CINP: cin rounded up to a multiple of KC.  KC: cin slice per step (8, 16 or 32).
// RT: 16-row tiles per wave.  VEC: cin % 4 == 0 (16-byte A loads).
template <int CINP, int COUT, int KC, int RT, bool VEC>
__global__ __launch_bounds__(256) void spconv_mfma_kernel(ConvArgs a) {
  constexpr int KS = KC / 4;          // k-steps per slice = floats per lane per row
  constexpr int NCH = CINP / KC;      // slices
  constexpr int CT = COUT / 16;       // column tiles
  constexpr int TM = 64 * RT;         // rows per workgroup
  constexpr int WROWS = 16 * RT;      // rows per wave
  constexpr int WT = KC * COUT;       // floats per W tile
  constexpr int WPT = (WT / 4 + 255) / 256;  // float4 per thread for staging
  constexpr bool SWZ = COUT >= 32;

  __shared__ float Wl[2][WT];
  __shared__ int nbrL[DF3D_MAX_KVOL][TM];
  __shared__ unsigned wg_mask;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;

  // XCD-aware tile order: consecutive tiles (which share neighbour rows) stay on one XCD/L2
  int nt = gridDim.x;
  int bid = blockIdx.x;
  int tile = bid;
  if ((nt & 7) == 0) tile = (bid & 7) * (nt >> 3) + (bid >> 3);
  const int row0 = tile * TM;

  if (tid == 0) wg_mask = 0u;
  for (int e = tid; e < a.K * TM; e += 256) {
    int k = e / TM, r = e - k * TM;
    int row = row0 + r;
    nbrL[k][r] = (row < a.n_out) ? a.nbr[(size_t)k * a.n_out + row] : -1;
  }
  __syncthreads();
  // per-wave activity mask over k (wave-uniform), workgroup mask = OR
  unsigned wmask = 0u;
  for (int k = 0; k < a.K; ++k) {
    bool any = false;
#pragma unroll
    for (int q = 0; q < (WROWS + 63) / 64; ++q) {
      int r = q * 64 + lane;
      int v = (r < WROWS) ? nbrL[k][wave * WROWS + r] : -1;
      any |= (__ballot(v >= 0) != 0ull);
    }
    if (any) wmask |= (1u << k);
  }
  if (lane == 0 && wmask) atomicOr(&wg_mask, wmask);
  __syncthreads();
  const unsigned gmask = wg_mask;
  const int nact = __popc(gmask);
  const int steps = nact * NCH;

  f32x4 acc[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- helpers -------------------------------------------------------------------
  auto kth_active = [&](int ai) -> int {  // index of the ai-th set bit of gmask
    unsigned m = gmask;
    for (int i = 0; i < ai; ++i) m &= m - 1;
    return __ffs((int)m) - 1;
  };
  f32x4 wreg[WPT];
  auto load_w = [&](int k, int ch) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      int e = tid + 256 * i;
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (e < WT / 4) {
        int r = e / (COUT / 4), c4 = e - r * (COUT / 4);
        int ci = ch * KC + r;
        if (ci < a.cin) v = *(const f32x4 *)(a.w + ((size_t)k * a.cin + ci) * COUT + c4 * 4);
      }
      wreg[i] = v;
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      int e = tid + 256 * i;
      if (e < WT / 4) {
        int r = e / (COUT / 4), c4 = e - r * (COUT / 4);
        int rp = 4 * (r % KS) + (r / KS);  // global row g*KS+j -> LDS row 4j+g
        int col = c4 * 4;
        if (SWZ) col ^= ((rp & 1) << 4);
        *(f32x4 *)(&Wl[buf][rp * COUT + col]) = wreg[i];
      }
    }
  };
  AFrag<KS> anext[RT], acur[RT];
  auto load_a = [&](int k, int ch) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      int idx = nbrL[k][wave * WROWS + rt * 16 + n];
      int c0 = ch * KC + g * KS;
      if (VEC) {
#pragma unroll
        for (int q = 0; q < KS / 4; ++q) {
          f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (idx >= 0) v = *(const f32x4 *)(a.feat + (size_t)idx * a.cin + c0 + q * 4);
          anext[rt].v[q * 4 + 0] = v[0];
          anext[rt].v[q * 4 + 1] = v[1];
          anext[rt].v[q * 4 + 2] = v[2];
          anext[rt].v[q * 4 + 3] = v[3];
        }
      } else {
#pragma unroll
        for (int j = 0; j < KS; ++j) {
          float v = 0.f;
          if (idx >= 0 && c0 + j < a.cin) v = a.feat[(size_t)idx * a.cin + c0 + j];
          anext[rt].v[j] = v;
        }
      }
    }
  };

  // ---- software pipeline over (active offset, cin slice) ---------------------------
  int k_next = 0;
  if (steps > 0) {
    k_next = kth_active(0);
    load_w(k_next, 0);
    store_w(0);
    if ((wmask >> k_next) & 1u) load_a(k_next, 0);
  }
  for (int s = 0; s < steps; ++s) {
    const int k_cur = k_next;
    const bool wave_on = (wmask >> k_cur) & 1u;
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acur[rt] = anext[rt];
    const bool more = s + 1 < steps;
    int ch_n = 0;
    if (more) {
      int s1 = s + 1;
      int ai = s1 / NCH;
      ch_n = s1 - ai * NCH;
      k_next = (ch_n == 0) ? kth_active(ai) : k_cur;
      load_w(k_next, ch_n);
      if ((wmask >> k_next) & 1u) load_a(k_next, ch_n);
    }
    if (wave_on) {
      const float *wb = Wl[s & 1];
#pragma unroll
      for (int j = 0; j < KS; ++j) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          int rp = 4 * j + g;
          int col = ct * 16 + n;
          if (SWZ) col ^= ((g & 1) << 4);
          float b = wb[rp * COUT + col];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[rt].v[j], b, acc[rt][ct], 0, 0, 0);
        }
      }
    }
    if (more) store_w((s + 1) & 1);
  }

  // ---- epilogue: bias, folded BN, residual, ReLU -----------------------------------
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    int col = ct * 16 + n;
    float bi = a.bias ? a.bias[col] : 0.f;
    float sc = a.scale ? a.scale[col] : 1.f;
    float sh = a.shift ? a.shift[col] : 0.f;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = row0 + wave * WROWS + rt * 16 + 4 * g + r;
        if (row < a.n_out) {
          float v = (acc[rt][ct][r] + bi) * sc + sh;
          if (a.residual) v += a.residual[(size_t)row * COUT + col];
          if (a.relu) v = fmaxf(v, 0.f);
          a.out[(size_t)row * COUT + col] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// v2 for the compute-bound layers (COUT >= 64): pair-compacted MFMA rows, LDS accumulators.
//
// The output-stationary kernel above issues MFMAs for every (row, offset) of its tile, also
// where the neighbour is absent (42 % of the slots at conv4's occupancy).  Here a workgroup
// still owns TM consecutive output rows, but
//   * its accumulators live in LDS ([wave][TM][COUT/4] fp32), so MFMA rows need not be output
//     rows: for each kernel offset the VALID (output row, input row) pairs of the tile are
//     compacted (ballot + popcount into a per-wave scratch list) and processed 16 at a time;
//     only the last chunk of an offset is partially filled;
//   * each of the 4 waves owns a quarter of the output columns, keeps the matching
//     [CIN x COUT/4] slice of W[k] in REGISTERS as MFMA B operands (loaded once per offset from
//     L2, 8-byte loads) and accumulates its slice privately: there is NO barrier in the main loop;
//   * A fragments (CIN/4 contiguous floats of the gathered input row per lane) are loaded
//     straight from HBM/L2 with 16-byte loads, one chunk ahead of the MFMAs;
//   * per chunk the 16 x COUT/4 product is added into the LDS accumulators (8-byte RMW, each
//     output row appears at most once per offset, so no atomics);
//   * epilogue from LDS: bias, folded BN, residual, ReLU, 16-byte stores.
template <int CIN, int COUT, int TM>
__global__ __launch_bounds__(256) void spconv_pair_kernel(ConvArgs a) {
  constexpr int KS = CIN / 4;        // k-steps = floats of one input row held by a lane
  constexpr int CS = COUT / 4;       // output columns per wave
  constexpr int CT = CS / 16;        // 16-wide column tiles per wave (1 or 2)
  constexpr int NCH = 2 / CT;        // chunks in flight -> always 2 independent accumulators
  static_assert(CT == 1 || CT == 2, "COUT must be 64 or 128");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *accL = (float *)smem;                                  // [4][TM][CS]
  int *nbrL = (int *)(smem + (size_t)4 * TM * CS * 4);          // [K][TM]
  unsigned short *listL = (unsigned short *)(nbrL + a.K * TM);  // [4][TM]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;
  int nt = gridDim.x, bid = blockIdx.x, tile = bid;
  if ((nt & 7) == 0) tile = (bid & 7) * (nt >> 3) + (bid >> 3);
  const int row0 = tile * TM;
  const int cs0 = wave * CS;

  for (int e = tid; e < a.K * TM; e += 256) {
    int k = e / TM, r = e - k * TM;
    int row = row0 + r;
    nbrL[e] = (row < a.n_out) ? a.nbr[(size_t)k * a.n_out + row] : -1;
  }
  float *myacc = accL + (size_t)wave * TM * CS;
  for (int e = lane; e < TM * CS / 4; e += 64) ((f32x4 *)myacc)[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  unsigned short *mylist = listL + wave * TM;
  __syncthreads();

  for (int k = 0; k < a.K; ++k) {
    // ---- compact the valid rows of this offset (identical in every wave) ----
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < TM / 64; ++q) {
      int r = q * 64 + lane;
      bool valid = nbrL[k * TM + r] >= 0;
      unsigned long long m = __ballot(valid);
      int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
      if (valid) mylist[pos] = (unsigned short)r;
      cnt += __popcll(m);
    }
    if (cnt == 0) continue;
    __builtin_amdgcn_wave_barrier();
    // ---- this wave's slice of W[k] as B operands ----
    float b[KS][CT];
    {
      const float *wk = a.w + ((size_t)k * CIN + (size_t)g * KS) * COUT + cs0;
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        if (CT == 2) {
          float2 v = *(const float2 *)(wk + (size_t)j * COUT + 2 * n);
          b[j][0] = v.x;
          b[j][CT - 1] = v.y;
        } else {
          b[j][0] = wk[(size_t)j * COUT + n];
        }
      }
    }
    const int nchunk = (cnt + 15) >> 4;
    const int ngroup = (nchunk + NCH - 1) / NCH;
    float acur[NCH][KS], anext[NCH][KS];
    auto load_a = [&](int grp, float (&dst)[NCH][KS]) {
#pragma unroll
      for (int h = 0; h < NCH; ++h) {
        int p = (grp * NCH + h) * 16 + n;
        int idx = -1;
        if (p < cnt) idx = nbrL[k * TM + mylist[p]];
        const float *src = a.feat + (size_t)(idx < 0 ? 0 : idx) * CIN + g * KS;
#pragma unroll
        for (int q = 0; q < KS / 4; ++q) {
          f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (idx >= 0) v = *(const f32x4 *)(src + q * 4);
          dst[h][q * 4 + 0] = v[0];
          dst[h][q * 4 + 1] = v[1];
          dst[h][q * 4 + 2] = v[2];
          dst[h][q * 4 + 3] = v[3];
        }
      }
    };
    load_a(0, anext);
    for (int grp = 0; grp < ngroup; ++grp) {
#pragma unroll
      for (int h = 0; h < NCH; ++h)
#pragma unroll
        for (int j = 0; j < KS; ++j) acur[h][j] = anext[h][j];
      if (grp + 1 < ngroup) load_a(grp + 1, anext);
      f32x4 acc[NCH][CT];
#pragma unroll
      for (int h = 0; h < NCH; ++h)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[h][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < KS; ++j)
#pragma unroll
        for (int h = 0; h < NCH; ++h)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
            acc[h][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[h][j], b[j][ct], acc[h][ct], 0, 0, 0);
      // ---- add the 16 x CS products into the LDS accumulators (row = pair slot 4g + r) ----
#pragma unroll
      for (int h = 0; h < NCH; ++h) {
        int pbase = (grp * NCH + h) * 16 + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int p = pbase + r;
          if (p < cnt) {
            int rl = mylist[p];
            if (CT == 2) {
              float2 *dst = (float2 *)(myacc + (size_t)rl * CS + 2 * n);
              float2 v = *dst;
              v.x += acc[h][0][r];
              v.y += acc[h][CT - 1][r];
              *dst = v;
            } else {
              myacc[(size_t)rl * CS + n] += acc[h][0][r];
            }
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }

  // ---- epilogue: this wave's CS columns of every row of the tile ----
  constexpr int LPR = CS / 4;          // lanes per row (float4 each)
  constexpr int RPI = 64 / LPR;        // rows per iteration
  const int lr = lane / LPR, lc = (lane % LPR) * 4;
  f32x4 bi = (f32x4){0.f, 0.f, 0.f, 0.f}, sc = (f32x4){1.f, 1.f, 1.f, 1.f}, sh = bi;
  if (a.bias) bi = *(const f32x4 *)(a.bias + cs0 + lc);
  if (a.scale) sc = *(const f32x4 *)(a.scale + cs0 + lc);
  if (a.shift) sh = *(const f32x4 *)(a.shift + cs0 + lc);
  for (int r0 = 0; r0 < TM; r0 += RPI) {
    int rl = r0 + lr;
    int row = row0 + rl;
    if (row < a.n_out) {
      f32x4 v = *(const f32x4 *)(myacc + (size_t)rl * CS + lc);
      v = (v + bi) * sc + sh;
      size_t o = (size_t)row * COUT + cs0 + lc;
      if (a.residual) v += *(const f32x4 *)(a.residual + o);
      if (a.relu) {
        v[0] = fmaxf(v[0], 0.f);
        v[1] = fmaxf(v[1], 0.f);
        v[2] = fmaxf(v[2], 0.f);
        v[3] = fmaxf(v[3], 0.f);
      }
      *(f32x4 *)(a.out + o) = v;
    }
  }
}

template <int CIN, int COUT, int TM>
static int launch_pair(const ConvArgs &a, hipStream_t stream) {
  size_t lds = (size_t)4 * TM * (COUT / 4) * 4 + (size_t)a.K * TM * 4 + (size_t)4 * TM * 2;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute((const void *)spconv_pair_kernel<CIN, COUT, TM>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS) failed: %s", hipGetErrorString(e));
      return DF3D_EHIP;
    }
    configured = true;
  }
  int nt = cdiv(a.n_out, TM);
  hipLaunchKernelGGL((spconv_pair_kernel<CIN, COUT, TM>), dim3(nt), dim3(256), lds, stream, a);
  return DF3D_OK;
}

// returns 1 if handled, 0 if not applicable, <0 on error
static int dispatch_pair(const ConvArgs &a, hipStream_t stream) {
  int rc = 0;
  if (a.cout == 128) {
    if (a.cin == 128) rc = launch_pair<128, 128, 128>(a, stream);
    else if (a.cin == 64) rc = launch_pair<64, 128, 128>(a, stream);
    else return 0;
  } else if (a.cout == 64) {
    if (a.cin == 64) rc = launch_pair<64, 64, 256>(a, stream);
    else if (a.cin == 32) rc = launch_pair<32, 64, 256>(a, stream);
    else return 0;
  } else {
    return 0;
  }
  return rc < 0 ? rc : 1;
}

// Any-shape fallback (correctness path for channel counts outside the tuned table).
__global__ __launch_bounds__(256) void spconv_generic_kernel(ConvArgs a) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)a.n_out * a.cout) return;
  int o = (int)(t / a.cout), co = (int)(t - (size_t)o * a.cout);
  float acc = 0.f;
  for (int k = 0; k < a.K; ++k) {
    int idx = a.nbr[(size_t)k * a.n_out + o];
    if (idx < 0) continue;
    const float *f = a.feat + (size_t)idx * a.cin;
    const float *w = a.w + (size_t)k * a.cin * a.cout + co;
    for (int ci = 0; ci < a.cin; ++ci) acc = fmaf(f[ci], w[(size_t)ci * a.cout], acc);
  }
  float v = (acc + (a.bias ? a.bias[co] : 0.f)) * (a.scale ? a.scale[co] : 1.f) + (a.shift ? a.shift[co] : 0.f);
  if (a.residual) v += a.residual[t];
  if (a.relu) v = fmaxf(v, 0.f);
  a.out[t] = v;
}

template <int CINP, int COUT, int KC, bool VEC>
static void launch_mfma(const ConvArgs &a, hipStream_t stream) {
  // small layers: 64-row tiles so that the grid still covers the 256 CUs several times
  if (a.n_out < 128 * 1024) {
    int nt = cdiv(a.n_out, 64);
    hipLaunchKernelGGL((spconv_mfma_kernel<CINP, COUT, KC, 1, VEC>), dim3(nt), dim3(256), 0, stream, a);
  } else {
    int nt = cdiv(a.n_out, 128);
    hipLaunchKernelGGL((spconv_mfma_kernel<CINP, COUT, KC, 2, VEC>), dim3(nt), dim3(256), 0, stream, a);
  }
}

template <int COUT>
static bool dispatch_cin(const ConvArgs &a, hipStream_t stream) {
  if (a.cin <= 8) {
    if (a.cin % 4 == 0) launch_mfma<8, COUT, 8, false>(a, stream);
    else launch_mfma<8, COUT, 8, false>(a, stream);
    return true;
  }
  switch (a.cin) {
    case 16: launch_mfma<16, COUT, 16, true>(a, stream); return true;
    case 32: launch_mfma<32, COUT, 32, true>(a, stream); return true;
    case 64: launch_mfma<64, COUT, 32, true>(a, stream); return true;
    case 128: launch_mfma<128, COUT, 32, true>(a, stream); return true;
    default: return false;
  }
}

}  // namespace df3d

using namespace df3d;

extern "C" int df3d_sparse_conv_fused(const float *features, int n_in, int cin, const float *filters, int kvol,
                                      int cout, const int32_t *nbr, int n_out, const float *bias, const float *scale,
                                      const float *shift, const float *residual, int relu, float *out,
                                      void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(features && filters && nbr && out, "sparse_conv_fused: null argument");
  DF3D_CHECK_ARG(kvol > 0 && kvol <= DF3D_MAX_KVOL, "sparse_conv_fused: kernel volume %d unsupported", kvol);
  DF3D_CHECK_ARG(cin > 0 && cout > 0 && n_in >= 0 && n_out >= 0, "sparse_conv_fused: bad sizes");
  if (n_out == 0) return DF3D_OK;
  ConvArgs a;
  a.feat = features;
  a.w = filters;
  a.nbr = nbr;
  a.bias = bias;
  a.scale = scale;
  a.shift = shift;
  a.residual = residual;
  a.out = out;
  a.n_in = n_in;
  a.n_out = n_out;
  a.K = kvol;
  a.cin = cin;
  a.cout = cout;
  a.relu = relu;
  bool done = false;
  // v2 (pair-compacted rows) is correct but not yet faster than v1 at nuScenes sizes (too few,
  // too large tiles for 256 CUs); opt-in until its tile scheduling is reworked.
  static const bool use_v2 = getenv("DF3D_SPCONV_V2") != nullptr;
  if (use_v2) {
    int r = dispatch_pair(a, stream);
    if (r < 0) return r;
    done = r == 1;
  }
  if (!done) switch (cout) {
    case 16: done = dispatch_cin<16>(a, stream); break;
    case 32: done = dispatch_cin<32>(a, stream); break;
    case 64: done = dispatch_cin<64>(a, stream); break;
    case 128: done = dispatch_cin<128>(a, stream); break;
    default: break;
  }
  if (!done) {
    size_t tot = (size_t)n_out * cout;
    hipLaunchKernelGGL(spconv_generic_kernel, dim3(cdiv((long long)tot, 256)), dim3(256), 0, stream, a);
  }
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}
