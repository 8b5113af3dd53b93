// Image side of the dual-query fusion on the 16-bit matrix cores (split precision: fp16 hi + lo operands, see common.h).
//
// The reference runs, per sample, over six [256, 150, 267] camera maps: the image gate's 1x1 summary
// (attention.py:456), ACTR's input_proj 1x1 conv + GroupNorm (actr.py:139-149) and, per encoder layer, value_proj
// over every pixel (ms_deform_attn.py:139).  dualfusion/fusion.py collapses that to two GEMMs (DESIGN.md section 3
// "image side"); hipBLASLt runs them in fp32 at ~90-100 TFLOP/s (173 + 175 us).  Here they are
//
//   img_proj_split_kernel   u[p][0:144] = Wcat[144x256] . img[:, p]   from the channel-first fp32 maps in place:
//                           a [32 k][128 pixel] tile is staged coalesced through LDS and read back transposed
//                           (8 k-values per lane = one MFMA B operand after the hi/lo split), Wcat is the packed A
//                           operand, so the accumulators hold u TRANSPOSED: lane (pixel, g) -> 4 consecutive
//                           channels per 16-channel tile, i.e. pixel-major rows.  Output: split rows of the 128
//                           projection channels (the A operand of the next GEMM) + the fp32 gate column.
//   split_moments_kernel    per (image, channel) sum_p a_p u_cp and sum_p (a_p u_cp)^2 from the split rows
//   gn_fold_pack_kernel     GroupNorm statistics -> per-image folded value weights, written directly as packed
//                           B operands
//   rows_gemm_split_kernel  value[p][0:256] = u_p . Wf[n]^T  (both encoder layers at once), rows contiguous
//
// Bytes: img 246 MB read once, u 123 MB written + read twice, value 246 MB written; 2*(17.7 + 15.7) GFLOP.
#include <stdlib.h>

#include "common.h"

namespace df3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

DF3D_SPLIT_OVERFLOW_TU(imgproj)

constexpr int IP_CIN = 256;          // camera feature channels
constexpr int IP_MT = 9;             // 16-row tiles of Wcat: 128 projection rows + gate row + padding = 144
constexpr int IP_C = 128;            // projection channels
constexpr int IP_TP = 128;           // pixels per workgroup (8 waves x 16)
constexpr int IP_LD = 130;           // LDS row pitch of the image tile (floats): 8*LD = 16 (mod 32) -> the four
                                     // k-groups of a ds_read_b32 hit different banks
constexpr int IP_WQ = IP_MT * 2 * 64;  // u32x4 per packed Wcat step tile

// Wcat [144][256] fp32 -> [kb 8][t 9][hi|lo][lane 64] u32x4: lane (m, g) = row 16t+m, channels 32kb + 8g + e
__global__ __launch_bounds__(256) void pack_proj_kernel(const float *__restrict__ w, int rows, u32x4 *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 8 * IP_WQ) return;
  int lane = i & 63, part = (i >> 6) & 1;
  int t = (i >> 7) % IP_MT, kb = (i >> 7) / IP_MT;
  int m = lane & 15, g = lane >> 4;
  int row = 16 * t + m;
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float x0 = row < rows ? w[(size_t)row * IP_CIN + kb * 32 + g * 8 + 2 * e] : 0.f;
    const float x1 = row < rows ? w[(size_t)row * IP_CIN + kb * 32 + g * 8 + 2 * e + 1] : 0.f;
    unsigned hi, lo;
    split_pair_w(x0, x1, hi, lo);
    o[e] = part ? lo : hi;
  }
  out[i] = o;
}

struct ProjArgs2 {
  const float *const *img;   // NI pointers to [256][S] fp32
  const u32x4 *w;            // packed Wcat
  u32x4 *usplit;             // [NI][S][128] split rows
  float *gate;               // [NI][S] row 128 of Wcat . img (no bias)
  int S;
  int dbg;                   // tuning experiments (DF3D_IP_DBG): 1 = no image loads, 2 = no MFMAs, 4 = no W loads, 8 = no stores
  // round 4: pixel-major copies of the RAW 256-channel rows of the pixels a query samples (pixrow[img * S + p] = row of the
  // compact buffer or -1; NULL = none): the tile passes through LDS anyway, and the query assembly then reads one contiguous
  // 1 KB row per query instead of 256 scattered 4-byte elements of the channel-first map
  const int32_t *pixrow;
  float *compact;            // [marked pixels][256]
};

__global__ __launch_bounds__(512) void img_proj_split_kernel(ProjArgs2 a) {
  // one raw LDS buffer: packed Wcat tiles (2 x 18 KB) + image tiles (2 x 16.25 KB); the epilogue reuses all of it
  // to turn the accumulators into whole 512-byte rows
  constexpr int W_BYTES = 2 * IP_WQ * 16, X_BYTES = 2 * 32 * IP_LD * 4;
  constexpr int ROW_PITCH = 528;                  // 512 B row + 16 B: rows start 4 banks apart
  static_assert(8 * 16 * ROW_PITCH <= W_BYTES + X_BYTES, "epilogue tile must fit");
  __shared__ __attribute__((aligned(16))) char smem[W_BYTES + X_BYTES];
  u32x4 (*Wl)[IP_WQ] = (u32x4(*)[IP_WQ])smem;
  float (*Xl)[32 * IP_LD] = (float(*)[32 * IP_LD])(smem + W_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;
  const int p0 = blockIdx.x * IP_TP;
  const float *img = a.img[blockIdx.y];
  const int S = a.S;
  const bool pair_ok = (S & 1) == 0;             // rows of an even-length map are 8-byte aligned

  // register staging of the next step's tiles: image tile 32 x 128 floats = 2048 float2, 4 per thread
  float2 xr[4];
  u32x4 wr[3];
  auto load_tiles = [&](int kb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int e = tid + 512 * i;                     // float2 index: k = e / 64, pixel pair = e % 64
      int k = e >> 6, p = p0 + 2 * (e & 63);
      const float *src = img + (size_t)(kb * 32 + k) * S + p;
      float2 v = make_float2(0.f, 0.f);
      if (a.dbg & 1) {
      } else if (pair_ok && p + 1 < S) v = *(const float2 *)src;
      else {
        if (p < S) v.x = src[0];
        if (p + 1 < S) v.y = src[1];
      }
      xr[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      int e = tid + 512 * i;
      wr[i] = a.w[(a.dbg & 4) ? (size_t)(e < IP_WQ ? e : 0) & 63 : (size_t)kb * IP_WQ + (e < IP_WQ ? e : 0)];
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int e = tid + 512 * i;
      int k = e >> 6, pp = 2 * (e & 63);
      *(float2 *)(&Xl[buf][k * IP_LD + pp]) = xr[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      int e = tid + 512 * i;
      if (e < IP_WQ) Wl[buf][e] = wr[i];
    }
  };

  f32x4 acc[IP_MT];
#pragma unroll
  for (int t = 0; t < IP_MT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // the tile's marked pixels as a compact list (pixel in tile | row of the compact buffer): ~13 % of the pixels
  __shared__ int mlist[IP_TP][2];
  __shared__ int mcount;
  if (a.pixrow) {
    if (tid == 0) mcount = 0;
    __syncthreads();
    if (tid < IP_TP && p0 + tid < S) {
      const int r = a.pixrow[(size_t)blockIdx.y * S + p0 + tid];
      if (r >= 0) {
        const int at = atomicAdd(&mcount, 1);
        mlist[at][0] = tid;
        mlist[at][1] = r;
      }
    }
  }

  load_tiles(0);
  store_tiles(0);
  load_tiles(1);
  for (int kb = 0; kb < 8; ++kb) {
    __syncthreads();
    if (kb + 1 < 8) store_tiles((kb + 1) & 1);
    if (kb + 2 < 8) load_tiles(kb + 2);
    if (a.pixrow) {
      // this step's 32 channels of every marked pixel: half a wave per pixel, one 128-byte segment each
      const int nm = mcount;
      for (int m = tid >> 5; m < nm; m += 16) {
        const int px = mlist[m][0], k = tid & 31;
        a.compact[(size_t)mlist[m][1] * IP_CIN + kb * 32 + k] = Xl[kb & 1][k * IP_LD + px];
      }
    }
    // B operand: the 8 k-values of this lane's pixel, read transposed from the staged tile, split hi/lo
    const float *xb = &Xl[kb & 1][(g * 8) * IP_LD + wave * 16 + n];
    u32x4 bh, bl;
#pragma unroll
    for (int e = 0; e < 4; ++e) split_pair_nc(xb[(2 * e) * IP_LD], xb[(2 * e + 1) * IP_LD], bh[e], bl[e]);   // (checked in the epilogue)
    const u32x4 *wb = Wl[kb & 1] + lane;
    if (a.dbg & 2) continue;
    // three row tiles at a time: their 9 MFMAs are interleaved (no back-to-back dependent pair) and the next
    // group's A fragments come from LDS meanwhile
    u32x4 fq[2][6];
#pragma unroll
    for (int q = 0; q < 6; ++q) fq[0][q] = wb[q * 64];
#pragma unroll
    for (int tg = 0; tg < IP_MT / 3; ++tg) {
      const int cur = tg & 1;
      if (tg + 1 < IP_MT / 3) {
#pragma unroll
        for (int q = 0; q < 6; ++q) fq[cur ^ 1][q] = wb[((tg + 1) * 6 + q) * 64];
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[tg * 3 + j] = DF3D_MFMA_F16(fq[cur][2 * j], bl, acc[tg * 3 + j]);
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[tg * 3 + j] = DF3D_MFMA_F16(fq[cur][2 * j + 1], bh, acc[tg * 3 + j]);
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[tg * 3 + j] = DF3D_MFMA_F16(fq[cur][2 * j], bh, acc[tg * 3 + j]);
    }
  }

  // epilogue: lane (pixel n, g) holds channels 16t + 4g + {0..3}.  The wave's 16 split rows (8 KB) are assembled
  // in LDS and stored as whole 512-byte rows (scattered 8-byte global stores cost 50 us of 135)
  __syncthreads();
  char *wt = smem + wave * 16 * ROW_PITCH;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    unsigned h[2], l[2];
    const f32x4 uv = acc[t] * DF3D_ACC_UNSCALE;
    split_pair(uv[0], uv[1], h[0], l[0]);
    split_pair(uv[2], uv[3], h[1], l[1]);
    char *blk = wt + n * ROW_PITCH + (2 * t + (g >> 1)) * 32 + (g & 1) * 8;   // 8-channel block = [hi 16 B | lo 16 B]
    *(u32x2 *)blk = (u32x2){h[0], h[1]};
    *(u32x2 *)(blk + 16) = (u32x2){l[0], l[1]};
  }
  const int pw = p0 + wave * 16;
  if (g == 0 && pw + n < S && !(a.dbg & 8)) a.gate[(size_t)blockIdx.y * S + pw + n] = acc[8][0] * DF3D_ACC_UNSCALE;
  __builtin_amdgcn_wave_barrier();
  if (a.dbg & 8) return;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int r = it * 2 + (lane >> 5), c16 = lane & 31;
    if (pw + r < S) {
      const u32x4 v = *(const u32x4 *)(wt + r * ROW_PITCH + c16 * 16);
      a.usplit[((size_t)blockIdx.y * S + pw + r) * 32 + c16] = v;
    }
  }
}

// Round 4, second structure of the projection: the image tile does NOT pass through LDS.
// Ablations of the kernel above (tools/ubench/imgproj_probe.py, DF3D_IP_DBG, MI355X): 141 us as shipped, 64 us without the
// image loads, 122 without the MFMAs, 99 without the stores, 146 without the Wcat loads -- the 246 MB of camera maps cost
// ~77 us that nothing overlaps: a workgroup keeps ONE 16 KB k-step of the maps in flight (4 float2 per thread, issued one
// barrier-separated step ahead of its LDS store), two workgroups per CU = 32 KB per CU, and a CU ingests 12 B/clk from HBM
// only while ~50-100 KB are outstanding.
// The B operand of a lane IS eight values of its own pixel (k = 8g .. 8g + 7 of the step's 32 channels): here every lane
// loads them straight from the channel-first map -- a wave instruction reads four 64-byte runs (16 pixels x 4 planes), the
// other half of each 128-byte line belongs to the neighbouring wave of the workgroup -- into a ring of DEPTH register sets
// (8 registers per k-step); no LDS write, no transposed LDS read, no barrier between a load and its use.  Only the packed
// Wcat tiles still go through LDS (shared by the eight waves; staged through registers one step ahead, ISSUED BEFORE the
// step's image loads so that the in-order vmcnt wait for them releases the image loads of the previous step only).
// Same operands, same accumulation order: the results are bit-identical to the kernel above.
// COMPACT: also the raw 256-channel rows of the pixels a query samples, pixel-major (a.pixrow / a.compact as in the kernel
// above): the lane HOLDS eight consecutive channels of its pixel per step -- two 16-byte stores under the lane's mark, no LDS.
template <int D, bool COMPACT = false>           // D image k-steps in registers: one in use, D - 1 in flight
__global__ __launch_bounds__(512, 4) void img_proj_direct_kernel(ProjArgs2 a) {
  constexpr int WQP = 3 * 512;                    // a stage padded to three stores per thread: no branch around the third
  constexpr int W_BYTES = 2 * WQP * 16;
  constexpr int ROW_PITCH = 528;                  // 512 B row + 16 B: rows start 4 banks apart
  constexpr int EPI_BYTES = 8 * 16 * ROW_PITCH;
  __shared__ __attribute__((aligned(16))) char smem[W_BYTES > EPI_BYTES ? W_BYTES : EPI_BYTES];
  u32x4 (*Wl)[WQP] = (u32x4(*)[WQP])smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;
  const int p0 = blockIdx.x * IP_TP;
  const int S = a.S;
  const int pix = p0 + wave * 16 + n;
  // this lane's column of the map: channel 8g + e of step kb at xp[(kb * 32 + e) * cs].  Pixels past the end of the map
  // read the last pixel (branch-free loads: the compiler can then count its vmcnt waits; their rows are never stored)
  // (a global-address-space pointer: the table entry is a generic pointer, and flat loads count in lgkmcnt as well --
  // the compiler then drains every queue at each LDS access)
  typedef const __attribute__((address_space(1))) float *gptr;
  const size_t cs = (size_t)S;
  gptr xp = (gptr)(a.img[blockIdx.y]) + (size_t)(g * 8) * cs + (pix < S ? pix : S - 1);
  int prow = -1;
  if constexpr (COMPACT) prow = pix < S ? a.pixrow[(size_t)blockIdx.y * S + pix] : -1;

  float xs[D][8];
  u32x4 wr[3];
  auto load_x = [&](int kb, float (&x)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = xp[(size_t)(kb * 32 + e) * cs];
  };
  auto load_w = [&](int kb) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      int e = tid + 512 * i;
      wr[i] = a.w[(size_t)kb * IP_WQ + (e < IP_WQ ? e : 0)];
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      int e = tid + 512 * i;
      Wl[buf][e] = wr[i];
    }
  };

  f32x4 acc[IP_MT];
#pragma unroll
  for (int t = 0; t < IP_MT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  load_w(0);
#pragma unroll
  for (int j = 0; j < D - 1; ++j) load_x(j, xs[j]);
  store_w(0);
  load_w(1);
#pragma unroll
  for (int kb = 0; kb < 8; ++kb) {
    __syncthreads();
    if (kb + 1 < 8) store_w((kb + 1) & 1);
    if (kb + 2 < 8) load_w(kb + 2);
    if (kb + D - 1 < 8) load_x(kb + D - 1, xs[(kb + D - 1) % D]);
    float (&x)[8] = xs[kb % D];
    if constexpr (COMPACT) {
      if (prow >= 0) {
        f32x4 *dst = (f32x4 *)(a.compact + (size_t)prow * IP_CIN + kb * 32 + g * 8);
        dst[0] = (f32x4){x[0], x[1], x[2], x[3]};
        dst[1] = (f32x4){x[4], x[5], x[6], x[7]};
      }
    }
    u32x4 bh, bl;
#pragma unroll
    for (int e = 0; e < 4; ++e) split_pair_nc(x[2 * e], x[2 * e + 1], bh[e], bl[e]);   // (checked in the epilogue)
    const u32x4 *wb = Wl[kb & 1] + lane;
#pragma unroll
    for (int tg = 0; tg < IP_MT / 3; ++tg) {
      u32x4 fq[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) fq[q] = wb[(tg * 6 + q) * 64];
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[tg * 3 + j] = DF3D_MFMA_F16(fq[2 * j], bl, acc[tg * 3 + j]);
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[tg * 3 + j] = DF3D_MFMA_F16(fq[2 * j + 1], bh, acc[tg * 3 + j]);
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[tg * 3 + j] = DF3D_MFMA_F16(fq[2 * j], bh, acc[tg * 3 + j]);
    }
  }

  // epilogue of the kernel above: the wave's 16 split rows (8 KB) meet in LDS and leave as whole 512-byte rows
  __syncthreads();
  char *wt = smem + wave * 16 * ROW_PITCH;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    unsigned h[2], l[2];
    const f32x4 uv = acc[t] * DF3D_ACC_UNSCALE;
    split_pair(uv[0], uv[1], h[0], l[0]);
    split_pair(uv[2], uv[3], h[1], l[1]);
    char *blk = wt + n * ROW_PITCH + (2 * t + (g >> 1)) * 32 + (g & 1) * 8;   // 8-channel block = [hi 16 B | lo 16 B]
    *(u32x2 *)blk = (u32x2){h[0], h[1]};
    *(u32x2 *)(blk + 16) = (u32x2){l[0], l[1]};
  }
  const int pw = p0 + wave * 16;
  if (g == 0 && pw + n < S) a.gate[(size_t)blockIdx.y * S + pw + n] = acc[8][0] * DF3D_ACC_UNSCALE;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int r = it * 2 + (lane >> 5), c16 = lane & 31;
    if (pw + r < S) {
      const u32x4 v = *(const u32x4 *)(wt + r * ROW_PITCH + c16 * 16);
      a.usplit[((size_t)blockIdx.y * S + pw + r) * 32 + c16] = v;
    }
  }
}

// mom[n][c] = (sum_p a_p u_cp, sum_p (a_p u_cp)^2) from split rows; a may be NULL (a_p = 1)
__global__ __launch_bounds__(256) void split_moments_kernel(const u32x4 *__restrict__ us, const float *__restrict__ a,
                                                            int S, int rows_per_block, double *__restrict__ mom) {
  const int nimg = blockIdx.y;
  const int blk = threadIdx.x & 15, rsub = threadIdx.x >> 4;     // 16 blocks of 8 channels per row, 16 rows in flight
  const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, S);
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
  for (int r = r0 + rsub; r < r1; r += 16) {
    const size_t row = (size_t)nimg * S + r;
    const u32x4 hi = us[row * 32 + blk * 2], lo = us[row * 32 + blk * 2 + 1];
    const float ar = a ? a[row] : 1.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const df3d_f32x2 uv = split_to_f32(hi[e], lo[e]);
      float v0 = uv[0] * ar;
      float v1 = uv[1] * ar;
      s1[2 * e] += v0;
      s2[2 * e] += v0 * v0;
      s1[2 * e + 1] += v1;
      s2[2 * e + 1] += v1 * v1;
    }
  }
  __shared__ float sh[2][16][IP_C + 1];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sh[0][rsub][blk * 8 + e] = s1[e];
    sh[1][rsub][blk * 8 + e] = s2[e];
  }
  __syncthreads();
  if (threadIdx.x < IP_C) {
    double d1 = 0.0, d2 = 0.0;
    for (int k = 0; k < 16; ++k) {
      d1 += sh[0][k][threadIdx.x];
      d2 += sh[1][k][threadIdx.x];
    }
    atomicAdd(&mom[((size_t)nimg * IP_C + threadIdx.x) * 2], d1);
    atomicAdd(&mom[((size_t)nimg * IP_C + threadIdx.x) * 2 + 1], d2);
  }
}

constexpr int RG_COUT = 256;                      // both layers' value projections
constexpr int RG_CT = RG_COUT / 16;
constexpr int RG_WQ = RG_CT * 2 * 64;             // u32x4 per step tile = 32 KB

// GroupNorm fold (see gn_fold_kernel in actr.hip) with the folded weights written as packed B operands
//   Wp[n][kb 4][ct 16][hi|lo][lane]: lane (col_n, g) -> output column (ct/4)*64 + col_n*4 + ct%4, channels 32kb+8g+e
__global__ __launch_bounds__(256) void gn_fold_pack_kernel(const double *__restrict__ mom, const float *__restrict__ b,
                                                           const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, float eps, int S, int groups,
                                                           const float *__restrict__ W, const float *__restrict__ wb,
                                                           u32x4 *__restrict__ Wp, float *__restrict__ cf) {
  const int n = blockIdx.x, tid = threadIdx.x;
  constexpr int C = IP_C, O = RG_COUT;
  __shared__ double m1[C], m2[C];
  __shared__ float sc[C], tc[C];
  if (tid < C) {
    double su = mom[((size_t)n * C + tid) * 2], sq = mom[((size_t)n * C + tid) * 2 + 1];
    double bc = b ? (double)b[tid] : 0.0;
    m1[tid] = su + S * bc;
    m2[tid] = sq + 2.0 * bc * su + S * bc * bc;
  }
  __syncthreads();
  const int cpg = C / groups;
  if (tid < C) {
    int gi = tid / cpg;
    double s = 0.0, q = 0.0;
    for (int k = 0; k < cpg; ++k) { s += m1[gi * cpg + k]; q += m2[gi * cpg + k]; }
    double cnt = (double)cpg * S;
    double mean = s / cnt, var = q / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    float rstd = (float)(1.0 / sqrt(var + (double)eps));
    float bc = b ? b[tid] : 0.f;
    sc[tid] = rstd * gamma[tid];
    tc[tid] = (bc - (float)mean) * rstd * gamma[tid] + beta[tid];
  }
  __syncthreads();
  // (round 4) gridDim.y workgroups share an image: one workgroup per image was six workgroups walking 32 dependent
  // iterations each -- 17 us of latency between the moments and the GEMM
  const int part = blockIdx.y, nparts = gridDim.y;
  // the folded constant: 16 lanes per output walk the row in channel order 16-apart, partial sums meet by shuffles
  for (int o = part * 16 + (tid >> 4); o < O; o += 16 * nparts) {
    const int l = tid & 15;
    float acc = 0.f;
    for (int c = l; c < C; c += 16) acc += W[(size_t)o * C + c] * tc[c];
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 16);
    if (l == 0) cf[(size_t)n * O + o] = acc + (wb ? wb[o] : 0.f);
  }
  for (int i = part * 256 + tid; i < 4 * RG_WQ; i += 256 * nparts) {
    int lane = i & 63, part = (i >> 6) & 1, ct = (i >> 7) & 15, kb = i >> 11;
    int col = (ct >> 2) * 64 + (lane & 15) * 4 + (ct & 3), gq = lane >> 4;   // store q = ct/4 writes 256 B runs
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = kb * 32 + gq * 8 + 2 * e;
      unsigned hi, lo;
      split_pair_w(W[(size_t)col * C + c] * sc[c], W[(size_t)col * C + c + 1] * sc[c + 1], hi, lo);
      o[e] = part ? lo : hi;
    }
    Wp[(size_t)n * 4 * RG_WQ + i] = o;
  }
}

// value[n][p][0:256] = u[n][p][0:128] . Wf[n]^T: rows contiguous, weights per image (blockIdx.y)
template <bool BF16>
__global__ __launch_bounds__(512) void rows_gemm_split_kernel(const u32x4 *__restrict__ us, const u32x4 *__restrict__ Wp,
                                                              int S, float *__restrict__ out) {
  __shared__ u32x4 Wl[2][RG_WQ];                 // 64 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;
  const int nimg = blockIdx.y;
  const int p0 = blockIdx.x * 128 + wave * 16;
  const int pr = min(p0 + n, S - 1);
  const u32x4 *w = Wp + (size_t)nimg * 4 * RG_WQ;
  // A operands of the wave's 16 rows, all four 32-channel blocks
  u32x4 ah[4], al[4];
  const u32x4 *urow = us + ((size_t)nimg * S + pr) * 32;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    ah[kb] = urow[(kb * 4 + g) * 2];
    al[kb] = urow[(kb * 4 + g) * 2 + 1];
  }
  u32x4 wr[4];
  auto load_w = [&](int kb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) wr[i] = w[(size_t)kb * RG_WQ + tid + 512 * i];
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) Wl[buf][tid + 512 * i] = wr[i];
  };
  f32x4 acc[RG_CT];
#pragma unroll
  for (int ct = 0; ct < RG_CT; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  load_w(0);
  store_w(0);
  load_w(1);
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    __syncthreads();
    if (kb + 1 < 4) store_w((kb + 1) & 1);
    if (kb + 2 < 4) load_w(kb + 2);
    const u32x4 *wb = Wl[kb & 1] + lane;
    u32x4 fq[2][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) fq[0][k] = wb[k * 64];
#pragma unroll
    for (int ct = 0; ct < RG_CT; ct += 2) {
      const int cur = (ct >> 1) & 1;
      if (ct + 2 < RG_CT) {
#pragma unroll
        for (int k = 0; k < 4; ++k) fq[cur ^ 1][k] = wb[((ct + 2) * 2 + k) * 64];
      }
      const u32x4 bh0 = fq[cur][0], bl0 = fq[cur][1], bh1 = fq[cur][2], bl1 = fq[cur][3];
      acc[ct] = DF3D_MFMA_F16(al[kb], bh0, acc[ct]);
      acc[ct + 1] = DF3D_MFMA_F16(al[kb], bh1, acc[ct + 1]);
      acc[ct] = DF3D_MFMA_F16(ah[kb], bl0, acc[ct]);
      acc[ct + 1] = DF3D_MFMA_F16(ah[kb], bl1, acc[ct + 1]);
      acc[ct] = DF3D_MFMA_F16(ah[kb], bh0, acc[ct]);
      acc[ct + 1] = DF3D_MFMA_F16(ah[kb], bh1, acc[ct + 1]);
    }
  }
  // lane (col_n, g) holds rows 4g+r, columns 64q + 4 col_n + {0..3} of column tile 4q+j: one store instruction
  // covers 256 contiguous bytes of a row
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int p = blockIdx.x * 128 + wave * 16 + 4 * g + r;
    if (p >= S) continue;
    if constexpr (BF16) {                       // bf16 rows (round to nearest even): half the bytes for the sampler's gathers
      unsigned short *o = (unsigned short *)out + ((size_t)nimg * S + p) * RG_COUT + n * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(u32x2 *)(o + q * 64) = (u32x2){bf16_pair(acc[q * 4][r] * DF3D_ACC_UNSCALE, acc[q * 4 + 1][r] * DF3D_ACC_UNSCALE),
                                         bf16_pair(acc[q * 4 + 2][r] * DF3D_ACC_UNSCALE, acc[q * 4 + 3][r] * DF3D_ACC_UNSCALE)};
    } else {
      float *o = out + ((size_t)nimg * S + p) * RG_COUT + n * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(f32x4 *)(o + q * 64) =
            (f32x4){acc[q * 4][r], acc[q * 4 + 1][r], acc[q * 4 + 2][r], acc[q * 4 + 3][r]} * DF3D_ACC_UNSCALE;
    }
  }
}

}  // namespace df3d

using namespace df3d;

extern "C" size_t df3d_imgproj_packed_bytes(int rows, int cin) {
  if (cin != IP_CIN || rows <= 0 || rows > IP_MT * 16) return 0;
  return (size_t)8 * IP_WQ * 16;
}

extern "C" int df3d_imgproj_pack(const float *wcat, int rows, int cin, void *packed, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(wcat && packed, "imgproj_pack: null argument");
  DF3D_CHECK_ARG(df3d_imgproj_packed_bytes(rows, cin) != 0,
                 "imgproj_pack: serves cin == 256 and at most 144 output rows (got %d x %d)", rows, cin);
  hipLaunchKernelGGL(pack_proj_kernel, dim3(cdiv(8 * IP_WQ, 256)), dim3(256), 0, stream, wcat, rows, (u32x4 *)packed);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_imgproj_split(const float *const *img_ptrs, int nimg, int cin, int S, const void *packed,
                                  void *u_split, float *gate, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(img_ptrs && packed && u_split && gate, "imgproj_split: null argument");
  DF3D_CHECK_ARG(cin == IP_CIN, "imgproj_split: serves 256 input channels (got %d)", cin);
  if (nimg == 0 || S == 0) return DF3D_OK;
  ProjArgs2 a = {img_ptrs, (const u32x4 *)packed, (u32x4 *)u_split, gate, S,
                 getenv("DF3D_IP_DBG") ? atoi(getenv("DF3D_IP_DBG")) : 0, nullptr, nullptr};
  // DF3D_IMGPROJ_DIRECT=0: the LDS-staged kernel of round 3 (A/B; read per call)
  const char *dm = getenv("DF3D_IMGPROJ_DIRECT");
  if (dm && dm[0] == '0') hipLaunchKernelGGL(img_proj_split_kernel, dim3(cdiv(S, IP_TP), nimg), dim3(512), 0, stream, a);
  else {
    static const int depth = getenv("DF3D_IP_DEPTH") ? atoi(getenv("DF3D_IP_DEPTH")) : 3;      // tuning aid
    if (depth == 2) hipLaunchKernelGGL(img_proj_direct_kernel<2>, dim3(cdiv(S, IP_TP), nimg), dim3(512), 0, stream, a);
    else if (depth == 4) hipLaunchKernelGGL(img_proj_direct_kernel<4>, dim3(cdiv(S, IP_TP), nimg), dim3(512), 0, stream, a);
    else if (depth == 5) hipLaunchKernelGGL(img_proj_direct_kernel<5>, dim3(cdiv(S, IP_TP), nimg), dim3(512), 0, stream, a);
    else hipLaunchKernelGGL(img_proj_direct_kernel<3>, dim3(cdiv(S, IP_TP), nimg), dim3(512), 0, stream, a);
  }
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_imgproj_split_compact(const float *const *img_ptrs, int nimg, int cin, int S, const void *packed,
                                          void *u_split, float *gate, const int32_t *pixrow, float *compact, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(img_ptrs && packed && u_split && gate && pixrow && compact, "imgproj_split_compact: null argument");
  DF3D_CHECK_ARG(cin == IP_CIN, "imgproj_split_compact: serves 256 input channels (got %d)", cin);
  if (nimg == 0 || S == 0) return DF3D_OK;
  ProjArgs2 a = {img_ptrs, (const u32x4 *)packed, (u32x4 *)u_split, gate, S, 0, pixrow, compact};
  const char *dm = getenv("DF3D_IMGPROJ_DIRECT");
  if (dm && dm[0] == '0') hipLaunchKernelGGL(img_proj_split_kernel, dim3(cdiv(S, IP_TP), nimg), dim3(512), 0, stream, a);
  else hipLaunchKernelGGL((img_proj_direct_kernel<3, true>), dim3(cdiv(S, IP_TP), nimg), dim3(512), 0, stream, a);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

static int value_fold_gemm(const void *u_split, const float *att, int nimg, int S, const float *conv_bias,
                          const float *gn_weight, const float *gn_bias, float eps, int groups, const float *W,
                          const float *wb, double *moments, void *packed_w, float *cf, void *value, bool bf16,
                          void *stream_);

extern "C" int df3d_value_fold_gemm(const void *u_split, const float *att, int nimg, int S, const float *conv_bias,
                                    const float *gn_weight, const float *gn_bias, float eps, int groups,
                                    const float *W, const float *wb, double *moments, void *packed_w, float *cf,
                                    float *value, void *stream_) {
  return value_fold_gemm(u_split, att, nimg, S, conv_bias, gn_weight, gn_bias, eps, groups, W, wb, moments, packed_w, cf,
                         value, false, stream_);
}

extern "C" int df3d_value_fold_gemm_bf16(const void *u_split, const float *att, int nimg, int S, const float *conv_bias,
                                         const float *gn_weight, const float *gn_bias, float eps, int groups,
                                         const float *W, const float *wb, double *moments, void *packed_w, float *cf,
                                         void *value_bf16, void *stream_) {
  return value_fold_gemm(u_split, att, nimg, S, conv_bias, gn_weight, gn_bias, eps, groups, W, wb, moments, packed_w, cf,
                         value_bf16, true, stream_);
}

static int value_fold_gemm(const void *u_split, const float *att, int nimg, int S, const float *conv_bias,
                          const float *gn_weight, const float *gn_bias, float eps, int groups, const float *W,
                          const float *wb, double *moments, void *packed_w, float *cf, void *value, bool bf16,
                          void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(u_split && gn_weight && gn_bias && W && moments && packed_w && cf && value,
                 "value_fold_gemm: null argument");
  DF3D_CHECK_ARG(groups > 0 && IP_C % groups == 0, "value_fold_gemm: bad group count %d", groups);
  if (nimg == 0 || S == 0) return DF3D_OK;
  DF3D_HIP(hipMemsetAsync(moments, 0, (size_t)nimg * IP_C * 2 * sizeof(double), stream));
  const int rpb = 160;          // ~1500 blocks at nuScenes size: enough waves to stream at HBM rate
  hipLaunchKernelGGL(split_moments_kernel, dim3(cdiv(S, rpb), nimg), dim3(256), 0, stream, (const u32x4 *)u_split, att,
                     S, rpb, moments);
  hipLaunchKernelGGL(gn_fold_pack_kernel, dim3(nimg, 16), dim3(256), 0, stream, moments, conv_bias, gn_weight, gn_bias, eps,
                     S, groups, W, wb, (u32x4 *)packed_w, cf);
  if (bf16)
    hipLaunchKernelGGL(rows_gemm_split_kernel<true>, dim3(cdiv(S, 128), nimg), dim3(512), 0, stream, (const u32x4 *)u_split,
                       (const u32x4 *)packed_w, S, (float *)value);
  else
    hipLaunchKernelGGL(rows_gemm_split_kernel<false>, dim3(cdiv(S, 128), nimg), dim3(512), 0, stream, (const u32x4 *)u_split,
                       (const u32x4 *)packed_w, S, (float *)value);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}
