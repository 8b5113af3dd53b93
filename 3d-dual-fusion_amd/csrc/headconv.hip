// Final convolutions of the detection head's (task, head) branches for gfx950.
//
// Reference: CP/det3d/models/bbox_heads/center_head.py:66-110 (SepHead): every branch ends with a 3x3 Conv2d 64 -> k
// with k = 1..3 output maps (heat-map classes, reg 2, height 1, dim 3, rot 2, vel 2) -- 36 branches, 72 maps in the
// nuScenes head.  As a block-diagonal launch of the matrix-core kernel they cost 16x the useful multiplies (k padded to
// 32 columns) and re-read the 298 MB of branch activations nine times through L2: 358 us.
// The work is 2.7 GFLOP over 298 MB: a vector-ALU / LDS problem.  A workgroup owns one (8x8 pixel tile, branch): it stages the
// branch's 10x10x64 halo of fp32 values (hi + lo of the split rows the previous conv emitted; zeros outside the map) in
// LDS as [channel][pixel]; the four waves of the workgroup take a quarter of the channels each and every lane (= pixel)
// walks 9 taps x 16 channels with ONE LDS read per (tap, channel) and k FMAs whose weight operand is wave-uniform (scalar
// loads); the four partial sums meet in LDS, + bias, stores into the packed [pixels, 72] buffer that
// `CenterHead.predict` / `loss_device` read in place.  Each activation is read from memory 1.56 times (halo) instead of 9;
// 30 KB of LDS per workgroup keeps five workgroups (20 waves) per CU in flight.
#include "common.h"
#include <algorithm>

namespace df3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int HF_TILE = 8, HF_HALO = HF_TILE + 2, HF_HP = HF_HALO * HF_HALO, HF_C = 64, HF_KMAX = 4;
constexpr int HF_LD = HF_HP + 1;                       // pixels per channel row in LDS (+1: rows of 10 shift the banks)

struct HeadFinalArgs {
  const u32x4 *in;       // split rows [B*H*W][ld] (u32x4 units), branch g at columns g*16 .. g*16+15 (64 channels)
  const float *w;        // [G][9][64][HF_KMAX] fp32, output maps padded to 4
  const u32x4 *wp;       // the same filters as B operands of head_final_mfma_kernel (head_final_pack_kernel) or null
  const float *bias;     // [G][HF_KMAX]
  const int32_t *cols;   // [G][2] (first output column, valid maps)
  float *out;            // [B*H*W][ldo]
  int ld, ldo, B, H, W, G, tiles_x, tiles_y, gpad;
};

DF3D_SPLIT_OVERFLOW_TU(headconv)

__global__ __launch_bounds__(256) void head_final_kernel(HeadFinalArgs a) {
  __shared__ float x[HF_C * HF_LD];                  // the branch's halo tile, [channel][pixel]
  __shared__ float part[4][64][HF_KMAX];             // partial sums of the four channel quarters
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = blockIdx.y;
  const int tile = blockIdx.x;
  const int b = tile / (a.tiles_x * a.tiles_y), t2 = tile - b * (a.tiles_x * a.tiles_y);
  const int ty = t2 / a.tiles_x, tx = t2 - ty * a.tiles_x;
  const int y0 = ty * HF_TILE - 1, x0 = tx * HF_TILE - 1;
  // ---- halo -> LDS: item = (halo pixel, 8-channel block): 32 B of split row = hi 8 x fp16 | lo 8 x fp16 (round 5; bf16 pairs before).  All loads of a
  //      thread are issued before the first conversion (independent addresses: one memory round trip, not four) ----
  constexpr int ITEMS = HF_HP * 8, PER = (ITEMS + 255) / 256;
  u32x4 hi[PER], lo[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int it = tid + 256 * i;
    const int p = it >> 3, blk = it & 7;
    const int hy = p / HF_HALO, hx = p - hy * HF_HALO;
    const int yy = y0 + hy, xx = x0 + hx;
    hi[i] = lo[i] = (u32x4){0u, 0u, 0u, 0u};
    if (it < ITEMS && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {
      const u32x4 *src = a.in + ((size_t)(b * a.H + yy) * a.W + xx) * a.ld + g * 16 + blk * 2;
      hi[i] = src[0];
      lo[i] = src[1];
    }
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int it = tid + 256 * i;
    if (it >= ITEMS) break;
    const int p = it >> 3, blk = it & 7;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const df3d_f32x2 xv = split_to_f32(hi[i][e], lo[i][e]);
      x[(blk * 8 + 2 * e) * HF_LD + p] = xv[0];
      x[(blk * 8 + 2 * e + 1) * HF_LD + p] = xv[1];
    }
  }
  __syncthreads();
  // ---- wave w: channels 16w .. 16w+15 of all 9 taps for the tile's 64 pixels (lane = pixel); the weights are uniform
  //      over the wave (scalar loads) ----
  const int py = lane >> 3, px = lane & 7;
  const float *wg = a.w + ((size_t)g * 9 * HF_C + wave * 16) * HF_KMAX;
  float acc[HF_KMAX] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int dy = tap / 3, dx = tap - dy * 3;
    const float *xp = x + (wave * 16) * HF_LD + (py + dy) * HF_HALO + (px + dx);
    const float *wt = wg + tap * HF_C * HF_KMAX;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const float xv = xp[c * HF_LD];
      const f32x4 w4 = *(const f32x4 *)(wt + c * HF_KMAX);
      acc[0] = fmaf(xv, w4[0], acc[0]);
      acc[1] = fmaf(xv, w4[1], acc[1]);
      acc[2] = fmaf(xv, w4[2], acc[2]);
      acc[3] = fmaf(xv, w4[3], acc[3]);
    }
  }
  *(f32x4 *)&part[wave][lane][0] = (f32x4){acc[0], acc[1], acc[2], acc[3]};
  __syncthreads();
  // ---- thread (map k = tid / 64, pixel = lane): sum of the four quarters + bias ----
  const int k = tid >> 6;
  const int oy = ty * HF_TILE + py, ox = tx * HF_TILE + px;
  if (k < a.cols[2 * g + 1] && oy < a.H && ox < a.W) {
    const float v = ((part[0][lane][k] + part[1][lane][k]) + (part[2][lane][k] + part[3][lane][k])) + a.bias[g * HF_KMAX + k];
    a.out[((size_t)(b * a.H + oy) * a.W + ox) * a.ldo + a.cols[2 * g] + k] = v;
  }
}


// ---- round 3: the same convolutions on the matrix cores, "multiply, then shift" ---------------------------------------------
// head_final_kernel is bound by its vector ALU, not by memory (tools/ubench/head_final.py: 150 us with the loads, 122 us
// without them): unpacking the 16-bit halves costs as many instructions as the 576 FMAs per pixel, all of them at one wave64
// instruction per 4 clocks.  The split rows ARE matrix-core operands, though: 16 bytes of a row = 8 fp16 channels = what one
// lane of v_mfma_f32_16x16x32_f16 holds of its A operand.  Contracting over the 576 = 9 taps x 64 channels directly would
// make every activation an operand nine times (and the k <= 4 output maps fill 4 of the 16 columns).  Instead the taps go
// into the COLUMNS: for every pixel q of a 16 x 16 halo tile
//     D[q][tap * 4 + j] = sum_c M[q][c] * W[tap][c][j]          (256 x 64 times 64 x 36: each activation is an operand ONCE,
//                                                                straight from global memory, no LDS, no conversion)
// and the convolution is the shifted sum  O[p][j] = sum_tap D[p + off(tap)][tap * 4 + j]  over the 14 x 14 interior, through
// LDS.  Three products per k-step as everywhere (M_hi W_hi + M_lo W_hi + M_hi W_lo; the filters are split in the kernel, 48
// values per lane): 288 MFMAs and 37 KB of LDS per (tile, branch); the launch is bound by reading the activations
// (1.31x for the halo).
constexpr int HM_HALO = 16, HM_TILE = HM_HALO - 2, HM_PS = 9 * HF_KMAX;      // LDS floats per halo pixel
#define HM_MFMA(A, B, C) DF3D_MFMA_F16(A, B, C)

// B operand of lane (column `col` = (tap, map j), channel group kg) for k-step s: channels 32 s + 8 kg .. + 7, hi and lo
__device__ __forceinline__ void hm_split_filters(const float *w, int g, int col, int tap, int j, int s, int kg, u32x4 &bh,
                                                 u32x4 &bl) {
  bh = bl = (u32x4){0u, 0u, 0u, 0u};
  if (col >= HM_PS) return;
  const float *wp = w + ((size_t)(g * 9 + tap) * HF_C + 32 * s + 8 * kg) * HF_KMAX + j;
#pragma unroll
  for (int e = 0; e < 4; ++e) split_pair_w(wp[(2 * e) * HF_KMAX], wp[(2 * e + 1) * HF_KMAX], bh[e], bl[e]);
}

// [G][9][64][4] fp32 -> [G][column tile 3][k-step 2][hi | lo][lane 64] x 16 B: what the lanes of the kernel below load
__global__ __launch_bounds__(64) void head_final_pack_kernel(const float *__restrict__ w, u32x4 *__restrict__ out) {
  const int lane = threadIdx.x, n = lane & 15, kg = lane >> 4;
  const int s = blockIdx.x & 1, ct = (blockIdx.x >> 1) % 3, g = blockIdx.x / 6;
  const int col = 16 * ct + n;
  u32x4 bh, bl;
  hm_split_filters(w, g, col, col >> 2, col & 3, s, kg, bh, bl);
  u32x4 *q = out + (size_t)blockIdx.x * 128 + lane;
  q[0] = bh;
  q[64] = bl;
}

__global__ __launch_bounds__(256) void head_final_mfma_kernel(HeadFinalArgs a) {
  __shared__ float d[HM_HALO * HM_HALO * HM_PS];
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, kg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // branch-fastest workgroup order: the workgroups in flight read neighbouring 256-byte pieces of the same pixels' rows.
  // With the tile fastest they would all read ONE branch -- addresses 9216 B apart (36 x 256 B) that fall on a quarter of the
  // memory channels.
  int g, tile;
  if (a.gpad) {
    g = blockIdx.x % a.gpad, tile = blockIdx.x / a.gpad;
    if (g >= a.G) return;
  } else {
    g = blockIdx.y, tile = blockIdx.x;
  }
  const int b = tile / (a.tiles_x * a.tiles_y), t2 = tile - b * (a.tiles_x * a.tiles_y);
  const int ty = t2 / a.tiles_x, tx = t2 - ty * a.tiles_x;
  const int y0 = ty * HM_TILE - 1, x0 = tx * HM_TILE - 1;
  // ---- A operands: MFMA row tile rt of this wave = halo row 4 * wave + rt, row n of the tile = halo column n; k-step s,
  //      lane group kg = channels 32 s + 8 kg .. + 7 = block 4 s + kg of the branch's 8 (hi 16 B | lo 16 B).  Zeros outside ----
  u32x4 ah[4][2], al[4][2];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    const int yy = y0 + 4 * wave + rt, xx = x0 + n;
    const bool in = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
    const u32x4 *src = a.in + ((size_t)(b * a.H + (in ? yy : 0)) * a.W + (in ? xx : 0)) * a.ld + g * 16 + kg * 2;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      ah[rt][s] = al[rt][s] = (u32x4){0u, 0u, 0u, 0u};
      if (in) {
        ah[rt][s] = src[s * 8];
        al[rt][s] = src[s * 8 + 1];
      }
    }
  }
  f32x4 acc[4][3];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ct = 0; ct < 3; ++ct) {
    // ---- B operands: column 16 ct + n = (tap, map j); lane group kg = the same 8 channels as in A ----
    const int col = 16 * ct + n, tap = col >> 2, j = col & 3;
    u32x4 bh[2], bl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (a.wp) {                                        // packed once per set of filters: two coalesced 16-byte loads
        const u32x4 *q = a.wp + ((size_t)(g * 3 + ct) * 2 + s) * 128 + lane;
        bh[s] = q[0];
        bl[s] = q[64];
        continue;
      }
      hm_split_filters(a.w, g, col, tap, j, s, kg, bh[s], bl[s]);
    }
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        acc[rt][ct] = HM_MFMA(al[rt][s], bh[s], acc[rt][ct]);
        acc[rt][ct] = HM_MFMA(ah[rt][s], bl[s], acc[rt][ct]);
        acc[rt][ct] = HM_MFMA(ah[rt][s], bh[s], acc[rt][ct]);
      }
  }
  // ---- D -> LDS [halo pixel][tap][map]: the lane holds halo columns 4 kg + r of halo row 4 wave + rt, column 16 ct + n ----
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) {
      const int col = 16 * ct + n;
      if (col < HM_PS) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          d[((4 * wave + rt) * HM_HALO + 4 * kg + r) * HM_PS + col] = acc[rt][ct][r] * DF3D_ACC_UNSCALE;
      }
    }
  __syncthreads();
  // ---- thread = interior pixel: the nine shifted partial sums, + bias ----
  if (tid >= HM_TILE * HM_TILE) return;
  const int oy = tid / HM_TILE, ox = tid - oy * HM_TILE;
  const int gy = ty * HM_TILE + oy, gx = tx * HM_TILE + ox;
  if (gy >= a.H || gx >= a.W) return;
  f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int dy = tap / 3, dx = tap - dy * 3;
    sum += *(const f32x4 *)&d[((oy + dy) * HM_HALO + ox + dx) * HM_PS + tap * HF_KMAX];
  }
  const int c0 = a.cols[2 * g], k = a.cols[2 * g + 1];
  float *dst = a.out + ((size_t)(b * a.H + gy) * a.W + gx) * a.ldo + c0;
#pragma unroll
  for (int j = 0; j < HF_KMAX; ++j)
    if (j < k) dst[j] = sum[j] + a.bias[g * HF_KMAX + j];
}


// ---- backward of the final convolutions (training rows) ------------------------------------------
// Forward (head_final_kernel): O[p][col_g + j] = bias + sum_tap sum_c M[p + off(tap)][g*64 + c] * W[g][tap][c][j].
//   data gradient:    dM[q][g*64 + c]   = sum_tap sum_j dO[q - off(tap)][col_g + j] * W[g][tap][c][j]
//   filter gradient:  dW[g][tap][c][j]  = sum_p M[p + off(tap)][g*64 + c] * dO[p][col_g + j]
// Both on the same (8 x 8 pixel tile, branch) decomposition as the forward kernel, fp32 activations (rows [P][ldm]) and fp32
// output gradients (rows [P][ldo], branch g at columns cols[2g] .. + cols[2g+1]).
struct HeadFinalBwdArgs {
  const float *m;        // activations [B*H*W][ldm] fp32 (filter gradient only)
  const float *go;       // output gradient [B*H*W][ldo]
  const float *w;        // [G][9][64][4]
  const int32_t *cols;   // [G][2]
  float *gm;             // data gradient [B*H*W][ldm]
  float *gw;             // filter gradient [G][9][64][4] (atomics; zero-filled by the host entry)
  int ldm, ldo, B, H, W, G, tiles_x, tiles_y, tiles_per_wg;
};

__global__ __launch_bounds__(256) void head_final_bwd_data_kernel(HeadFinalBwdArgs a) {
  __shared__ float go[HF_HP * HF_KMAX];               // the branch's output-gradient halo tile, [pixel][map]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = blockIdx.y, tile = blockIdx.x;
  const int b = tile / (a.tiles_x * a.tiles_y), t2 = tile - b * (a.tiles_x * a.tiles_y);
  const int ty = t2 / a.tiles_x, tx = t2 - ty * a.tiles_x;
  const int y0 = ty * HF_TILE - 1, x0 = tx * HF_TILE - 1;
  const int c0 = a.cols[2 * g], k = a.cols[2 * g + 1];
  for (int it = tid; it < HF_HP * HF_KMAX; it += 256) {
    const int p = it >> 2, j = it & 3;
    const int hy = p / HF_HALO, hx = p - hy * HF_HALO;
    const int yy = y0 + hy, xx = x0 + hx;
    float v = 0.f;
    if (j < k && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) v = a.go[((size_t)(b * a.H + yy) * a.W + xx) * a.ldo + c0 + j];
    go[it] = v;
  }
  __syncthreads();
  // lane = pixel q of the tile, wave = 16 channels; dM[q][c] = sum_tap <dO[q - off(tap)], W[tap][c]> -- q - off(tap) is halo
  // pixel (py + 2 - dy, px + 2 - dx) for tap (dy, dx)
  const int py = lane >> 3, px = lane & 7;
  f32x4 d[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int dy = tap / 3, dx = tap - dy * 3;
    d[tap] = *(const f32x4 *)&go[((py + 2 - dy) * HF_HALO + (px + 2 - dx)) * HF_KMAX];
  }
  const float *wg = a.w + ((size_t)g * 9 * HF_C + wave * 16) * HF_KMAX;
  float out[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    float acc = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const f32x4 w4 = *(const f32x4 *)(wg + (tap * HF_C + c) * HF_KMAX);      // wave-uniform: scalar loads
      acc = fmaf(d[tap][0], w4[0], acc);
      acc = fmaf(d[tap][1], w4[1], acc);
      acc = fmaf(d[tap][2], w4[2], acc);
      acc = fmaf(d[tap][3], w4[3], acc);
    }
    out[c] = acc;
  }
  const int oy = ty * HF_TILE + py, ox = tx * HF_TILE + px;
  if (oy < a.H && ox < a.W) {
    float *dst = a.gm + ((size_t)(b * a.H + oy) * a.W + ox) * a.ldm + g * HF_C + wave * 16;
#pragma unroll
    for (int c = 0; c < 16; c += 4) *(f32x4 *)(dst + c) = (f32x4){out[c], out[c + 1], out[c + 2], out[c + 3]};
  }
}

__global__ __launch_bounds__(256) void head_final_bwd_filter_kernel(HeadFinalBwdArgs a) {
  __shared__ float x[HF_C * HF_LD];                  // activation halo tile, [channel][pixel]
  __shared__ float go[64 * HF_KMAX];                 // output gradient of the tile's 64 pixels, [pixel][map]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = blockIdx.y;
  const int c0 = a.cols[2 * g], k = a.cols[2 * g + 1];
  // lane = channel, wave w = taps {w, w + 4, w + 8}: acc[i][j] = dW[tap_i][lane][j]
  float acc[3][HF_KMAX];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < HF_KMAX; ++j) acc[i][j] = 0.f;
  const int ntiles = a.B * a.tiles_x * a.tiles_y;
  const int t_begin = blockIdx.x * a.tiles_per_wg, t_end = min(t_begin + a.tiles_per_wg, ntiles);
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int b = tile / (a.tiles_x * a.tiles_y), t2 = tile - b * (a.tiles_x * a.tiles_y);
    const int ty = t2 / a.tiles_x, tx = t2 - ty * a.tiles_x;
    const int y0 = ty * HF_TILE - 1, x0 = tx * HF_TILE - 1;
    __syncthreads();                                 // the previous tile's reads are done
    // halo -> LDS: item = (halo pixel, 4-channel piece): 16-byte loads of the fp32 rows
    for (int it = tid; it < HF_HP * 16; it += 256) {
      const int p = it >> 4, q = it & 15;
      const int hy = p / HF_HALO, hx = p - hy * HF_HALO;
      const int yy = y0 + hy, xx = x0 + hx;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W)
        v = *(const f32x4 *)(a.m + ((size_t)(b * a.H + yy) * a.W + xx) * a.ldm + g * HF_C + q * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) x[(q * 4 + e) * HF_LD + p] = v[e];
    }
    {
      const int p = tid >> 2, j = tid & 3;           // 64 pixels x 4 maps
      const int oy = ty * HF_TILE + (p >> 3), ox = tx * HF_TILE + (p & 7);
      float v = 0.f;
      if (j < k && oy < a.H && ox < a.W) v = a.go[((size_t)(b * a.H + oy) * a.W + ox) * a.ldo + c0 + j];
      go[tid] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int tap = wave + 4 * i;
      if (tap < 9) {
        const int dy = tap / 3, dx = tap - dy * 3;
        const float *xp = x + lane * HF_LD + dy * HF_HALO + dx;
        for (int p = 0; p < 64; ++p) {
          const float xv = xp[(p >> 3) * HF_HALO + (p & 7)];
          const f32x4 g4 = *(const f32x4 *)&go[p * HF_KMAX];              // broadcast read
          acc[i][0] = fmaf(xv, g4[0], acc[i][0]);
          acc[i][1] = fmaf(xv, g4[1], acc[i][1]);
          acc[i][2] = fmaf(xv, g4[2], acc[i][2]);
          acc[i][3] = fmaf(xv, g4[3], acc[i][3]);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int tap = wave + 4 * i;
    if (tap < 9) {
      float *dst = a.gw + (((size_t)g * 9 + tap) * HF_C + lane) * HF_KMAX;
#pragma unroll
      for (int j = 0; j < HF_KMAX; ++j)
        if (j < k && acc[i][j] != 0.f) unsafeAtomicAdd(dst + j, acc[i][j]);
    }
  }
}

}  // namespace df3d

using namespace df3d;

extern "C" size_t df3d_head_final_packed_bytes(int groups) { return groups > 0 ? (size_t)groups * 6 * 128 * 16 : 0; }

extern "C" int df3d_head_final_pack(const float *weights, int groups, void *packed, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(weights && packed && groups >= 1, "head_final_pack: bad argument");
  hipLaunchKernelGGL(head_final_pack_kernel, dim3(groups * 6), dim3(64), 0, stream, weights, (u32x4 *)packed);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

static int head_final_impl(const void *in_split, int in_channels, int batch, int H, int W, int groups, const float *weights,
                           const void *packed, const float *bias, const int32_t *out_cols, float *out, int out_channels,
                           void *stream_);

extern "C" int df3d_head_final_conv(const void *in_split, int in_channels, int batch, int H, int W, int groups,
                                    const float *weights, const float *bias, const int32_t *out_cols, float *out,
                                    int out_channels, void *stream_) {
  return head_final_impl(in_split, in_channels, batch, H, W, groups, weights, nullptr, bias, out_cols, out, out_channels,
                         stream_);
}

extern "C" int df3d_head_final_conv_packed(const void *in_split, int in_channels, int batch, int H, int W, int groups,
                                           const void *packed, const float *bias, const int32_t *out_cols, float *out,
                                           int out_channels, void *stream_) {
  DF3D_CHECK_ARG(packed, "head_final_conv_packed: null argument");
  return head_final_impl(in_split, in_channels, batch, H, W, groups, nullptr, packed, bias, out_cols, out, out_channels,
                         stream_);
}

static int head_final_impl(const void *in_split, int in_channels, int batch, int H, int W, int groups, const float *weights,
                           const void *packed, const float *bias, const int32_t *out_cols, float *out, int out_channels,
                           void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(in_split && (weights || packed) && bias && out_cols && out, "head_final_conv: null argument");
  DF3D_CHECK_ARG(groups >= 1 && in_channels >= groups * HF_C && in_channels % 8 == 0,
                 "head_final_conv: %d branches of 64 channels do not fit %d-channel rows", groups, in_channels);
  DF3D_CHECK_ARG(batch >= 1 && H >= 1 && W >= 1 && out_channels >= 1, "head_final_conv: bad sizes");
  HeadFinalArgs a;
  a.in = (const u32x4 *)in_split;
  a.w = weights;
  a.wp = (const u32x4 *)packed;
  a.bias = bias;
  a.cols = out_cols;
  a.out = out;
  a.ld = in_channels / 4;                              // u32x4 per split row: 16 bytes carry 4 channels' worth (hi+lo of 8 per 32 B)
  a.ldo = out_channels;
  a.B = batch, a.H = H, a.W = W, a.G = groups;
  static const bool valu = getenv("DF3D_HEADFINAL") && getenv("DF3D_HEADFINAL")[0] == 'v';   // rounds 1-2: fp32 FMAs
  if (valu) {
    DF3D_CHECK_ARG(weights, "head_final_conv: the vector-ALU kernel (DF3D_HEADFINAL=valu) takes the fp32 filters");
    a.gpad = 0;
    a.tiles_x = cdiv(W, HF_TILE), a.tiles_y = cdiv(H, HF_TILE);
    hipLaunchKernelGGL(head_final_kernel, dim3(batch * a.tiles_x * a.tiles_y, groups), dim3(256), 0, stream, a);
  } else {
    static const int order = getenv("DF3D_HEADFINAL_ORDER") ? atoi(getenv("DF3D_HEADFINAL_ORDER")) : 2;
    a.tiles_x = cdiv(W, HM_TILE), a.tiles_y = cdiv(H, HM_TILE);
    const long long tiles = (long long)batch * a.tiles_x * a.tiles_y;
    a.gpad = order == 1 ? (groups + 7) & ~7 : (order == 2 ? groups : 0);
    if (a.gpad && tiles * a.gpad > 0x7fffffffLL) a.gpad = 0;
    if (a.gpad)
      hipLaunchKernelGGL(head_final_mfma_kernel, dim3((unsigned)(tiles * a.gpad)), dim3(256), 0, stream, a);
    else
      hipLaunchKernelGGL(head_final_mfma_kernel, dim3((unsigned)tiles, groups), dim3(256), 0, stream, a);
  }
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_head_final_conv_backward(const float *acts, int act_channels, const float *grad_out, int out_channels,
                                             int batch, int H, int W, int groups, const float *weights,
                                             const int32_t *out_cols, float *grad_acts, float *grad_weights, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(grad_out && weights && out_cols && (grad_acts || grad_weights), "head_final_conv_backward: null argument");
  DF3D_CHECK_ARG(groups >= 1 && act_channels >= groups * HF_C && act_channels % 4 == 0,
                 "head_final_conv_backward: %d branches of 64 channels do not fit %d-channel rows", groups, act_channels);
  DF3D_CHECK_ARG(batch >= 1 && H >= 1 && W >= 1 && out_channels >= 1, "head_final_conv_backward: bad sizes");
  DF3D_CHECK_ARG(!grad_weights || acts, "head_final_conv_backward: the filter gradient needs the activations");
  HeadFinalBwdArgs a;
  a.m = acts, a.go = grad_out, a.w = weights, a.cols = out_cols, a.gm = grad_acts, a.gw = grad_weights;
  a.ldm = act_channels, a.ldo = out_channels, a.B = batch, a.H = H, a.W = W, a.G = groups;
  a.tiles_x = cdiv(W, HF_TILE), a.tiles_y = cdiv(H, HF_TILE);
  const int ntiles = batch * a.tiles_x * a.tiles_y;
  a.tiles_per_wg = 1;
  if (grad_acts)
    hipLaunchKernelGGL(head_final_bwd_data_kernel, dim3(ntiles, groups), dim3(256), 0, stream, a);
  if (grad_weights) {
    DF3D_HIP(hipMemsetAsync(grad_weights, 0, (size_t)groups * 9 * HF_C * HF_KMAX * sizeof(float), stream));
    a.tiles_per_wg = std::max(1, cdiv(ntiles, 64));             // <= 64 workgroups per branch add into its 2304 filters
    hipLaunchKernelGGL(head_final_bwd_filter_kernel, dim3(cdiv(ntiles, a.tiles_per_wg), groups), dim3(256), 0, stream, a);
  }
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}
