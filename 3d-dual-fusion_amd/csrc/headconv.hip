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

namespace df3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int HF_TILE = 8, HF_HALO = HF_TILE + 2, HF_HP = HF_HALO * HF_HALO, HF_C = 64, HF_KMAX = 4;
constexpr int HF_LD = HF_HP + 1;                       // pixels per channel row in LDS (+1: rows of 10 shift the banks)

struct HeadFinalArgs {
  const u32x4 *in;       // split rows [B*H*W][ld] (u32x4 units), branch g at columns g*16 .. g*16+15 (64 channels)
  const float *w;        // [G][9][64][HF_KMAX] fp32, output maps padded to 4
  const float *bias;     // [G][HF_KMAX]
  const int32_t *cols;   // [G][2] (first output column, valid maps)
  float *out;            // [B*H*W][ldo]
  int ld, ldo, B, H, W, G, tiles_x, tiles_y;
};

__device__ __forceinline__ float bf16_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf16_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

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
  // ---- halo -> LDS: item = (halo pixel, 8-channel block): 32 B of split row = hi 8 x bf16 | lo 8 x bf16.  All loads of a
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
      x[(blk * 8 + 2 * e) * HF_LD + p] = bf16_lo(hi[i][e]) + bf16_lo(lo[i][e]);
      x[(blk * 8 + 2 * e + 1) * HF_LD + p] = bf16_hi(hi[i][e]) + bf16_hi(lo[i][e]);
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

}  // namespace df3d

using namespace df3d;

extern "C" int df3d_head_final_conv(const void *in_split, int in_channels, int batch, int H, int W, int groups,
                                    const float *weights, const float *bias, const int32_t *out_cols, float *out,
                                    int out_channels, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(in_split && weights && bias && out_cols && out, "head_final_conv: null argument");
  DF3D_CHECK_ARG(groups >= 1 && in_channels >= groups * HF_C && in_channels % 8 == 0,
                 "head_final_conv: %d branches of 64 channels do not fit %d-channel rows", groups, in_channels);
  DF3D_CHECK_ARG(batch >= 1 && H >= 1 && W >= 1 && out_channels >= 1, "head_final_conv: bad sizes");
  HeadFinalArgs a;
  a.in = (const u32x4 *)in_split;
  a.w = weights;
  a.bias = bias;
  a.cols = out_cols;
  a.out = out;
  a.ld = in_channels / 4;                              // u32x4 per split row: 16 bytes carry 4 channels' worth (hi+lo of 8 per 32 B)
  a.ldo = out_channels;
  a.B = batch, a.H = H, a.W = W, a.G = groups;
  a.tiles_x = cdiv(W, HF_TILE), a.tiles_y = cdiv(H, HF_TILE);
  hipLaunchKernelGGL(head_final_kernel, dim3(batch * a.tiles_x * a.tiles_y, groups), dim3(256), 0, stream, a);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}
