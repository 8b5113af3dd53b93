// Multi-scale deformable attention forward for gfx950.
//
// Semantics = the reference kernel ms_deformable_im2col_gpu_kernel
// (CP/det3d/models/model_utils/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299, bilinear
// helper :33-84): h_im = loc_h*H - 0.5, w_im = loc_w*W - 0.5; a sample contributes iff
// -1 < h_im < H and -1 < w_im < W; corners outside the map read 0;
// out[b,q,m,:] = sum_{l,p} w[b,q,m,l,p] * bilinear(value[b, level l, m, :]).
//
// The reference runs one thread per output CHANNEL (16 lanes repeat the same index math and
// share one 64-byte fetch) in 1024-thread blocks, chunked over the batch by im2col_step.
// Here a lane owns VEC=4 channels (one 16-byte load per corner), so D=16 needs 4 lanes per
// (query, head): a wave covers 16 (q, m) pairs = two whole queries of the 8-head layout, its
// 64 x 16 B stores are one contiguous 1 KiB segment, and the four lanes of a (q, m) group
// split the sampling points between them: each lane fetches one point's (x, y, w) and the
// group exchanges them with wave shuffles instead of re-reading them 16 times.
// Algorithmic bytes: min(N*S*M*D, N*Lq*M*L*P*4*D)*4 + N*Lq*M*L*P*3*4 +
// N*Lq*M*D*4.  Bound: HBM/L2 gather bandwidth (10 flop per fetched element).
#include "common.h"

namespace df3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MsdaArgs {
  const float *value;
  const int64_t *shapes, *lstart;
  const float *loc, *aw;
  float *out;
  int N, S, M, D, Lq, L, P;
};

// Fast path: D % 4 == 0 and (D/4) a power of two <= 16 (D = 4, 8, 16, 32, 64).
template <int LPG /*lanes per (q,m) group = D/4*/>
__global__ __launch_bounds__(256) void msda_vec4_kernel(MsdaArgs a) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over N*Lq*M*LPG
  const long long total = (long long)a.N * a.Lq * a.M * LPG;
  const bool live = gid < total;
  const long long qm = live ? gid / LPG : (total - 1) / LPG;  // (b*Lq + q)*M + m
  const int sub = (int)(gid % LPG);
  const int m = (int)(qm % a.M);
  const int b = (int)(qm / ((long long)a.M * a.Lq));
  const int LP = a.L * a.P;
  const float *loc = a.loc + (size_t)qm * LP * 2;
  const float *aw = a.aw + (size_t)qm * LP;
  const int lane = threadIdx.x & 63;
  const int gbase = lane & ~(LPG - 1);
  const int qstride = a.M * a.D;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int lp0 = 0; lp0 < LP; lp0 += LPG) {
    // lane `sub` of the group fetches point lp0+sub
    float mx = 0.f, my = 0.f, mw = 0.f;
    int lp = lp0 + sub;
    if (lp < LP) {
      mx = loc[lp * 2];
      my = loc[lp * 2 + 1];
      mw = aw[lp];
    }
    const int cnt = (LP - lp0) < LPG ? (LP - lp0) : LPG;
    for (int i = 0; i < cnt; ++i) {
      float lx = __shfl(mx, gbase + i, 64);
      float ly = __shfl(my, gbase + i, 64);
      float w = __shfl(mw, gbase + i, 64);
      int l = (lp0 + i) / a.P;
      int H = (int)a.shapes[l * 2], W = (int)a.shapes[l * 2 + 1];
      const float *vbase = a.value + ((size_t)b * a.S + (size_t)a.lstart[l]) * qstride + m * a.D + sub * 4;
      float h_im = ly * (float)H - 0.5f;
      float w_im = lx * (float)W - 0.5f;
      if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
        int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        int h_high = h_low + 1, w_high = w_low + 1;
        float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
        float hh = 1.f - lh, hw = 1.f - lw;
        f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 v1 = z, v2 = z, v3 = z, v4 = z;
        const size_t hs = (size_t)W * qstride;
        if (h_low >= 0 && w_low >= 0) v1 = *(const f32x4 *)(vbase + h_low * hs + (size_t)w_low * qstride);
        if (h_low >= 0 && w_high <= W - 1) v2 = *(const f32x4 *)(vbase + h_low * hs + (size_t)w_high * qstride);
        if (h_high <= H - 1 && w_low >= 0) v3 = *(const f32x4 *)(vbase + h_high * hs + (size_t)w_low * qstride);
        if (h_high <= H - 1 && w_high <= W - 1) v4 = *(const f32x4 *)(vbase + h_high * hs + (size_t)w_high * qstride);
        float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        acc += (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4) * w;
      }
    }
  }
  if (live) *(f32x4 *)(a.out + (size_t)qm * a.D + sub * 4) = acc;
}

// General path: one thread per output channel (any D).
__global__ __launch_bounds__(256) void msda_scalar_kernel(MsdaArgs a) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)a.N * a.Lq * a.M * a.D;
  if (gid >= total) return;
  const int c = (int)(gid % a.D);
  const long long qm = gid / a.D;
  const int m = (int)(qm % a.M);
  const int b = (int)(qm / ((long long)a.M * a.Lq));
  const int LP = a.L * a.P;
  const float *loc = a.loc + (size_t)qm * LP * 2;
  const float *aw = a.aw + (size_t)qm * LP;
  const int qstride = a.M * a.D;
  float acc = 0.f;
  for (int l = 0; l < a.L; ++l) {
    int H = (int)a.shapes[l * 2], W = (int)a.shapes[l * 2 + 1];
    const float *vbase = a.value + ((size_t)b * a.S + (size_t)a.lstart[l]) * qstride + m * a.D + c;
    for (int p = 0; p < a.P; ++p) {
      float lx = loc[(l * a.P + p) * 2], ly = loc[(l * a.P + p) * 2 + 1], w = aw[l * a.P + p];
      float h_im = ly * (float)H - 0.5f;
      float w_im = lx * (float)W - 0.5f;
      if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
        int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        int h_high = h_low + 1, w_high = w_low + 1;
        float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
        float hh = 1.f - lh, hw = 1.f - lw;
        float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
        const size_t hs = (size_t)W * qstride;
        if (h_low >= 0 && w_low >= 0) v1 = vbase[h_low * hs + (size_t)w_low * qstride];
        if (h_low >= 0 && w_high <= W - 1) v2 = vbase[h_low * hs + (size_t)w_high * qstride];
        if (h_high <= H - 1 && w_low >= 0) v3 = vbase[h_high * hs + (size_t)w_low * qstride];
        if (h_high <= H - 1 && w_high <= W - 1) v4 = vbase[h_high * hs + (size_t)w_high * qstride];
        float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        acc += (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4) * w;
      }
    }
  }
  a.out[gid] = acc;
}


// ---------------------------------------------------------------------------------------------------------
// Backward.  Reference: ms_deformable_col2im_gpu_kernel* and
// ms_deform_attn_col2im_bilinear (ms_deform_im2col_cuda.cuh:87-232, 301-921): per (b, q, m, c) and sampling point
//   grad_value[corner]   += w_corner * attn * g                (atomics; four corners, zero outside the map)
//   grad_attn[b,q,m,l,p] += g * bilinear(value)                (sum over the D channels)
//   grad_loc[..,0]       += W * (dval/dw_im) * attn * g        (x), grad_loc[..,1] += H * (dval/dh_im) * attn * g (y)
// with dval/dh_im = -hw*v1 - lw*v2 + hw*v3 + lw*v4 and dval/dw_im = -hh*v1 + hh*v2 - lh*v3 + lh*v4.
// The reference has seven kernel variants that differ in how the channel sums are reduced (shared memory trees,
// block-size specialisations, global atomics for > 1024 channels).  Here the forward's layout is reused: a lane owns
// four channels, the D/4 lanes of a (q, m) group reduce the two location gradients and the weight gradient with
// wave shuffles (no LDS, no atomics for them), and the value gradient uses the hardware fp32 atomic add.
struct MsdaBwdArgs {
  const float *value;
  const int64_t *shapes, *lstart;
  const float *loc, *aw, *gout;
  float *gvalue, *gloc, *gaw;
  int N, S, M, D, Lq, L, P;
  unsigned char *nz;          // [N * Lq * M]: 1 = the group's upstream gradient is not all zero (binned path)
};

__device__ __forceinline__ void atomic_add4(float *p, f32x4 v) {
  unsafeAtomicAdd(p, v[0]);
  unsafeAtomicAdd(p + 1, v[1]);
  unsafeAtomicAdd(p + 2, v[2]);
  unsafeAtomicAdd(p + 3, v[3]);
}

__device__ __forceinline__ float hsum4(f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }

// BINNED: the value gradient is accumulated by msda_bin_* below (LDS tiles); this kernel then only computes the location /
// weight gradients and notes which (query, head) groups carry a gradient at all (a.nz)
template <int LPG, bool BINNED = false>
__global__ __launch_bounds__(256) void msda_bwd_vec4_kernel(MsdaBwdArgs a) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)a.N * a.Lq * a.M * LPG;
  const bool live = gid < total;
  const long long qm = live ? gid / LPG : (total - 1) / LPG;
  const int sub = (int)(gid % LPG);
  const int m = (int)(qm % a.M);
  const int b = (int)(qm / ((long long)a.M * a.Lq));
  const int LP = a.L * a.P;
  const float *loc = a.loc + (size_t)qm * LP * 2;
  const float *aw = a.aw + (size_t)qm * LP;
  const int lane = threadIdx.x & 63;
  const int gbase = lane & ~(LPG - 1);
  const int qstride = a.M * a.D;
  f32x4 g = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (live) g = *(const f32x4 *)(a.gout + (size_t)qm * a.D + sub * 4);
  // A (query, head) whose upstream gradient is all zero contributes nothing: every product below carries g.  The padded
  // rows of the zero-padded per-camera query lists are exactly that (nobody reads their outputs), and they all sample around
  // reference point (0, 0): ~8 k rows per image adding zeros to the same few cache lines of grad_value (the atomics of one line
  // serialise in the L2: 3.4 ms per call at the TransFusion training shape, 0.12 ms for the forward).
  int nz = (g[0] != 0.f) | (g[1] != 0.f) | (g[2] != 0.f) | (g[3] != 0.f);
#pragma unroll
  for (int d = 1; d < LPG; d <<= 1) nz |= __shfl_xor(nz, d, 64);
  const bool work = live && nz;
  if (BINNED && live && sub == 0) a.nz[qm] = (unsigned char)(nz != 0);
  for (int lp0 = 0; lp0 < LP; lp0 += LPG) {
    float mx = 0.f, my = 0.f, mw = 0.f;
    const int lp = lp0 + sub;
    if (lp < LP) {
      mx = loc[lp * 2];
      my = loc[lp * 2 + 1];
      mw = aw[lp];
    }
    float keep_x = 0.f, keep_y = 0.f, keep_w = 0.f;       // gradients of the point this lane fetched
    const int cnt = (LP - lp0) < LPG ? (LP - lp0) : LPG;
    for (int i = 0; i < cnt; ++i) {
      const float lx = __shfl(mx, gbase + i, 64), ly = __shfl(my, gbase + i, 64), w = __shfl(mw, gbase + i, 64);
      const int l = (lp0 + i) / a.P;
      const int H = (int)a.shapes[l * 2], W = (int)a.shapes[l * 2 + 1];
      const size_t off = ((size_t)b * a.S + (size_t)a.lstart[l]) * qstride + m * a.D + sub * 4;
      const float h_im = ly * (float)H - 0.5f, w_im = lx * (float)W - 0.5f;
      float px = 0.f, py = 0.f, pw = 0.f;
      if (work && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const int h_high = h_low + 1, w_high = w_low + 1;
        const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
        const float hh = 1.f - lh, hw = 1.f - lw;
        const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 v1 = z, v2 = z, v3 = z, v4 = z;
        const size_t hs = (size_t)W * qstride;
        const f32x4 tg = g * w;                              // top_grad * attention weight
        const size_t o1 = off + h_low * hs + (size_t)w_low * qstride, o2 = off + h_low * hs + (size_t)w_high * qstride;
        const size_t o3 = off + h_high * hs + (size_t)w_low * qstride, o4 = off + h_high * hs + (size_t)w_high * qstride;
        if (h_low >= 0 && w_low >= 0) {
          v1 = *(const f32x4 *)(a.value + o1);
          if (!BINNED) atomic_add4(a.gvalue + o1, tg * (hh * hw));
        }
        if (h_low >= 0 && w_high <= W - 1) {
          v2 = *(const f32x4 *)(a.value + o2);
          if (!BINNED) atomic_add4(a.gvalue + o2, tg * (hh * lw));
        }
        if (h_high <= H - 1 && w_low >= 0) {
          v3 = *(const f32x4 *)(a.value + o3);
          if (!BINNED) atomic_add4(a.gvalue + o3, tg * (lh * hw));
        }
        if (h_high <= H - 1 && w_high <= W - 1) {
          v4 = *(const f32x4 *)(a.value + o4);
          if (!BINNED) atomic_add4(a.gvalue + o4, tg * (lh * lw));
        }
        const f32x4 val = (hh * hw) * v1 + (hh * lw) * v2 + (lh * hw) * v3 + (lh * lw) * v4;
        const f32x4 dh = hw * (v3 - v1) + lw * (v4 - v2), dw = hh * (v2 - v1) + lh * (v4 - v3);
        pw = hsum4(g * val);
        px = (float)W * hsum4(dw * tg);
        py = (float)H * hsum4(dh * tg);
      }
#pragma unroll
      for (int d = 1; d < LPG; d <<= 1) {                    // sum over the lanes (channel quads) of the group
        px += __shfl_xor(px, d, 64);
        py += __shfl_xor(py, d, 64);
        pw += __shfl_xor(pw, d, 64);
      }
      if (sub == i) {
        keep_x = px;
        keep_y = py;
        keep_w = pw;
      }
    }
    if (live && lp < LP) {
      a.gloc[((size_t)qm * LP + lp) * 2] = keep_x;
      a.gloc[((size_t)qm * LP + lp) * 2 + 1] = keep_y;
      a.gaw[(size_t)qm * LP + lp] = keep_w;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Value gradient WITHOUT global atomics (round 6; heads of 16 channels, one single-level map).  The col2im above issues 16 scalar
// fp32 atomics per lane and sampling point: 0.5 G atomics per call at the TransFusion training shape (24 maps x ~10 k queries x 8
// heads x 4 points x 4 corners x 16 channels), and the L2's atomic units retire ~80 G of them per second whatever their scope or
// spread (tools/debug/msda_bwd_probe.py: 5.0 ms with uniformly spread queries, 5.3 ms with a third of them on one pixel) -- 3 ms
// per layer in the step against 0.12 ms for the forward.  Accumulating tiles in LDS with ds_add_f32 (first attempt) retires
// ~0.4 lanes per clock and CU and collapses where thousands of queries share a pixel (the unseen voxels' reference point).
// The col2im of a tile IS a matrix product:  footprint[81 px][16 ch] = Wt[81 px][points] . G[points][16 ch],  Wt = the
// bilinear weights (four non-zeros per column), G = attention weight x upstream gradient -- dense in the accumulator, so a
// pile-up on one pixel costs nothing.  Hence:
//   1. the sampling points are counting-sorted by (map, 8 x 8 pixel tile of their top-left corner, head): LDS histograms per
//      workgroup, one range reservation per workgroup and bin (msda_bin_kernel, msda_bin_scan_kernel);
//   2. a WAVE per <= 512 points of a bin runs the product on v_mfma_f32_16x16x4_f32 (exact fp32 products): six 16-pixel row
//      tiles of the 9 x 9 footprint (tile + one-pixel halo: the four corners of a point stay inside), A operand = the weights
//      formed in registers from (footprint index, lh, lw), B operand = the gradient row, and stores its footprint as a slab;
//   3. a gather kernel adds the slabs of a tile and the halo row / column / corner of its three neighbours into grad_value:
//      every element written exactly once -- no atomics, no zero fill.
constexpr int MB_T = 8, MB_FOOT = (MB_T + 1) * (MB_T + 1), MB_CHUNK = 512, MB_D = 16;

// Everything below is per MAP n (blockIdx.y): bins, point ranges, work items and slabs are numbered within the map, so no
// prefix sum crosses maps (nbm = tiles * M bins, ppm = Lq * M * LP point slots, mw = nbm + ppm / MB_CHUNK + 1 work items).
struct MsdaBinArgs {
  const float *loc, *aw, *gout;
  const unsigned char *nz;
  float *gvalue;
  int N, S, M, D, Lq, LP, H, W, tx, ty, tiles;
  unsigned *count, *cursor, *offset;   // [N][nbm]: points of a bin; scatter cursor; first point slot
  unsigned *binw;                      // [N][nbm + 1]: first work item of a bin
  unsigned *nwork;                     // [N]
  unsigned *work;                      // [N][mw][3]: (bin, first, last)
  unsigned *items;                     // [N][ppm]: packed (q << 4 | lp), bin after bin
  int mw;
  float *slabs;                        // [N][mw][MB_FOOT][16]: the work items' footprints
};

__device__ __forceinline__ bool mb_point(const MsdaBinArgs &a, float lx, float ly, int &hl, int &wl, float &lh, float &lw) {
  const float h_im = ly * (float)a.H - 0.5f, w_im = lx * (float)a.W - 0.5f;
  const bool in = h_im > -1.f && w_im > -1.f && h_im < (float)a.H && w_im < (float)a.W;
  hl = (int)floorf(h_im), wl = (int)floorf(w_im);
  lh = h_im - (float)hl, lw = w_im - (float)wl;
  return in;
}
__device__ __forceinline__ int mb_tile(const MsdaBinArgs &a, int hl, int wl) {
  return (max(hl, 0) / MB_T) * a.tx + max(wl, 0) / MB_T;
}

// pass 1 (SCATTER = false): points per (tile, head) of map blockIdx.y; pass 2 (SCATTER = true): the points' packed ids into
// their bin's range.  A thread owns one (q, m) group and walks its L * P points.
template <bool SCATTER>
__global__ __launch_bounds__(256) void msda_bin_kernel(MsdaBinArgs a) {
  extern __shared__ unsigned mb_hist[];                      // [nbm] counts, then (scatter) [nbm] bases
  const int n = blockIdx.y, tid = threadIdx.x, nbm = a.tiles * a.M;
  for (int t = tid; t < nbm; t += 256) mb_hist[t] = 0u;
  __syncthreads();
  const long long g = (long long)blockIdx.x * 256 + tid;     // (q, m) of this map
  const bool live = g < (long long)a.Lq * a.M && a.nz[(size_t)n * a.Lq * a.M + g];
  const int m = (int)(g % a.M);
  const float *loc = a.loc + ((size_t)n * a.Lq * a.M + (size_t)(live ? g : 0)) * a.LP * 2;
  unsigned rank[16];
  int bin[16];
#pragma unroll 1
  for (int lp = 0; lp < a.LP; ++lp) {
    bin[lp & 15] = -1;
    if (!live) continue;
    int hl, wl;
    float lh, lw;
    if (!mb_point(a, loc[lp * 2], loc[lp * 2 + 1], hl, wl, lh, lw)) continue;
    const int t = mb_tile(a, hl, wl) * a.M + m;
    bin[lp & 15] = t;
    rank[lp & 15] = atomicAdd(&mb_hist[t], 1u);
  }
  __syncthreads();
  if (!SCATTER) {
    for (int t = tid; t < nbm; t += 256)
      if (mb_hist[t]) atomicAdd(&a.count[(size_t)n * nbm + t], mb_hist[t]);
    return;
  }
  unsigned *base = mb_hist + nbm;
  for (int t = tid; t < nbm; t += 256)
    base[t] = mb_hist[t] ? a.offset[(size_t)n * nbm + t] + atomicAdd(&a.cursor[(size_t)n * nbm + t], mb_hist[t]) : 0u;
  __syncthreads();
  if (!live) return;
  const unsigned q = (unsigned)(g / a.M);
  unsigned *items = a.items + (size_t)n * a.Lq * a.M * a.LP;
  for (int lp = 0; lp < a.LP; ++lp)
    if (bin[lp & 15] >= 0) items[base[bin[lp & 15]] + rank[lp & 15]] = (q << 4) | (unsigned)lp;
}

// exclusive scan of one map's bin counts + its work list: a bin with c points becomes ceil(c / MB_CHUNK) work items.  1024
// threads, <= 8 consecutive bins per thread (nbm <= 7680), the (points, work items) pair scanned through wave shuffles.
__global__ __launch_bounds__(1024) void msda_bin_scan_kernel(MsdaBinArgs a) {
  __shared__ unsigned long long s_wave[16];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nbm = a.tiles * a.M;
  const int per = (nbm + 1023) / 1024, b0 = min(tid * per, nbm), b1 = min(b0 + per, nbm);
  const unsigned *count = a.count + (size_t)n * nbm;
  unsigned c[8];
  unsigned long long mine = 0;                                 // low word: points; high word: work items
  for (int b = b0, k = 0; b < b1; ++b, ++k) {
    c[k] = count[b];
    mine += (unsigned long long)c[k] | ((unsigned long long)((c[k] + MB_CHUNK - 1) / MB_CHUNK) << 32);
  }
  unsigned long long incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long up = __shfl_up(incl, d);
    if (lane >= d) incl += up;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  unsigned long long before = incl - mine;
  for (int w = 0; w < wave; ++w) before += s_wave[w];
  unsigned run = (unsigned)before, wrun = (unsigned)(before >> 32);
  unsigned *offset = a.offset + (size_t)n * nbm, *binw = a.binw + (size_t)n * (nbm + 1), *work = a.work + (size_t)n * a.mw * 3;
  for (int b = b0, k = 0; b < b1; ++b, ++k) {
    offset[b] = run;
    binw[b] = wrun;
    for (unsigned s0 = 0; s0 < c[k]; s0 += MB_CHUNK, ++wrun)
      work[wrun * 3] = b, work[wrun * 3 + 1] = run + s0, work[wrun * 3 + 2] = run + min(s0 + (unsigned)MB_CHUNK, c[k]);
    run += c[k];
  }
  if (tid == 1023) binw[nbm] = wrun, a.nwork[n] = wrun;
}

// a wave = one work item: <= MB_CHUNK points of one (map, tile, head).  MFMA operand layout (v_mfma_f32_16x16x4_f32): lane
// (i = lane & 15, k = lane >> 4) holds A[i][k] and B[k][i]; the accumulator of row tile rt holds footprint pixels
// rt * 16 + 4 * (lane >> 4) + r of channel lane & 15.  Four points per matrix instruction, sixteen per loop pass; the ids of
// pass k + 2 and the locations / weights / gradient rows of pass k + 1 are in flight while pass k multiplies.
struct MbPoints {
  float lx[4], ly[4], aw[4], g[4];
};
__global__ __launch_bounds__(256) void msda_bin_accumulate_kernel(MsdaBinArgs a) {
  const int n = blockIdx.y, lane = threadIdx.x & 63, wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (int)a.nwork[n]) return;
  const unsigned *work = a.work + ((size_t)n * a.mw + wid) * 3;
  const unsigned bin = work[0], first = work[1], last = work[2];
  const int t = bin / a.M, m = bin - t * a.M;
  const int y0 = (t / a.tx) * MB_T, x0 = (t % a.tx) * MB_T;
  const int i16 = lane & 15, kg = lane >> 4;
  const unsigned *items = a.items + (size_t)n * a.Lq * a.M * a.LP;
  auto load_ids = [&](unsigned i0, unsigned(&id)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) id[u] = items[min(i0 + 4 * u + kg, last - 1)];
  };
  auto load_points = [&](const unsigned(&id)[4], MbPoints &v) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t qm = ((size_t)n * a.Lq + (id[u] >> 4)) * a.M + m, lp = id[u] & 15u;
      const float2 l = *(const float2 *)(a.loc + (qm * a.LP + lp) * 2);
      v.lx[u] = l.x, v.ly[u] = l.y;
      v.aw[u] = a.aw[qm * a.LP + lp];
      v.g[u] = a.gout[qm * MB_D + i16];
    }
  };
  f32x4 acc[6];
#pragma unroll
  for (int rt = 0; rt < 6; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int prow[6], pcol[6];                                        // footprint row / column of this lane's pixel in row tile rt
#pragma unroll
  for (int rt = 0; rt < 6; ++rt) prow[rt] = (rt * 16 + i16) / (MB_T + 1), pcol[rt] = (rt * 16 + i16) % (MB_T + 1);
  unsigned id[4];
  MbPoints cur, nxt;
  load_ids(first, id);
  load_points(id, cur);
  load_ids(first + 16, id);
  for (unsigned i0 = first; i0 < last; i0 += 16) {
    load_points(id, nxt);
    load_ids(i0 + 32, id);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int hl, wl;
      float lh, lw;
      mb_point(a, cur.lx[u], cur.ly[u], hl, wl, lh, lw);      // (in bounds: it was binned)
      const float b = i0 + 4 * u + kg < last ? cur.g[u] * cur.aw[u] : 0.f;
      // the four corner weights are separable: (row weight) x (column weight), a corner outside the map = a zero factor
      const int ry = hl - y0, rx = wl - x0, ry1 = ry + 1, rx1 = rx + 1;      // footprint row / column of the top-left corner
      const float wt = hl >= 0 ? 1.f - lh : 0.f, wb = hl + 1 <= a.H - 1 ? lh : 0.f;
      const float wl_ = wl >= 0 ? 1.f - lw : 0.f, wr = wl + 1 <= a.W - 1 ? lw : 0.f;
#pragma unroll
      for (int rt = 0; rt < 6; ++rt) {
        float wy = prow[rt] == ry ? wt : 0.f, wx = pcol[rt] == rx ? wl_ : 0.f;
        wy = prow[rt] == ry1 ? wb : wy;
        wx = pcol[rt] == rx1 ? wr : wx;
        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wy * wx, b, acc[rt], 0, 0, 0);
      }
    }
    cur = nxt;
  }
  float *slab = a.slabs + ((size_t)n * a.mw + wid) * MB_FOOT * MB_D;
#pragma unroll
  for (int rt = 0; rt < 6; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int px = rt * 16 + 4 * kg + r;
      if (px < MB_FOOT) slab[px * MB_D + i16] = acc[rt][r];
    }
}

// grad_value of one (map, tile): every pixel of the tile's 8 x 8 interior = the sum of that pixel over the tile's own work
// items + the halo row / column / corner of the tiles above, to the left and above-left, head by head.  A pure gather.
__global__ __launch_bounds__(256) void msda_bin_reduce_kernel(MsdaBinArgs a) {
  const int n = blockIdx.y, t = blockIdx.x, tyi = t / a.tx, txi = t - tyi * a.tx, y0 = tyi * MB_T, x0 = txi * MB_T;
  const int C = a.M * MB_D, C4 = C / 4, nbm = a.tiles * a.M;
  const unsigned *binw = a.binw + (size_t)n * (nbm + 1);
  const float *slabs = a.slabs + (size_t)n * a.mw * MB_FOOT * MB_D;
  for (int i = threadIdx.x; i < MB_T * MB_T * C4; i += 256) {
    const int px = i / C4, c = (i - px * C4) * 4, ly = px / MB_T, lx = px - ly * MB_T, m = c / MB_D, cc = c - m * MB_D;
    if (y0 + ly >= a.H || x0 + lx >= a.W) continue;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto add = [&](int b, int fy, int fx) {
      const unsigned w0 = binw[b], w1 = binw[b + 1];
      for (unsigned w = w0; w < w1; ++w)
        acc += *(const f32x4 *)(slabs + ((size_t)w * MB_FOOT + fy * (MB_T + 1) + fx) * MB_D + cc);
    };
    add(t * a.M + m, ly, lx);
    if (ly == 0 && tyi > 0) add((t - a.tx) * a.M + m, MB_T, lx);
    if (lx == 0 && txi > 0) add((t - 1) * a.M + m, ly, MB_T);
    if (ly == 0 && lx == 0 && tyi > 0 && txi > 0) add((t - a.tx - 1) * a.M + m, MB_T, MB_T);
    *(f32x4 *)(a.gvalue + ((size_t)n * a.S + (size_t)(y0 + ly) * a.W + x0 + lx) * C + c) = acc;
  }
}

// any D: one thread per channel, global atomics for all three gradients (the reference's fallback,
// ms_deform_im2col_cuda.cuh:818-921)
__global__ __launch_bounds__(256) void msda_bwd_scalar_kernel(MsdaBwdArgs a) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)a.N * a.Lq * a.M * a.D;
  if (gid >= total) return;
  const int c = (int)(gid % a.D);
  const long long qm = gid / a.D;
  const int m = (int)(qm % a.M);
  const int b = (int)(qm / ((long long)a.M * a.Lq));
  const int LP = a.L * a.P;
  const int qstride = a.M * a.D;
  const float g = a.gout[gid];
  for (int lp = 0; lp < LP; ++lp) {
    const int l = lp / a.P;
    const int H = (int)a.shapes[l * 2], W = (int)a.shapes[l * 2 + 1];
    const float lx = a.loc[((size_t)qm * LP + lp) * 2], ly = a.loc[((size_t)qm * LP + lp) * 2 + 1];
    const float w = a.aw[(size_t)qm * LP + lp];
    const float h_im = ly * (float)H - 0.5f, w_im = lx * (float)W - 0.5f;
    if (!(h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W)) continue;
    const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im), h_high = h_low + 1, w_high = w_low + 1;
    const float lh = h_im - (float)h_low, lw = w_im - (float)w_low, hh = 1.f - lh, hw = 1.f - lw;
    const size_t off = ((size_t)b * a.S + (size_t)a.lstart[l]) * qstride + m * a.D + c, hs = (size_t)W * qstride;
    float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
    const float tg = g * w;
    if (h_low >= 0 && w_low >= 0) {
      v1 = a.value[off + h_low * hs + (size_t)w_low * qstride];
      unsafeAtomicAdd(a.gvalue + off + h_low * hs + (size_t)w_low * qstride, tg * hh * hw);
    }
    if (h_low >= 0 && w_high <= W - 1) {
      v2 = a.value[off + h_low * hs + (size_t)w_high * qstride];
      unsafeAtomicAdd(a.gvalue + off + h_low * hs + (size_t)w_high * qstride, tg * hh * lw);
    }
    if (h_high <= H - 1 && w_low >= 0) {
      v3 = a.value[off + h_high * hs + (size_t)w_low * qstride];
      unsafeAtomicAdd(a.gvalue + off + h_high * hs + (size_t)w_low * qstride, tg * lh * hw);
    }
    if (h_high <= H - 1 && w_high <= W - 1) {
      v4 = a.value[off + h_high * hs + (size_t)w_high * qstride];
      unsafeAtomicAdd(a.gvalue + off + h_high * hs + (size_t)w_high * qstride, tg * lh * lw);
    }
    unsafeAtomicAdd(a.gaw + (size_t)qm * LP + lp, g * (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4));
    unsafeAtomicAdd(a.gloc + ((size_t)qm * LP + lp) * 2, (float)W * tg * (hh * (v2 - v1) + lh * (v4 - v3)));
    unsafeAtomicAdd(a.gloc + ((size_t)qm * LP + lp) * 2 + 1, (float)H * tg * (hw * (v3 - v1) + lw * (v4 - v2)));
  }
}

}  // namespace df3d

using namespace df3d;

extern "C" int df3d_ms_deform_attn_forward(const float *value, const int64_t *spatial_shapes,
                                           const int64_t *level_start_index, const float *sampling_loc,
                                           const float *attn_weight, int N, int S, int M, int D, int Lq, int L, int P,
                                           float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(value && spatial_shapes && level_start_index && sampling_loc && attn_weight && out,
                 "ms_deform_attn_forward: null argument");
  DF3D_CHECK_ARG(N >= 0 && S > 0 && M > 0 && D > 0 && Lq >= 0 && L > 0 && P > 0, "ms_deform_attn_forward: bad sizes");
  if (N == 0 || Lq == 0) return DF3D_OK;
  MsdaArgs a = {value, spatial_shapes, level_start_index, sampling_loc, attn_weight, out, N, S, M, D, Lq, L, P};
  int lpg = (D % 4 == 0) ? D / 4 : 0;
  long long total;
  switch (lpg) {
#define DF3D_MSDA_CASE(G)                                                                                     \
  case G:                                                                                                     \
    total = (long long)N * Lq * M * G;                                                                        \
    hipLaunchKernelGGL(msda_vec4_kernel<G>, dim3(cdiv(total, 256)), dim3(256), 0, stream, a);                 \
    break;
    DF3D_MSDA_CASE(1)
    DF3D_MSDA_CASE(2)
    DF3D_MSDA_CASE(4)
    DF3D_MSDA_CASE(8)
    DF3D_MSDA_CASE(16)
#undef DF3D_MSDA_CASE
    default:
      total = (long long)N * Lq * M * D;
      hipLaunchKernelGGL(msda_scalar_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, a);
  }
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_ms_deform_attn_backward(const float *value, const int64_t *spatial_shapes,
                                            const int64_t *level_start_index, const float *sampling_loc,
                                            const float *attn_weight, const float *grad_output, int N, int S, int M, int D,
                                            int Lq, int L, int P, float *grad_value, float *grad_sampling_loc,
                                            float *grad_attn_weight, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(N >= 0 && S > 0 && M > 0 && D > 0 && Lq >= 0 && L > 0 && P > 0, "ms_deform_attn_backward: bad sizes");
  if (N == 0) return DF3D_OK;
  DF3D_CHECK_ARG(value && spatial_shapes && level_start_index && grad_value, "ms_deform_attn_backward: null argument");
  DF3D_HIP(hipMemsetAsync(grad_value, 0, (size_t)N * S * M * D * sizeof(float), stream));
  if (Lq == 0) return DF3D_OK;
  DF3D_CHECK_ARG(sampling_loc && attn_weight && grad_output && grad_sampling_loc && grad_attn_weight,
                 "ms_deform_attn_backward: null argument");
  MsdaBwdArgs a = {value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                   grad_value, grad_sampling_loc, grad_attn_weight, N, S, M, D, Lq, L, P, nullptr};
  const int lpg = (D % 4 == 0) ? D / 4 : 0;
  long long total;
  switch (lpg) {
#define DF3D_MSDA_BWD_CASE(G)                                                                                 \
  case G:                                                                                                     \
    total = (long long)N * Lq * M * G;                                                                        \
    hipLaunchKernelGGL(msda_bwd_vec4_kernel<G>, dim3(cdiv(total, 256)), dim3(256), 0, stream, a);             \
    break;
    DF3D_MSDA_BWD_CASE(1)
    DF3D_MSDA_BWD_CASE(2)
    DF3D_MSDA_BWD_CASE(4)
    DF3D_MSDA_BWD_CASE(8)
    DF3D_MSDA_BWD_CASE(16)
#undef DF3D_MSDA_BWD_CASE
    default: {
      const size_t nlp = (size_t)N * Lq * M * L * P;
      DF3D_HIP(hipMemsetAsync(grad_sampling_loc, 0, nlp * 2 * sizeof(float), stream));
      DF3D_HIP(hipMemsetAsync(grad_attn_weight, 0, nlp * sizeof(float), stream));
      total = (long long)N * Lq * M * D;
      hipLaunchKernelGGL(msda_bwd_scalar_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, a);
    }
  }
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

// ---- binned backward (single-level maps, 16-channel heads): see msda_bin_* above ----------------------------------------------
static size_t mb_align(size_t v) { return (v + 255) & ~(size_t)255; }
static size_t mb_nbm(int M, int H, int W) { return (size_t)cdiv(H, MB_T) * cdiv(W, MB_T) * M; }
static size_t mb_mw(int M, int Lq, int LP, int H, int W) { return mb_nbm(M, H, W) + (size_t)Lq * M * LP / MB_CHUNK + 1; }

extern "C" size_t df3d_ms_deform_attn_backward_binned_workspace_bytes(int N, int M, int Lq, int LP, int H, int W) {
  const size_t nbm = mb_nbm(M, H, W), n = (size_t)N;
  return mb_align(n * Lq * M) + mb_align(n * nbm * 4) * 3 + mb_align(n * (nbm + 1) * 4) + mb_align(n * 4) +
         mb_align(n * mb_mw(M, Lq, LP, H, W) * 12) + mb_align(n * Lq * M * LP * 4);
}
/* + the work items' footprints: N * mw * 81 * 16 floats, mw = tiles * M + Lq * M * P / 512 + 1 work items per map */
extern "C" size_t df3d_ms_deform_attn_backward_binned_slab_bytes(int N, int M, int D, int Lq, int LP, int H, int W) {
  (void)D;
  return (size_t)N * mb_mw(M, Lq, LP, H, W) * MB_FOOT * MB_D * 4;
}

extern "C" int df3d_ms_deform_attn_backward_binned(const float *value, const int64_t *spatial_shapes, const int64_t *level_start_index,
                                                   const float *sampling_loc, const float *attn_weight, const float *grad_output,
                                                   int N, int M, int D, int Lq, int P, int H, int W, float *grad_value,
                                                   float *grad_sampling_loc, float *grad_attn_weight, void *workspace,
                                                   size_t workspace_bytes, float *slabs, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(N >= 0 && M > 0 && D > 0 && Lq >= 0 && P > 0 && H > 0 && W > 0, "ms_deform_attn_backward_binned: bad sizes");
  DF3D_CHECK_ARG(D == MB_D && P <= 16 && Lq < (1 << 28), "ms_deform_attn_backward_binned: head width %d / %d points not served", D, P);
  const int S = H * W, C = M * D, tx = cdiv(W, MB_T), ty = cdiv(H, MB_T), tiles = tx * ty;
  DF3D_CHECK_ARG((size_t)tiles * M <= 7680, "ms_deform_attn_backward_binned: %d tiles x %d heads exceed the LDS budget", tiles, M);
  DF3D_CHECK_ARG(N <= 65535 && (long long)Lq * M * P < (1LL << 31), "ms_deform_attn_backward_binned: %d maps x %d queries", N, Lq);
  if (N == 0) return DF3D_OK;
  DF3D_CHECK_ARG(value && spatial_shapes && level_start_index && grad_value, "ms_deform_attn_backward_binned: null argument");
  if (Lq == 0) {
    DF3D_HIP(hipMemsetAsync(grad_value, 0, (size_t)N * S * C * sizeof(float), stream));
    return DF3D_OK;
  }
  DF3D_CHECK_ARG(sampling_loc && attn_weight && grad_output && grad_sampling_loc && grad_attn_weight && workspace && slabs,
                 "ms_deform_attn_backward_binned: null argument");
  DF3D_CHECK_ARG(workspace_bytes >= df3d_ms_deform_attn_backward_binned_workspace_bytes(N, M, Lq, P, H, W),
                 "ms_deform_attn_backward_binned: workspace too small");
  const size_t nbm = mb_nbm(M, H, W), n = (size_t)N;
  const int mw = (int)mb_mw(M, Lq, P, H, W);
  char *w = (char *)workspace;
  unsigned char *nz = (unsigned char *)w;
  w += mb_align(n * Lq * M);
  unsigned *count = (unsigned *)w;
  w += mb_align(n * nbm * 4);
  unsigned *cursor = (unsigned *)w;                            // (adjacent to `count`: one memset clears both)
  w += mb_align(n * nbm * 4);
  unsigned *offset = (unsigned *)w;
  w += mb_align(n * nbm * 4);
  unsigned *binw = (unsigned *)w;
  w += mb_align(n * (nbm + 1) * 4);
  unsigned *nwork = (unsigned *)w;
  w += mb_align(n * 4);
  unsigned *work = (unsigned *)w;
  w += mb_align(n * mw * 12);
  unsigned *items = (unsigned *)w;
  DF3D_HIP(hipMemsetAsync(count, 0, mb_align(n * nbm * 4) * 2, stream));
  // location / weight gradients + the groups that carry a gradient (the gather half of the col2im)
  MsdaBwdArgs g = {value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                   grad_value, grad_sampling_loc, grad_attn_weight, N, S, M, D, Lq, 1, P, nz};
  const long long total = (long long)N * Lq * M * (D / 4);
  hipLaunchKernelGGL((msda_bwd_vec4_kernel<MB_D / 4, true>), dim3(cdiv(total, 256)), dim3(256), 0, stream, g);
  MsdaBinArgs b = {sampling_loc, attn_weight, grad_output, nz, grad_value, N, S, M, D, Lq, P, H, W, tx, ty, tiles,
                   count, cursor, offset, binw, nwork, work, items, mw, slabs};
  const dim3 bgrid(cdiv((long long)Lq * M, 256), N);
  hipLaunchKernelGGL(msda_bin_kernel<false>, bgrid, dim3(256), nbm * 4, stream, b);
  hipLaunchKernelGGL(msda_bin_scan_kernel, dim3(N), dim3(1024), 0, stream, b);
  hipLaunchKernelGGL(msda_bin_kernel<true>, bgrid, dim3(256), nbm * 8, stream, b);
  hipLaunchKernelGGL(msda_bin_accumulate_kernel, dim3(cdiv(mw, 4), N), dim3(256), 0, stream, b);
  hipLaunchKernelGGL(msda_bin_reduce_kernel, dim3(tiles, N), dim3(256), 0, stream, b);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}
