// Sparse convolution backward.
//
// Reference: indiceConvBackward (TF/mmdet3d/ops/spconv/include/spconv/spconv_ops.h:363-456): per kernel offset k --
// gather the input rows and the output-gradient rows of its rulebook pairs into two buffers, filtersGrad[k] =
// in_buf^T . out_buf (GEMM), in_buf = out_buf . W[k]^T (GEMM), scatter-add in_buf into inputGrad; 27 x (2 gathers +
// 2 GEMMs + 1 scatter-add) launches and a D2H copy of the pair counts.
//
// Here:
//   input gradient   = the FORWARD kernel on the output gradient with transposed filters and the inverse neighbour
//                      table inv[k][i] = o (df3d_invert_neighbors; for submanifold convolutions the inverse of offset
//                      k is the forward table of the mirrored offset K-1-k, so nothing is built at all).  Output-
//                      stationary in the input rows: no atomics, no scatter pass, fused into one launch.
//   filter gradient  = df3d_sparse_conv_grad_filters: wgrad_f32_kernel (further down; channel counts that are multiples
//                      of 4) or grad_filters_kernel: grid (row slice, offset, 64-channel group); a wave owns a
//                      16-input-channel x COUT tile of filtersGrad[k] in fp32 MFMA accumulators
//                      (v_mfma_f32_16x16x4_f32: the contraction runs over ROWS, 4 per instruction, operands are
//                      read straight from the feature / gradient rows -- 16 consecutive floats per row and lane
//                      group, no transposition), and adds its partial tile to HBM with fp32 atomics once.
#include "common.h"
#include <algorithm>
#include <cstdlib>

namespace df3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

DF3D_SPLIT_OVERFLOW_TU(spconv_bwd)

__global__ __launch_bounds__(256) void invert_fill_kernel(int32_t *__restrict__ inv, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) inv[i] = -1;
}

__global__ __launch_bounds__(256) void invert_neighbors_kernel(const int32_t *__restrict__ nbr, int kvol, int n_out,
                                                               int n_in, int32_t *__restrict__ inv) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)kvol * n_out) return;
  const int k = (int)(i / n_out), o = (int)(i - (size_t)k * n_out);
  const int src = nbr[i];
  if (src >= 0 && src < n_in) inv[(size_t)k * n_in + src] = o;       // a (k, input) pair has at most one output
}

constexpr int GF_ROWS = 1024;      // output rows per workgroup slice

template <int CT /*16-wide output-channel tiles*/>
__global__ __launch_bounds__(256) void grad_filters_kernel(const float *__restrict__ feat, const float *__restrict__ gout,
                                                           const int32_t *__restrict__ nbr, int n_out, int cin, int cout,
                                                           float *__restrict__ gw) {
  const int k = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ci_tile = blockIdx.z * 4 + wave;
  if (ci_tile * 16 >= cin) return;
  const int m = lane & 15, kg = lane >> 4;
  const int ci = ci_tile * 16 + m;
  const bool ci_ok = ci < cin;
  const int r0 = blockIdx.x * GF_ROWS, r1 = min(r0 + GF_ROWS, n_out);
  const int32_t *nb = nbr + (size_t)k * n_out;
  f32x4 acc[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bool any = false;
  for (int base = r0; base < r1; base += 16) {         // 4 MFMA k-steps per iteration: rows base + 4*u + kg
    int idx[4];
    float a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int o = base + 4 * u + kg;
      idx[u] = o < r1 ? nb[o] : -1;
    }
    if (__ballot(idx[0] >= 0 || idx[1] >= 0 || idx[2] >= 0 || idx[3] >= 0) == 0ull) continue;   // no pair in 16 rows
    any = true;
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = (idx[u] >= 0 && ci_ok) ? feat[(size_t)idx[u] * cin + ci] : 0.f;
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      const int co = t * 16 + m;
      float b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int o = base + 4 * u + kg;
        b[u] = (idx[u] >= 0 && co < cout) ? gout[(size_t)o * cout + co] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc[t], 0, 0, 0);
    }
  }
  if (!any) return;
  // accumulator lane (col j = lane & 15, rows 4*(lane >> 4) + r): D[i][j] = sum_rows in[row][ci_tile*16 + i] * gout[row][t*16 + j]
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const int co = t * 16 + m;
    if (co >= cout) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = ci_tile * 16 + 4 * kg + r;
      if (i < cin && acc[t][r] != 0.f) unsafeAtomicAdd(gw + ((size_t)k * cin + i) * cout + co, acc[t][r]);
    }
  }
}


// ---- filter gradient, second kernel: rows staged through LDS, pairs compacted, software-prefetched -------------------------
// grad_filters_kernel above issues its operand loads (one float per lane and row) right before the matrix instruction that
// consumes them and multiplies the rows WITHOUT a pair as zeros: at the 128 -> 128 layer of conv4 it runs at a fifth of the
// fp32 matrix rate (450 us, 17 GFLOP of pairs).  Here a workgroup owns (row slice, offset k, 16RT x 16CT block of
// filtersGrad[k]); each of its four waves walks a quarter of the slice on its own -- no workgroup barrier in the loop:
//   1. 256 rows of nbr[k] at a time (loaded a batch ahead) -> the (input row, output row) PAIRS, compacted into a
//      wave-private LDS list;
//   2. 16 pairs per stage: the input rows' 16RT channels and the gradient rows' 16CT channels as 16-byte loads into
//      registers (issued one stage ahead), stored row-major into a wave-private LDS tile whose row stride (channels + 16
//      floats) spreads the four rows a fragment read touches over all 64 banks;
//   3. four k-steps of v_mfma_f32_16x16x4_f32 (exact fp32 products) per stage, RT x CT accumulator tiles per wave.
// The four waves' tiles meet in LDS and leave as one atomic add per element and workgroup.
struct WgradArgs {
  const float *feat, *gout;
  const int32_t *nbr;
  float *gw;
  int n_out, cin, cout, slice, tiles_n, tiles, kvol, slices;
  unsigned char order[DF3D_MAX_KVOL];   // offsets, the ones with the most pairs first
  int dbg;                              // wgrad_split3_kernel tuning experiments (DF3D_W3_DBG): 1 no atomics, 2 no MFMAs, 4 no split + LDS stores
  const float *sa, *sg;                 // two-part form: device scales of `feat` / `gout` (NULL: the fixed activation scale 2^5)
};

constexpr int WG_STAGE = 16, WG_BATCH = 256, WG_LIST = WG_BATCH + 64;

template <int RT, int CT>
__global__ __launch_bounds__(256) void wgrad_f32_kernel(WgradArgs a) {
  constexpr int TM = RT * 16, TN = CT * 16, SA = TM + 16, SG = TN + 16;
  constexpr int WAVE_FLOATS = WG_STAGE * (SA + SG) + 2 * WG_LIST;
  extern __shared__ __align__(16) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float *sA = smem + wave * WAVE_FLOATS, *sG = sA + WG_STAGE * SA;
  int *li = (int *)(sG + WG_STAGE * SG), *lo = li + WG_LIST;
  // Workgroups are dealt to the eight XCDs round robin.  Offset-major, the offsets with the most pairs first (the centre
  // offset of a submanifold table pairs every row, a corner offset one row in ten): the long workgroups start first and
  // the short ones fill the tail of the launch.
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int slices_x = (a.slices + 7) >> 3, per_k = slices_x * a.tiles;
  const int kr = j / per_k, rest = j - kr * per_k;
  const int sl = (rest / a.tiles) * 8 + xcd, tile = rest % a.tiles;
  if (sl >= a.slices) return;
  const int k = a.order[kr];
  const int tm = tile / a.tiles_n, tn = tile - tm * a.tiles_n;
  const int ci0 = tm * TM, co0 = tn * TN;
  const int r0 = sl * a.slice, r1 = min(r0 + a.slice, a.n_out);
  const int quarter = ((r1 - r0 + 255) >> 8) << 6;               // multiple of 64 rows
  const int wr0 = min(r0 + wave * quarter, r1), wr1 = min(wr0 + quarter, r1);
  const int32_t *nb = a.nbr + (size_t)k * a.n_out;
  const int m = lane & 15, kg = lane >> 4;

  f32x4 acc[RT][CT];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // this lane's pieces of a stage: item = lane + 64 j -> (pair, 4-channel piece)
  f32x4 ra[RT], rg[CT];
  unsigned okm = 0u;
  // (no branches around the loads: a missing pair / a channel piece beyond the row reads row 0 / piece 0 and is zeroed)
  auto issue = [&](int st) {
    int ia[RT], ig[CT], og[CT];
    okm = 0u;
#pragma unroll
    for (int j = 0; j < RT; ++j) ia[j] = li[st * WG_STAGE + (lane + 64 * j) / (4 * RT)];
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      ig[j] = li[st * WG_STAGE + (lane + 64 * j) / (4 * CT)];
      og[j] = lo[st * WG_STAGE + (lane + 64 * j) / (4 * CT)];
    }
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      const int c = ci0 + ((lane + 64 * j) % (4 * RT)) * 4;
      const bool ok = ia[j] >= 0 && c < a.cin;
      ra[j] = *(const f32x4 *)(a.feat + (size_t)max(ia[j], 0) * a.cin + (c < a.cin ? c : 0));
      okm |= ok ? 1u << j : 0u;                                  // zeroed when stashed (keeps the load out of a branch)
    }
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      const int c = co0 + ((lane + 64 * j) % (4 * CT)) * 4;
      const bool ok = ig[j] >= 0 && c < a.cout;
      rg[j] = *(const f32x4 *)(a.gout + (size_t)og[j] * a.cout + (c < a.cout ? c : 0));
      okm |= ok ? 1u << (8 + j) : 0u;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      const int item = lane + 64 * j, pair = item / (4 * RT), c4 = item - pair * (4 * RT);
      *(f32x4 *)(sA + pair * SA + c4 * 4) = (okm >> j & 1u) ? ra[j] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      const int item = lane + 64 * j, pair = item / (4 * CT), c4 = item - pair * (4 * CT);
      *(f32x4 *)(sG + pair * SG + c4 * 4) = (okm >> (8 + j) & 1u) ? rg[j] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };

  int cnt = 0, next = wr0;
  bool had = false;
  int pre[WG_BATCH / 64];                                        // the next 256 rows of the table, loaded a batch ahead
  auto fetch = [&](int base) {
#pragma unroll
    for (int u = 0; u < WG_BATCH / 64; ++u) {
      const int o = base + 64 * u + lane;
      pre[u] = o < wr1 ? nb[o] : -1;
    }
  };
  if (next < wr1) fetch(next);
  while (true) {
    if (next < wr1) {                                            // 256 rows of the table -> pairs (the list holds < 16)
#pragma unroll
      for (int u = 0; u < WG_BATCH / 64; ++u) {
        const int idx = pre[u];
        const unsigned long long mask = __ballot(idx >= 0);
        if (idx >= 0) {
          const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
          li[pos] = idx, lo[pos] = next + 64 * u + lane;
        }
        cnt += __popcll(mask);
      }
      next += WG_BATCH;
      if (next < wr1) fetch(next);
    }
    const bool last = next >= wr1;
    const int stages = last ? (cnt + WG_STAGE - 1) / WG_STAGE : cnt / WG_STAGE;
    if (last && lane < WG_STAGE && cnt + lane < stages * WG_STAGE) li[cnt + lane] = -1, lo[cnt + lane] = r0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (stages > 0) {
      had = true;
      issue(0);
      for (int st = 0; st < stages; ++st) {
        stash();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (st + 1 < stages) issue(st + 1);
#pragma unroll
        for (int q = 0; q < WG_STAGE / 4; ++q) {
          float fa[RT], fb[CT];
#pragma unroll
          for (int i = 0; i < RT; ++i) fa[i] = sA[(4 * q + kg) * SA + i * 16 + m];
#pragma unroll
          for (int j = 0; j < CT; ++j) fb[j] = sG[(4 * q + kg) * SG + j * 16 + m];
#pragma unroll
          for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (last) break;
    const int done = stages * WG_STAGE, rem = cnt - done;         // < 16 pairs wait for the next refill
    int ci_ = 0, co_ = 0;
    if (lane < rem) ci_ = li[done + lane], co_ = lo[done + lane];
    __builtin_amdgcn_wave_barrier();
    if (lane < rem) li[lane] = ci_, lo[lane] = co_;
    cnt = rem;
  }

  // ---- the four waves' tiles -> one tile in LDS -> atomics ----
  if (!__syncthreads_or(had ? 1 : 0)) return;
  // waves 0 / 1 store their tiles into two buffers, waves 2 / 3 add theirs, every thread sums the two at the end
  float *red = smem + (wave & 1) * (TM * TN);
  if (wave < 2) {
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(i * 16 + 4 * kg + r) * TN + j * 16 + m] = acc[i][j][r];
  }
  __syncthreads();
  if (wave >= 2) {
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(i * 16 + 4 * kg + r) * TN + j * 16 + m] += acc[i][j][r];
  }
  __syncthreads();
  for (int idx = tid; idx < TM * TN; idx += 256) {
    const int ci = ci0 + idx / TN, co = co0 + idx % TN;
    const float v = smem[idx] + smem[TM * TN + idx];
    if (v != 0.f && ci < a.cin && co < a.cout) unsafeAtomicAdd(a.gw + ((size_t)k * a.cin + ci) * a.cout + co, v);
  }
}

template <int RT, int CT>
static void launch_wgrad(const WgradArgs &a, int kvol, hipStream_t stream) {
  constexpr int TM = RT * 16, TN = CT * 16;
  constexpr size_t lds = (size_t)4 * (WG_STAGE * (TM + 16 + TN + 16) + 2 * WG_LIST) * sizeof(float);
  static_assert(lds >= (size_t)2 * TM * TN * sizeof(float), "the two reduction tiles must fit the staging area");
  WgradArgs b = a;
  b.tiles = cdiv(a.cin, TM) * a.tiles_n, b.kvol = kvol, b.slices = cdiv(a.n_out, a.slice);
  for (int k = 0; k < kvol; ++k) b.order[k] = (unsigned char)k;
  if (kvol == 27 || kvol == 9) {                                  // 3^d offsets: by the number of non-zero components
    const int d = kvol == 27 ? 3 : 2;
    auto norm = [d](int k) { int s = 0; for (int i = 0; i < d; ++i, k /= 3) s += (k % 3) != 1; return s; };
    std::stable_sort(b.order, b.order + kvol, [&](unsigned char x, unsigned char y) { return norm(x) < norm(y); });
  }
  const dim3 grid(8 * cdiv(b.slices, 8) * kvol * b.tiles);
  hipLaunchKernelGGL((wgrad_f32_kernel<RT, CT>), grid, dim3(256), lds, stream, b);
}

// ---- filter gradient, third kernel (round 5): the contraction over PAIRS on the 16-bit matrix cores ---------------------------
// wgrad_f32_kernel runs v_mfma_f32_16x16x4_f32 at about half of its 157 TFLOP/s: 4.4 ms of the 34 ms training step.  Here both
// operands are split where they are staged into three bf16 parts (x = hi + mid + lo exactly: 24 significand bits and fp32's
// exponent range, so gradients of any magnitude are safe -- the "split3" format of the gradient convolutions) and six of the
// nine part products are accumulated in fp32 (lo*hi, mid*mid, hi*lo, mid*hi, hi*mid, hi*hi: the dropped ones are <= 2^-24 of
// the product): 6 / 16 of the fp32 matrix time per pair.
// The contraction index of filtersGrad[k] = in^T . gout is the PAIR, i.e. both operands are needed K-major while the rows are
// channel-major.  gfx950's transposing LDS read does that for free: the staged rows sit row-major in LDS ([32 pairs][channels]
// bf16 per part, row stride + 32 B so that the four rows a read touches fall on four bank quarters), and
// ds_read_b64_tr_b16 hands lane (c, g) the column c of four rows -- two reads = the 8 k-slots 8 g .. 8 g + 7 of an A (or B)
// operand of v_mfma_f32_16x16x32_bf16 (tools/ubench/trread_probe.hip pins the lane <-> element map).
// A workgroup owns (row slice, offset k, (64 WM) x (64 WN) block of filtersGrad[k]); its WM x WN waves own one 64 x 64 block each
// and share the staged rows: a 128 x 128 layer reads every pair's two rows ONCE (the fp32 kernel's 64 x 64 workgroups read them
// twice).  Pairs are compacted from 256 table rows at a time into a workgroup list; a stage = 32 pairs: global loads a stage
// ahead (registers), split + LDS store, barrier, 48 transposing reads + 96 MFMAs per wave, barrier.  nbr == NULL: pair i = (row
// i, row i) -- the weight gradient of a linear layer over rows (df3d_rows_grad_weights).
typedef short w3_s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 w3_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int w3_u32x2 __attribute__((ext_vector_type(2)));

constexpr int W3_STAGE = 32, W3_BATCH = 256, W3_LIST = W3_BATCH + W3_STAGE;

__device__ __forceinline__ void w3_split_pair(float x0, float x1, unsigned &hi, unsigned &mid, unsigned &lo) {
  unsigned h, m, l, unused;
  split_pair_bf16_ref(x0, x1, h, m);
  const float r0 = (x0 - __uint_as_float(h << 16)) - __uint_as_float(m << 16);
  const float r1 = (x1 - __uint_as_float(h & 0xffff0000u)) - __uint_as_float(m & 0xffff0000u);
  split_pair_bf16_ref(r0, r1, l, unused);
  hi = h, mid = m, lo = l;
}

// NP = 2 (second half of round 5): fp16 PAIRS instead of three bf16 parts -- three products, two stored parts.  Activations
// carry the fixed scale 2^5 of every split kernel; a gradient operand carries the power-of-two block scale of its tensor
// (df3d_split_rows_scaled computes it for the input-gradient convolution of the same layer: a.sa / a.sg point at it), and the
// accumulators are multiplied by 1 / (sa sg) on their way out.  Same staging, same transposing reads, half the MFMAs.
template <int WM, int WN, int NP>
__global__ __launch_bounds__(64 * WM * WN, 2) void wgrad_split3_kernel(WgradArgs a) {
  constexpr int NT = 64 * WM * WN, NWV = WM * WN, TM = 64 * WM, TN = 64 * WN, ST = W3_STAGE;
  // LDS image of one part: row `r` (a pair) at r * RB + (r >> 3) * 128, its 32-byte slots XOR-ed with (r & 3): a
  // transposing read serves 32 lanes per pass = the rows r .. r + 3 of TWO 8-row groups; with this image the eight 32-byte
  // pieces of a pass fall on the eight bank eighths (64 banks x 4 B)
  constexpr int RBA = TM * 2, RBG = TN * 2, IMA = ST * RBA + 512, IMG = ST * RBG + 512;
  constexpr int PA = ST * (TM / 4) / NT, PG = ST * (TN / 4) / NT;  // 4-channel pieces per thread and stage
  static_assert(ST * (TM / 4) % NT == 0 && ST * (TN / 4) % NT == 0, "pieces divide over the workgroup");
  extern __shared__ __align__(16) unsigned char w3_smem[];
  unsigned char *imA = w3_smem, *imG = imA + NP * IMA;
  int *li = (int *)(imG + NP * IMG), *lo = li + W3_LIST;
  int *s_pop = lo + W3_LIST;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float sca = (NP == 2) ? (a.sa ? *a.sa : DF3D_SA_SCALE) : 1.f, scg = (NP == 2) ? (a.sg ? *a.sg : DF3D_SA_SCALE) : 1.f;
  float amax = 0.f;                                              // largest |scaled value| split here (two-part form: range flag)
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int slices_x = (a.slices + 7) >> 3, per_k = slices_x * a.tiles;
  const int kr = j / per_k, rest = j - kr * per_k;
  const int sl = (rest / a.tiles) * 8 + xcd, tile = rest % a.tiles;
  if (sl >= a.slices) return;
  const int k = a.order[kr];
  const int tm = tile / a.tiles_n, tn = tile - tm * a.tiles_n;
  const int ci0 = tm * TM, co0 = tn * TN;
  const int r0 = sl * a.slice, r1 = min(r0 + a.slice, a.n_out);
  const int32_t *nb = a.nbr ? a.nbr + (size_t)k * a.n_out : nullptr;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int li16 = lane & 15, g = lane >> 4;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 ra[PA], rg[PG];
  unsigned okm = 0u;
  auto issue = [&](int st) {
    okm = 0u;
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      const int p = tid + NT * q, pair = p / (TM / 4), c = ci0 + (p - pair * (TM / 4)) * 4;
      const int ia = li[st * ST + pair];
      ra[q] = *(const f32x4 *)(a.feat + (size_t)max(ia, 0) * a.cin + (c < a.cin ? c : 0));
      okm |= (ia >= 0 && c < a.cin) ? 1u << q : 0u;
    }
#pragma unroll
    for (int q = 0; q < PG; ++q) {
      const int p = tid + NT * q, pair = p / (TN / 4), c = co0 + (p - pair * (TN / 4)) * 4;
      const int ia = li[st * ST + pair], og = lo[st * ST + pair];
      rg[q] = *(const f32x4 *)(a.gout + (size_t)og * a.cout + (c < a.cout ? c : 0));
      okm |= (ia >= 0 && c < a.cout) ? 1u << (16 + q) : 0u;
    }
  };
  auto put = [&](unsigned char *im, int rb, int imb, int pair, int c4, f32x4 v, float sc) {
    unsigned char *d = im + pair * rb + (pair >> 3) * 128 + ((c4 * 8) ^ ((pair & 3) * 32));
    if constexpr (NP == 1) {                                   // bf16 mixed precision: one rounded part per operand
      unsigned h0, h1, unused;
      split_pair_bf16_ref(v[0], v[1], h0, unused);
      split_pair_bf16_ref(v[2], v[3], h1, unused);
      *(w3_u32x2 *)d = (w3_u32x2){h0, h1};
    } else if constexpr (NP == 2) {
      unsigned h0, l0, h1, l1;
      v *= sc;
      amax = fmaxf(amax, fmaxf(fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])), fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3]))));
      const float nan_probe = (v[0] + v[1]) + (v[2] + v[3]);     // fmaxf drops a NaN operand: a NaN (or inf - inf)
      if (nan_probe != nan_probe) amax = __builtin_inff();       // anywhere in the piece raises the flag like an overflow
      split_pair_f16_ref<false>(v[0], v[1], 1.f, h0, l0);
      split_pair_f16_ref<false>(v[2], v[3], 1.f, h1, l1);
      *(w3_u32x2 *)d = (w3_u32x2){h0, h1};
      *(w3_u32x2 *)(d + imb) = (w3_u32x2){l0, l1};
    } else {
      unsigned h0, m0, l0, h1, m1, l1;
      w3_split_pair(v[0], v[1], h0, m0, l0);
      w3_split_pair(v[2], v[3], h1, m1, l1);
      *(w3_u32x2 *)d = (w3_u32x2){h0, h1};
      *(w3_u32x2 *)(d + imb) = (w3_u32x2){m0, m1};
      *(w3_u32x2 *)(d + 2 * imb) = (w3_u32x2){l0, l1};
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      const int p = tid + NT * q, pair = p / (TM / 4);
      put(imA, RBA, IMA, pair, p - pair * (TM / 4), (okm >> q & 1u) ? ra[q] : (f32x4){0.f, 0.f, 0.f, 0.f}, sca);
    }
#pragma unroll
    for (int q = 0; q < PG; ++q) {
      const int p = tid + NT * q, pair = p / (TN / 4);
      put(imG, RBG, IMG, pair, p - pair * (TN / 4), (okm >> (16 + q) & 1u) ? rg[q] : (f32x4){0.f, 0.f, 0.f, 0.f}, scg);
    }
  };
  typedef __attribute__((address_space(3))) w3_s16x4 *lds_tr_ptr;
  // fragment of tile t (16 channels), part p: k-slots 8 g .. 8 g + 7 of channel li16 = two transposing reads of four rows each.
  // Lane i of a 16-lane group passes the address of piece i & 3 of row i >> 2 (whose slot XOR is i >> 2)
  auto frag = [&](const unsigned char *base, int rb, int imb, int t, int p) -> w3_bf16x8 {
    const unsigned char *q = base + p * imb + (((t ^ (li16 >> 2)) & 3) * 32);
    const w3_s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(q));
    const w3_s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(q + 4 * rb));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    return __builtin_bit_cast(w3_bf16x8, v);
  };
  const unsigned char *abase = imA + (g * 8 + (li16 >> 2)) * RBA + g * 128 + wm * 128 + (li16 & 3) * 8;
  const unsigned char *gbase = imG + (g * 8 + (li16 >> 2)) * RBG + g * 128 + wn * 128 + (li16 & 3) * 8;
#define W3_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, C, 0, 0, 0)
#define W2_MFMA(A, B, C) DF3D_MFMA_F16(A, B, C)
  auto compute = [&]() {
    if constexpr (NP == 1) {
      w3_bf16x8 af[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) af[mt] = frag(abase, RBA, IMA, mt, 0);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const w3_bf16x8 b0 = frag(gbase, RBG, IMG, nt, 0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = W3_MFMA(af[mt], b0, acc[mt][nt]);
      }
      return;
    }
    if constexpr (NP == 2) {
      w3_bf16x8 af[4][2];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int p = 0; p < 2; ++p) af[mt][p] = frag(abase, RBA, IMA, mt, p);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const w3_bf16x8 b0 = frag(gbase, RBG, IMG, nt, 0), b1 = frag(gbase, RBG, IMG, nt, 1);
        // lo*hi, hi*lo, hi*hi: the summation order of every two-part kernel
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = W2_MFMA(af[mt][1], b0, acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = W2_MFMA(af[mt][0], b1, acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = W2_MFMA(af[mt][0], b0, acc[mt][nt]);
      }
      return;
    }
    w3_bf16x8 af[4][3];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int p = 0; p < 3; ++p) af[mt][p] = frag(abase, RBA, IMA, mt, p);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const w3_bf16x8 b0 = frag(gbase, RBG, IMG, nt, 0), b1 = frag(gbase, RBG, IMG, nt, 1), b2 = frag(gbase, RBG, IMG, nt, 2);
      // six products, smallest terms first (the order of the three-part convolution kernels); product-major over the four
      // row tiles: dependent MFMAs four apart
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = W3_MFMA(af[mt][2], b0, acc[mt][nt]);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = W3_MFMA(af[mt][1], b1, acc[mt][nt]);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = W3_MFMA(af[mt][0], b2, acc[mt][nt]);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = W3_MFMA(af[mt][1], b0, acc[mt][nt]);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = W3_MFMA(af[mt][0], b1, acc[mt][nt]);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = W3_MFMA(af[mt][0], b0, acc[mt][nt]);
    }
  };
#undef W2_MFMA
#undef W3_MFMA

  int cnt = 0, next = r0;
  bool had = false;
  while (true) {
    if (next < r1) {                                             // 256 table rows -> pairs, appended to the list
#pragma unroll 1
      for (int u = 0; u < W3_BATCH / NT; ++u) {
        const int o = next + u * NT + tid;
        const int idx = o < r1 ? (nb ? nb[o] : o) : -1;
        const unsigned long long mask = __ballot(idx >= 0);
        if (lane == 0) s_pop[wave] = __popcll(mask);
        __syncthreads();
        int base = cnt, total = 0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) {
          const int pw = s_pop[w];
          base += w < wave ? pw : 0;
          total += pw;
        }
        if (idx >= 0) {
          const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
          li[pos] = idx, lo[pos] = o;
        }
        cnt += total;
        __syncthreads();
      }
      next += W3_BATCH;
    }
    const bool last = next >= r1;
    const int stages = last ? (cnt + ST - 1) / ST : cnt / ST;
    if (last) {
      if (tid < ST && cnt + tid < stages * ST) li[cnt + tid] = -1, lo[cnt + tid] = r0;
      __syncthreads();
    }
    if (stages > 0) {
      had = true;
      issue(0);
      for (int st = 0; st < stages; ++st) {
        if (!(a.dbg & 4)) stash();
        __syncthreads();
        if (st + 1 < stages) issue(st + 1);
        if (!(a.dbg & 2)) compute();
        __syncthreads();
      }
    }
    if (last) break;
    const int done = stages * ST, rem = cnt - done;               // < 32 pairs wait for the next refill
    int ci_ = 0, co_ = 0;
    if (tid < rem) ci_ = li[done + tid], co_ = lo[done + tid];
    __syncthreads();
    if (tid < rem) li[tid] = ci_, lo[tid] = co_;
    cnt = rem;
    __syncthreads();
  }
  if (NP == 2) split_range_flag_scaled(amax);
  if (!had || (a.dbg & 1)) return;
  const float unscale = NP == 2 ? 1.f / (sca * scg) : 1.f;
  // every wave owns its 64 x 64 block: one atomic add per element and workgroup
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = ci0 + wm * 64 + mt * 16 + 4 * g + r, co = co0 + wn * 64 + nt * 16 + li16;
        const float v = acc[mt][nt][r] * unscale;
        if (v != 0.f && ci < a.cin && co < a.cout) unsafeAtomicAdd(a.gw + ((size_t)k * a.cin + ci) * a.cout + co, v);
      }
}

template <int WM, int WN, int NP>
static int launch_wgrad3(const WgradArgs &a, int kvol, hipStream_t stream) {
  constexpr int TM = 64 * WM, TN = 64 * WN;
  constexpr size_t lds = (size_t)NP * (W3_STAGE * (TM * 2 + TN * 2) + 1024) + (size_t)(2 * W3_LIST + 8) * sizeof(int);
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute((const void *)wgrad_split3_kernel<WM, WN, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      set_error("wgrad_split3: cannot raise the dynamic LDS limit");
      return DF3D_EHIP;
    }
    configured = true;
  }
  WgradArgs b = a;
  b.tiles_n = cdiv(a.cout, TN);
  b.tiles = cdiv(a.cin, TM) * b.tiles_n, b.kvol = kvol;
  // enough workgroups to fill the chip a few times over, slices of at least 1024 rows
  const char *ws = getenv("DF3D_W3_WGS");                          // tuning aid: workgroups wanted per launch
  const int want = std::max(1, cdiv(ws ? atoi(ws) : 2048, kvol * b.tiles));
  const int slices = std::min(want, cdiv(a.n_out, 1024));
  b.dbg = getenv("DF3D_W3_DBG") ? atoi(getenv("DF3D_W3_DBG")) : 0;
  b.slice = cdiv(cdiv(a.n_out, slices), 256) * 256;
  b.slices = cdiv(a.n_out, b.slice);
  for (int k = 0; k < kvol; ++k) b.order[k] = (unsigned char)k;
  if (kvol == 27 || kvol == 9) {
    const int d = kvol == 27 ? 3 : 2;
    auto norm = [d](int k) { int s = 0; for (int i = 0; i < d; ++i, k /= 3) s += (k % 3) != 1; return s; };
    std::stable_sort(b.order, b.order + kvol, [&](unsigned char x, unsigned char y) { return norm(x) < norm(y); });
  }
  const dim3 grid(8 * cdiv(b.slices, 8) * kvol * b.tiles);
  hipLaunchKernelGGL((wgrad_split3_kernel<WM, WN, NP>), grid, dim3(64 * WM * WN), lds, stream, b);
  return DF3D_OK;
}

template <int NP>
static int launch_wgrad3_np(const WgradArgs &a, int kvol, hipStream_t stream) {
  if (a.cin >= 128 && a.cout >= 128) return launch_wgrad3<2, 2, NP>(a, kvol, stream);
  if (a.cin >= 128) return launch_wgrad3<2, 1, NP>(a, kvol, stream);
  if (a.cout >= 128) return launch_wgrad3<1, 2, NP>(a, kvol, stream);
  return launch_wgrad3<1, 1, NP>(a, kvol, stream);
}
// parts = 1: one bf16 part per operand (bf16 mixed-precision training: the operands the forward rounded the same way); 2: fp16
// pairs (a.sa / a.sg = device scales or NULL); 3: bf16 triples
static int launch_wgrad3_any(const WgradArgs &a, int kvol, hipStream_t stream, int parts = 3) {
  if (parts == 1) return launch_wgrad3_np<1>(a, kvol, stream);
  return parts == 2 ? launch_wgrad3_np<2>(a, kvol, stream) : launch_wgrad3_np<3>(a, kvol, stream);
}

template <int RT>
static void launch_wgrad_ct(WgradArgs &a, int kvol, hipStream_t stream) {
  if (a.cout >= 64) a.tiles_n = cdiv(a.cout, 64), launch_wgrad<RT, 4>(a, kvol, stream);
  else if (a.cout > 16) a.tiles_n = cdiv(a.cout, 32), launch_wgrad<RT, 2>(a, kvol, stream);
  else a.tiles_n = 1, launch_wgrad<RT, 1>(a, kvol, stream);
}

}  // namespace df3d

using namespace df3d;

extern "C" int df3d_invert_neighbors(const int32_t *nbr, int kvol, int n_out, int n_in, int32_t *inv, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(kvol > 0 && kvol <= DF3D_MAX_KVOL && n_out >= 0 && n_in >= 0, "invert_neighbors: bad sizes");
  if (n_in == 0) return DF3D_OK;
  DF3D_CHECK_ARG(inv, "invert_neighbors: null output");
  const size_t n = (size_t)kvol * n_in;
  hipLaunchKernelGGL(invert_fill_kernel, dim3(cdiv((long long)n, 256)), dim3(256), 0, stream, inv, n);
  if (n_out > 0) {
    DF3D_CHECK_ARG(nbr, "invert_neighbors: null table");
    hipLaunchKernelGGL(invert_neighbors_kernel, dim3(cdiv((long long)kvol * n_out, 256)), dim3(256), 0, stream, nbr, kvol,
                       n_out, n_in, inv);
  }
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_sparse_conv_grad_filters(const float *features, int n_in, int cin, const float *grad_out, int n_out,
                                             int cout, const int32_t *nbr, int kvol, float *grad_filters,
                                             void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(kvol > 0 && kvol <= DF3D_MAX_KVOL && cin > 0 && cout > 0 && n_in >= 0 && n_out >= 0,
                 "sparse_conv_grad_filters: bad sizes");
  DF3D_CHECK_ARG(grad_filters, "sparse_conv_grad_filters: null output");
  DF3D_HIP(hipMemsetAsync(grad_filters, 0, (size_t)kvol * cin * cout * sizeof(float), stream));
  if (n_out == 0 || n_in == 0) return DF3D_OK;
  DF3D_CHECK_ARG(features && grad_out && nbr, "sparse_conv_grad_filters: null argument");
  const char *env = getenv("DF3D_WGRAD");                    // read per call: tests switch between the kernels
  // 0 = grad_filters_kernel, 1 = wgrad_f32_kernel (exact fp32 products), 3 = wgrad_split3_kernel (three bf16 parts, six
  // products: fp32-grade); default: 3 where it measured faster (tools/ubench/wgrad3_probe.py, MI355X: 128 -> 128 K = 27 230 ->
  // 171 us, dense 3 x 3 128 -> 128 121 -> 112, 256 -> 256 118 -> 108, 256 -> 128 202 -> 179, the head's 64 -> 36 x 64 1257 -> 940;
  // slower on 64 -> 64 K = 27 116 -> 138, 512 -> 64 198 -> 214 and on maps under 16 k rows), else 1
  const bool wide = (cin >= 128 && cout >= 128 && n_out >= 16384) || cout >= 1024;
  const int kernel_choice = env ? atoi(env) : (wide ? 3 : 1);
  if (kernel_choice == 3 && cin % 4 == 0 && cout % 4 == 0) {
    WgradArgs a{features, grad_out, nbr, grad_filters, n_out, cin, cout, 0, 1, 0, 0, 0, {0}};
    int rc = launch_wgrad3_any(a, kvol, stream);
    if (rc) return rc;
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
  }
  if (kernel_choice && cin % 4 == 0 && cout % 4 == 0) {
    WgradArgs a{features, grad_out, nbr, grad_filters, n_out, cin, cout, 0, 1, 0, 0, 0, {0}};
    // enough workgroups to fill the chip a few times over, slices of at least 1024 rows (the atomics of a slice are
    // amortised over its pairs)
    const int tiles = cdiv(cin, cin >= 64 ? 64 : (cin > 16 ? 32 : 16)) * cdiv(cout, cout >= 64 ? 64 : (cout > 16 ? 32 : 16));
    const int want = std::max(1, cdiv(3072, kvol * tiles));
    const int slices = std::min(want, cdiv(n_out, 1024));
    a.slice = cdiv(cdiv(n_out, slices), 256) * 256;
    if (cin >= 64) launch_wgrad_ct<4>(a, kvol, stream);
    else if (cin > 16) launch_wgrad_ct<2>(a, kvol, stream);
    else launch_wgrad_ct<1>(a, kvol, stream);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
  }
  DF3D_CHECK_ARG(cout <= 128, "sparse_conv_grad_filters: at most 128 output channels (got %d) unless both channel counts "
                 "are multiples of 4", cout);
  const dim3 grid(cdiv(n_out, GF_ROWS), kvol, cdiv(cdiv(cin, 16), 4));
  const int ct = cdiv(cout, 16);
  if (ct <= 1) hipLaunchKernelGGL(grad_filters_kernel<1>, grid, dim3(256), 0, stream, features, grad_out, nbr, n_out, cin, cout, grad_filters);
  else if (ct <= 2) hipLaunchKernelGGL(grad_filters_kernel<2>, grid, dim3(256), 0, stream, features, grad_out, nbr, n_out, cin, cout, grad_filters);
  else if (ct <= 4) hipLaunchKernelGGL(grad_filters_kernel<4>, grid, dim3(256), 0, stream, features, grad_out, nbr, n_out, cin, cout, grad_filters);
  else hipLaunchKernelGGL(grad_filters_kernel<8>, grid, dim3(256), 0, stream, features, grad_out, nbr, n_out, cin, cout, grad_filters);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_rows_grad_weights(const float *x, const float *grad_out, long long n, int cin, int cout, float *grad_weights,
                                      void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(grad_weights && cin > 0 && cout > 0 && n >= 0 && n < (1ll << 31), "rows_grad_weights: bad sizes");
  DF3D_CHECK_ARG(cin % 4 == 0 && cout % 4 == 0, "rows_grad_weights: channel counts must be multiples of 4 (got %d, %d)", cin, cout);
  DF3D_HIP(hipMemsetAsync(grad_weights, 0, (size_t)cin * cout * sizeof(float), stream));
  if (n == 0) return DF3D_OK;
  DF3D_CHECK_ARG(x && grad_out, "rows_grad_weights: null argument");
  WgradArgs a{x, grad_out, nullptr, grad_weights, (int)n, cin, cout, 0, 1, 0, 0, 0, {0}};
  int rc = launch_wgrad3_any(a, 1, stream);
  if (rc) return rc;
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

// bf16 mixed-precision training (round 6; BASELINE configs[2] / [3] are bf16 configurations): both operands rounded to ONE bf16
// part where they are staged, one product, fp32 accumulate -- what an fp16-AMP reference does for its filter gradients
// (spconv_ops.h:363-456 under autocast), at a third of the two-part kernel's matrix work and half of its staging.
extern "C" int df3d_sparse_conv_grad_filters_bf16(const float *features, int n_in, int cin, const float *grad_out, int n_out,
                                                  int cout, const int32_t *nbr, int kvol, float *grad_filters, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (cin % 4 || cout % 4 || cin < 64 || cout < 64)             // narrow layers: the fp32 kernels (the 64-wide blocks would be half empty)
    return df3d_sparse_conv_grad_filters(features, n_in, cin, grad_out, n_out, cout, nbr, kvol, grad_filters, stream_);
  DF3D_CHECK_ARG(kvol > 0 && kvol <= DF3D_MAX_KVOL && n_in >= 0 && n_out >= 0, "sparse_conv_grad_filters_bf16: bad sizes");
  DF3D_CHECK_ARG(grad_filters, "sparse_conv_grad_filters_bf16: null output");
  DF3D_HIP(hipMemsetAsync(grad_filters, 0, (size_t)kvol * cin * cout * sizeof(float), stream));
  if (n_out == 0 || n_in == 0) return DF3D_OK;
  DF3D_CHECK_ARG(features && grad_out && nbr, "sparse_conv_grad_filters_bf16: null argument");
  WgradArgs a{features, grad_out, nbr, grad_filters, n_out, cin, cout, 0, 1, 0, 0, 0, {0}};
  int rc = launch_wgrad3_any(a, kvol, stream, 1);
  if (rc) return rc;
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

// Two-part forms (second half of round 5): fp16 pairs, three products.  grad_scale / x_scale / g_scale point at the power-of-two
// block scale of a GRADIENT operand (df3d_split_rows_scaled / df3d_rows_pow2_scale write it); NULL = an activation operand at
// the fixed scale 2^5 of every split kernel.
extern "C" int df3d_sparse_conv_grad_filters_scaled(const float *features, int n_in, int cin, const float *grad_out, int n_out,
                                                    int cout, const int32_t *nbr, int kvol, const float *grad_scale,
                                                    float *grad_filters, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const char *env = getenv("DF3D_WGRAD");                    // (a forced kernel choice: the unscaled entry decides)
  // measured (tools/ubench/wgrad3_probe.py, us: exact-fp32 kernel / three bf16 parts / this): conv4 128 -> 128 K = 27 226 / 167 /
  // 117, conv3 64 -> 64 116 / 147 / 96, dense 3 x 3 256 -> 128 202 / 175 / 127, 512 -> 64 202 / 208 / 146, the head's 64 -> 36 x 64
  // 1254 / 918 / 582; 32 -> 32 K = 27 40 / 149 / 107: the 64-wide blocks are half empty there -- the fp32 kernel keeps < 64 channels
  const bool fits = cin % 4 == 0 && cout % 4 == 0 && cin >= 64 && cout >= 64 && grad_scale;
  if (env || !fits)
    return df3d_sparse_conv_grad_filters(features, n_in, cin, grad_out, n_out, cout, nbr, kvol, grad_filters, stream_);
  DF3D_CHECK_ARG(kvol > 0 && kvol <= DF3D_MAX_KVOL && n_in >= 0 && n_out >= 0, "sparse_conv_grad_filters_scaled: bad sizes");
  DF3D_CHECK_ARG(grad_filters, "sparse_conv_grad_filters_scaled: null output");
  DF3D_HIP(hipMemsetAsync(grad_filters, 0, (size_t)kvol * cin * cout * sizeof(float), stream));
  if (n_out == 0 || n_in == 0) return DF3D_OK;
  DF3D_CHECK_ARG(features && grad_out && nbr, "sparse_conv_grad_filters_scaled: null argument");
  WgradArgs a{features, grad_out, nbr, grad_filters, n_out, cin, cout, 0, 1, 0, 0, 0, {0}};
  a.sa = nullptr, a.sg = grad_scale;
  int rc = launch_wgrad3_any(a, kvol, stream, 2);
  if (rc) return rc;
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_rows_grad_weights_scaled(const float *x, const float *grad_out, long long n, int cin, int cout,
                                             const float *x_scale, const float *g_scale, float *grad_weights, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(grad_weights && cin > 0 && cout > 0 && n >= 0 && n < (1ll << 31), "rows_grad_weights_scaled: bad sizes");
  DF3D_CHECK_ARG(cin % 4 == 0 && cout % 4 == 0, "rows_grad_weights_scaled: channel counts must be multiples of 4 (got %d, %d)", cin, cout);
  DF3D_HIP(hipMemsetAsync(grad_weights, 0, (size_t)cin * cout * sizeof(float), stream));
  if (n == 0) return DF3D_OK;
  DF3D_CHECK_ARG(x && grad_out, "rows_grad_weights_scaled: null argument");
  WgradArgs a{x, grad_out, nullptr, grad_weights, (int)n, cin, cout, 0, 1, 0, 0, 0, {0}};
  a.sa = x_scale, a.sg = g_scale;
  int rc = launch_wgrad3_any(a, 1, stream, 2);
  if (rc) return rc;
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}
