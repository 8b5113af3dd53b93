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
  const char *env = getenv("DF3D_WGRAD");                    // read per call: tests switch between the two kernels
  const int kernel_choice = env ? atoi(env) : 1;
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
