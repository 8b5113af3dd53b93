// The K smallest 64-bit keys of every segment, in ascending order -- the selection step of both detection heads.
//
// Reference: CenterHead.post_processing sorts the masked scores of a (task, sample) and keeps the first
// nms_pre_max_size (CP/det3d/models/bbox_heads/center_head.py:470-478 -> box_torch_ops.rotate_nms_pcdet :248-279);
// TransFusionHead argsorts all C*H*W scores of a sample for its first num_proposals entries
// (TF/mmdet3d/models/dense_heads/transfusion_head.py:866).  Both heads here encode (score, index) in one key
// [segment 8 | 0x3F800000 - score bits 32 | index 24], so "descending score, ties by ascending index" = ascending key.
//
// A full device sort of 194 k / 324 k keys per sample (rocPRIM: block sort + 9 merge passes, ~120 us) is replaced by a
// two-level radix SELECT on the 24 leading bits of the 54 significant key bits, four launches over the keys (the all-ones
// key means "masked" -- below the score threshold / outside the range: it is never a candidate, the output is padded
// with it):
//   topk_hist1     4096-bin histogram of bits [53:42] per segment (LDS-privatised)
//   topk_hist2     every workgroup re-derives the level-1 threshold bin from hist1, histograms bits [41:30] inside it
//   topk_compact   re-derives both thresholds; keys whose 24-bit prefix is BELOW the threshold (fewer than K) go to
//                  region 0, keys that TIE with it to region 1 (bounded; the total is still counted)
//   topk_final     one workgroup per segment: bitonic sort of region 0 + region 1 in LDS, first K written out.  If more
//                  keys tie on the 24-bit prefix than region 1 holds (near-constant, saturated or all-zero score maps:
//                  thousands of equal scores, only the index decides), it continues the radix select by itself on
//                  key bits [29:18], [17:6], [5:0] (one pass over the segment per level) -- exact for any input.
#include <cstring>

#include "common.h"

namespace df3d {

typedef unsigned long long u64;

constexpr int TK_BINS = 4096;
constexpr int TK_TIE_CAP = 4096;     // region 1 capacity per segment
constexpr int TK_BUF = 8192;         // LDS sort buffer (keys)
constexpr int TK_KEYS_PER_BLOCK = 256 * 16;

__device__ __forceinline__ unsigned tk_prefix24(u64 key) {
  const u64 p = (key & 0x00FFFFFFFFFFFFFFull) >> 30;
  return p > 0xFFFFFFull ? 0xFFFFFFu : (unsigned)p;
}

// smallest bin T whose inclusive cumulative count reaches `need` (T = 4095 if none does); A = count below T.
// 256 threads, result through shared memory (every thread returns the same values).
__device__ void tk_find_threshold(const unsigned *hist, unsigned need, unsigned &T, unsigned &A) {
  __shared__ unsigned s_part[256], s_T, s_A;
  const int t = threadIdx.x;
  const bool act = t < 256;                        // wider workgroups: the other threads only join the barriers
  unsigned mine[16], sum = 0;
  if (act) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      mine[e] = hist[t * 16 + e];
      sum += mine[e];
    }
  }
  __syncthreads();                                 // protects s_part / s_T across repeated calls
  if (act) s_part[t] = sum;
  if (t == 0) {
    s_T = 0xFFFFFFFFu;
    s_A = 0;
  }
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {        // inclusive Hillis-Steele scan
    const unsigned v = (act && t >= off) ? s_part[t - off] : 0;
    __syncthreads();
    if (act) s_part[t] += v;
    __syncthreads();
  }
  const unsigned incl = act ? s_part[t] : 0, excl = incl - sum;
  if (act && excl < need && incl >= need) {        // exactly one thread
    unsigned c = excl;
    for (int e = 0; e < 16; ++e) {
      if (c + mine[e] >= need) {
        s_T = t * 16 + e;
        s_A = c;
        break;
      }
      c += mine[e];
    }
  }
  __syncthreads();
  // every thread snapshots the flag BEFORE anyone may overwrite it (a wave that read it after thread 255's store
  // would skip the barrier below and leave the workgroup's barriers skewed)
  const bool none = (s_T == 0xFFFFFFFFu);          // fewer than `need` keys in total
  __syncthreads();
  if (none && t == 255) {
    s_T = TK_BINS - 1;
    s_A = incl - mine[15];
  }
  __syncthreads();
  T = s_T;
  A = s_A;
}

// level = 0: bits [23:12] of the prefix of every key; level = 1: bits [11:0] of the keys inside the level-1 threshold bin
template <int LEVEL>
__global__ __launch_bounds__(256) void topk_hist_kernel(const u64 *__restrict__ keys, long long n, int K,
                                                        unsigned *__restrict__ hist1, unsigned *__restrict__ hist2) {
  const int s = blockIdx.y;
  __shared__ unsigned s_hist[TK_BINS];
  for (int e = threadIdx.x; e < TK_BINS; e += 256) s_hist[e] = 0;
  unsigned T1 = 0, A1 = 0;
  if (LEVEL == 1) tk_find_threshold(hist1 + (size_t)s * TK_BINS, (unsigned)K, T1, A1);
  __syncthreads();
  const u64 *seg = keys + (size_t)s * n;
  const long long base = (long long)blockIdx.x * TK_KEYS_PER_BLOCK;
#pragma unroll 4
  for (int e = 0; e < 16; ++e) {
    const long long i = base + e * 256 + threadIdx.x;
    if (i < n) {
      const u64 key = seg[i];
      const unsigned p = tk_prefix24(key);
      if (key == ~0ull) continue;                  // masked entry: not a candidate
      if (LEVEL == 0) atomicAdd(&s_hist[p >> 12], 1u);
      else if ((p >> 12) == T1) atomicAdd(&s_hist[p & 0xFFFu], 1u);
    }
  }
  __syncthreads();
  unsigned *out = (LEVEL == 0 ? hist1 : hist2) + (size_t)s * TK_BINS;
  for (int e = threadIdx.x; e < TK_BINS; e += 256)
    if (s_hist[e]) atomicAdd(&out[e], s_hist[e]);
}

__global__ __launch_bounds__(256) void topk_compact_kernel(const u64 *__restrict__ keys, long long n, int K,
                                                           const unsigned *__restrict__ hist1,
                                                           const unsigned *__restrict__ hist2, unsigned *__restrict__ cnt,
                                                           unsigned *__restrict__ thr, u64 *__restrict__ region0,
                                                           u64 *__restrict__ region1) {
  const int s = blockIdx.y;
  unsigned T1, A1, T2, A2;
  tk_find_threshold(hist1 + (size_t)s * TK_BINS, (unsigned)K, T1, A1);
  tk_find_threshold(hist2 + (size_t)s * TK_BINS, (unsigned)K - A1, T2, A2);
  const unsigned thr24 = (T1 << 12) | T2;
  if (blockIdx.x == 0 && threadIdx.x == 0) thr[s] = thr24;
  const u64 *seg = keys + (size_t)s * n;
  const long long base = (long long)blockIdx.x * TK_KEYS_PER_BLOCK;
  // the workgroup's keys stay in registers: count per region, ONE global atomic per workgroup and region, then write
  u64 key[16];
  unsigned mine[2] = {0, 0};
  unsigned short sel = 0, tie = 0;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const long long i = base + e * 256 + threadIdx.x;
    key[e] = i < n ? seg[i] : ~0ull;
    const unsigned p = tk_prefix24(key[e]);
    if (key[e] != ~0ull && p <= thr24) {
      if (p < thr24) {
        sel |= 1u << e;
        ++mine[0];
      } else {
        tie |= 1u << e;
        ++mine[1];
      }
    }
  }
  __shared__ unsigned s_wave[2][4], s_base[2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned pre[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {                    // inclusive wave scan
    unsigned v = mine[r];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned u = __shfl_up(v, off);
      if (lane >= off) v += u;
    }
    pre[r] = v - mine[r];
    if (lane == 63) s_wave[r][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const int r = threadIdx.x;
    const unsigned tot = s_wave[r][0] + s_wave[r][1] + s_wave[r][2] + s_wave[r][3];
    s_base[r] = tot ? atomicAdd(&cnt[s * 2 + r], tot) : 0;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    unsigned pos = s_base[r] + pre[r];
    for (int w = 0; w < wave; ++w) pos += s_wave[r][w];
    const unsigned short bits = r == 0 ? sel : tie;
#pragma unroll
    for (int e = 0; e < 16; ++e)
      if ((bits >> e) & 1u) {
        if (r == 0) region0[(size_t)s * K + pos] = key[e];                      // fewer than K keys lie below the threshold
        else if (pos < TK_TIE_CAP) region1[(size_t)s * TK_TIE_CAP + pos] = key[e];
        ++pos;
      }
  }
}

// ascending bitonic sort of buf[0, N), N a power of two <= TK_BUF, blockDim.x = 1024
__device__ void tk_bitonic(u64 *buf, int N) {
  for (int k = 2; k <= N; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < (N >> 1); t += 1024) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
        const u64 a = buf[i], b = buf[l];
        const bool up = (i & k) == 0;
        if ((a > b) == up) {
          buf[i] = b;
          buf[l] = a;
        }
      }
    }
  __syncthreads();
}

__device__ __forceinline__ int tk_pow2(int v) {
  int n = 2;
  while (n < v) n <<= 1;
  return n;
}

__global__ __launch_bounds__(1024) void topk_final_kernel(const u64 *__restrict__ keys, long long n, int K,
                                                          const unsigned *__restrict__ cnt, const unsigned *__restrict__ thr,
                                                          const u64 *__restrict__ region0, const u64 *__restrict__ region1,
                                                          u64 *__restrict__ out, int32_t *__restrict__ out_count) {
  extern __shared__ u64 buf[];                     // TK_BUF keys + TK_BINS counters
  __shared__ int s_cnt;
  const int s = blockIdx.x, t = threadIdx.x;
  const int A = (int)cnt[s * 2], P = (int)cnt[s * 2 + 1];
  int M;
  if (P <= TK_TIE_CAP) {
    M = A + P;
    for (int e = t; e < A; e += 1024) buf[e] = region0[(size_t)s * K + e];
    for (int e = t; e < P; e += 1024) buf[A + e] = region1[(size_t)s * TK_TIE_CAP + e];
    const int N = tk_pow2(M);
    for (int e = M + t; e < N; e += 1024) buf[e] = ~0ull;
    tk_bitonic(buf, N);
  } else {
    // more ties on the 24-bit prefix than region 1 holds (near-constant or saturated score maps, where thousands of
    // scores are equal and only the index decides): continue the radix select inside this workgroup on key bits
    // [29:18], [17:6], [5:0] -- one histogram pass over the segment per level -- until the candidates fit the buffer
    const int R = K - A;                           // >= 1 keys are still needed from the tie group
    const unsigned thr24 = thr[s];
    const u64 *seg = keys + (size_t)s * n;
    unsigned *hist3 = (unsigned *)(buf + TK_BUF);  // TK_BINS counters behind the sort buffer
    const u64 low56 = 0x00FFFFFFFFFFFFFFull;
    u64 chain = (u64)thr24 << 30;                  // threshold prefix found so far (bits [55:sh_prev])
    int sh_prev = 30;
    unsigned below = 0;                            // tie-group keys already known to be selected
    bool fits = false;
    int sh = 18;
    for (int level = 0; level < 3 && !fits; ++level) {
      sh = level == 0 ? 18 : (level == 1 ? 6 : 0);
      const unsigned nbits = sh_prev - sh, bmask = (1u << nbits) - 1u;
      for (int e = t; e < TK_BINS; e += 1024) hist3[e] = 0;
      __syncthreads();
      for (long long i0 = 0; i0 < n; i0 += 1024) {  // whole waves enter: the ballots below need every lane
        const long long i = i0 + t;
        const u64 key = i < n ? seg[i] : ~0ull;
        const bool in = key != ~0ull && ((key & low56) >> sh_prev) == (chain >> sh_prev);
        const unsigned bin = (unsigned)(key >> sh) & bmask;
        const u64 m = __ballot(in);
        if (m == 0) continue;
        const int leader = __ffsll((long long)m) - 1;
        const unsigned b0 = __shfl(bin, leader);
        if (__ballot(in && bin == b0) == m) {       // equal scores: the whole wave hits one counter -- one atomic
          if ((threadIdx.x & 63) == leader) atomicAdd(&hist3[b0], (unsigned)__popcll(m));
        } else if (in) {
          atomicAdd(&hist3[bin], 1u);
        }
      }
      __syncthreads();
      unsigned T3, A3;
      tk_find_threshold(hist3, (unsigned)R - below, T3, A3);
      const unsigned P3 = hist3[T3];
      __syncthreads();
      chain |= (u64)T3 << sh;
      below += A3;
      sh_prev = sh;
      fits = (unsigned)A + below + P3 <= (unsigned)TK_BUF;
    }
    if (fits) {
      if (t == 0) s_cnt = A;
      for (int e = t; e < A; e += 1024) buf[e] = region0[(size_t)s * K + e];
      __syncthreads();
      const u64 lo = ((u64)thr24 << 30) >> sh, hi = chain >> sh;
      for (long long i0 = 0; i0 < n; i0 += 1024) {
        const long long i = i0 + t;
        const u64 key = i < n ? seg[i] : ~0ull;
        const u64 v = (key & low56) >> sh;
        const bool in = key != ~0ull && v >= lo && v <= hi && tk_prefix24(key) == thr24;
        const u64 m = __ballot(in);
        if (m == 0) continue;
        const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
        int start = 0;
        if (lane == leader) start = atomicAdd(&s_cnt, __popcll(m));
        start = __shfl(start, leader);
        if (in) buf[start + __popcll(m & ((1ull << lane) - 1ull))] = key;
      }
      __syncthreads();
      M = s_cnt;                                   // >= K
      const int N = tk_pow2(M);
      for (int e = M + t; e < N; e += 1024) buf[e] = ~0ull;
      tk_bitonic(buf, N);
    } else {
      // more than a buffer of IDENTICAL keys (not produced by the heads): stream, keeping the R best of the tie group
      for (int e = t; e < TK_BUF; e += 1024) buf[e] = ~0ull;
      if (t == 0) s_cnt = R;
      __syncthreads();
      for (long long base = 0; base < n; base += 1024) {
        if (s_cnt + 1024 > TK_BUF) {               // uniform: s_cnt is read after a barrier
          tk_bitonic(buf, TK_BUF);
          for (int e = R + t; e < TK_BUF; e += 1024) buf[e] = ~0ull;
          if (t == 0) s_cnt = R;
          __syncthreads();
        }
        const long long i = base + t;
        if (i < n) {
          const u64 key = seg[i];
          if (key != ~0ull && tk_prefix24(key) == thr24) buf[atomicAdd(&s_cnt, 1)] = key;
        }
        __syncthreads();
      }
      tk_bitonic(buf, TK_BUF);                     // buf[0, R) = the R smallest tie keys
      __syncthreads();
      for (int e = t; e < A; e += 1024) buf[R + e] = region0[(size_t)s * K + e];
      M = K;
      const int N = tk_pow2(M);
      for (int e = M + t; e < N; e += 1024) buf[e] = ~0ull;
      tk_bitonic(buf, N);
    }
  }
  for (int e = t; e < K; e += 1024) out[(size_t)s * K + e] = e < M ? buf[e] : ~0ull;
  if (out_count && t == 0) {
    int lo = 0, hi = min(K, M);                    // number of keys below the all-ones key
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (buf[mid] != ~0ull) lo = mid + 1;
      else hi = mid;
    }
    out_count[s] = lo;
  }
}

struct TopkWs {
  size_t hist1, hist2, cnt, thr, zero_bytes, region0, region1, total;
};

static void topk_layout(int S, int K, TopkWs &w) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = align_up(off, 256);
    off = o + bytes;
    return o;
  };
  w.hist1 = take((size_t)S * TK_BINS * 4);
  w.hist2 = take((size_t)S * TK_BINS * 4);
  w.cnt = take((size_t)S * 2 * 4);
  w.thr = take((size_t)S * 4);
  w.zero_bytes = off;                              // everything up to here is cleared per call
  w.region0 = take((size_t)S * K * 8);
  w.region1 = take((size_t)S * TK_TIE_CAP * 8);
  w.total = align_up(off, 256);
}

size_t topk_keys_workspace(int S, long long n, int K) {
  if (S <= 0 || n <= 0 || K <= 0 || K > TK_TIE_CAP) return 0;
  TopkWs w;
  topk_layout(S, K, w);
  return w.total;
}

int topk_keys(const u64 *keys, int S, long long n, int K, u64 *out, int32_t *out_count, void *ws_, size_t ws_bytes,
              hipStream_t stream) {
  DF3D_CHECK_ARG(keys && out && ws_, "topk_keys: null argument");
  DF3D_CHECK_ARG(S > 0 && S <= 65535 && n > 0 && K > 0 && K <= TK_TIE_CAP, "topk_keys: need 1..65535 segments, n > 0, 0 < K <= %d",
                 TK_TIE_CAP);
  TopkWs w;
  topk_layout(S, K, w);
  DF3D_CHECK_ARG(ws_bytes >= w.total, "topk_keys: workspace %zu < %zu bytes", ws_bytes, w.total);
  static bool configured = false;
  const size_t lds = (size_t)TK_BUF * 8 + (size_t)TK_BINS * 4;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute((const void *)topk_final_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS) failed: %s", hipGetErrorString(e));
      return DF3D_EHIP;
    }
    configured = true;
  }
  char *ws = (char *)ws_;
  unsigned *hist1 = (unsigned *)(ws + w.hist1), *hist2 = (unsigned *)(ws + w.hist2), *cnt = (unsigned *)(ws + w.cnt),
           *thr = (unsigned *)(ws + w.thr);
  u64 *r0 = (u64 *)(ws + w.region0), *r1 = (u64 *)(ws + w.region1);
  DF3D_HIP(hipMemsetAsync(ws, 0, w.zero_bytes, stream));
  const dim3 grid(cdiv(n, TK_KEYS_PER_BLOCK), S);
  hipLaunchKernelGGL(topk_hist_kernel<0>, grid, dim3(256), 0, stream, keys, n, K, hist1, hist2);
  hipLaunchKernelGGL(topk_hist_kernel<1>, grid, dim3(256), 0, stream, keys, n, K, hist1, hist2);
  hipLaunchKernelGGL(topk_compact_kernel, grid, dim3(256), 0, stream, keys, n, K, hist1, hist2, cnt, thr, r0, r1);
  hipLaunchKernelGGL(topk_final_kernel, dim3(S), dim3(1024), lds, stream, keys, n, K, cnt, thr, r0, r1, out, out_count);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

}  // namespace df3d

using namespace df3d;

extern "C" size_t df3d_topk_keys_workspace_bytes(int segments, long long n, int k) { return topk_keys_workspace(segments, n, k); }

extern "C" int df3d_topk_keys(const unsigned long long *keys, int segments, long long n, int k, unsigned long long *out,
                              int32_t *out_count, void *workspace, size_t workspace_bytes, void *stream) {
  return topk_keys(keys, segments, n, k, out, out_count, workspace, workspace_bytes, (hipStream_t)stream);
}
