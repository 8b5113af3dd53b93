// Shared helpers for libdf3d_hip.so (gfx950 only: wave64, no portability shims).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/df3d_hip.h"

namespace df3d {

void set_error(const char *fmt, ...);

#define DF3D_CHECK_ARG(cond, ...)            \
  do {                                       \
    if (!(cond)) {                           \
      df3d::set_error(__VA_ARGS__);          \
      return DF3D_EINVAL;                    \
    }                                        \
  } while (0)

#define DF3D_HIP(call)                                                                   \
  do {                                                                                   \
    hipError_t e__ = (call);                                                             \
    if (e__ != hipSuccess) {                                                             \
      df3d::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, \
                      __LINE__);                                                         \
      return DF3D_EHIP;                                                                  \
    }                                                                                    \
  } while (0)

#define DF3D_LAUNCH_CHECK()                                                                  \
  do {                                                                                       \
    hipError_t e__ = hipGetLastError();                                                      \
    if (e__ != hipSuccess) {                                                                 \
      df3d::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, \
                      __LINE__);                                                             \
      return DF3D_EHIP;                                                                      \
    }                                                                                        \
  } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
#ifdef __HIPCC__
// hi/lo bf16 split of two fp32 values with the hardware converter (v_cvt_pk_bf16_f32, round to nearest even):
// hi = bf16(x), lo = bf16(x - hi); each result packs the pair (x0 in the low 16 bits).  5 VALU per pair.
typedef __bf16 df3d_bf16x2 __attribute__((ext_vector_type(2)));
typedef float df3d_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair_ref(float x0, float x1, unsigned &hi, unsigned &lo) {
  df3d_f32x2 v = {x0, x1};
  df3d_bf16x2 h = __builtin_convertvector(v, df3d_bf16x2);
  df3d_f32x2 r = v - __builtin_convertvector(h, df3d_f32x2);
  df3d_bf16x2 l = __builtin_convertvector(r, df3d_bf16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}
// outputs may be vector elements (which cannot bind to references)
#define split_pair(x0, x1, HI, LO)                 \
  do {                                             \
    unsigned sp_h__, sp_l__;                       \
    df3d::split_pair_ref(x0, x1, sp_h__, sp_l__);  \
    (HI) = sp_h__;                                 \
    (LO) = sp_l__;                                 \
  } while (0)
#endif

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Bump allocator over a caller-provided workspace.
struct Arena {
  char *base;
  size_t size, off;
  Arena(void *p, size_t n) : base((char *)p), size(n), off(0) {}
  template <typename T>
  T *take(size_t count) {
    size_t o = align_up(off, 256);
    size_t end = o + count * sizeof(T);
    if (end > size) return nullptr;
    off = end;
    return (T *)(base + o);
  }
};
static inline size_t arena_need(size_t cur, size_t bytes) { return align_up(cur, 256) + bytes; }

// ---- device-wide exclusive scan of uint32 (3 launches; n up to 2^31) -------------------
// scan_bytes(n): scratch needed.  out may alias in.  total (device u32*, may be NULL)
// receives the grand total.
size_t scan_scratch_bytes(size_t n);
int exclusive_scan_u32(const uint32_t *in, uint32_t *out, size_t n, uint32_t *total,
                       void *scratch, size_t scratch_bytes, hipStream_t stream);
// same, where the input element i is popcount(words[i])
int exclusive_scan_popc64(const unsigned long long *words, uint32_t *out, size_t n,
                          uint32_t *total, void *scratch, size_t scratch_bytes,
                          hipStream_t stream);

// ---- occupancy directory -------------------------------------------------------------
// Layout of a grid blob: [Header][bits: nwords u64][prefix: nwords u32][total u32][scan scratch]
struct GridHeader {
  int batch, shape[3];
  unsigned long long ncells;  // batch * vol
  unsigned long long nwords;  // ceil(ncells / 64)
  size_t off_bits, off_prefix, off_total, off_scratch, scratch_bytes;
};
struct GridView {
  const unsigned long long *bits;
  const uint32_t *prefix;
  int shape[3];
  long long vol;
};
GridHeader grid_layout(int batch, const int *shape);
static inline GridView grid_view(const void *blob, const GridHeader &h) {
  GridView v;
  v.bits = (const unsigned long long *)((const char *)blob + h.off_bits);
  v.prefix = (const uint32_t *)((const char *)blob + h.off_prefix);
  v.shape[0] = h.shape[0];
  v.shape[1] = h.shape[1];
  v.shape[2] = h.shape[2];
  v.vol = (long long)h.shape[0] * h.shape[1] * h.shape[2];
  return v;
}

__device__ __forceinline__ int grid_rank(const GridView &g, long long flat) {
  unsigned long long w = g.bits[flat >> 6];
  int b = (int)(flat & 63);
  if (!((w >> b) & 1ull)) return -1;
  return (int)(g.prefix[flat >> 6] + __popcll(w & ((1ull << b) - 1ull)));
}

// topk.hip: the K smallest 64-bit keys of each of S equally long segments, ascending (K <= 4096); out_count[s] (optional)
// = how many of them are not the all-ones key
size_t topk_keys_workspace(int S, long long n, int K);
int topk_keys(const unsigned long long *keys, int S, long long n, int K, unsigned long long *out, int32_t *out_count,
              void *ws, size_t ws_bytes, hipStream_t stream);

}  // namespace df3d
