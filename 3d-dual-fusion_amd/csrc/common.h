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
// ---- operand splits of the matrix-core kernels ------------------------------------------------------------------------
// Round 5: the two-part split is an fp16 split.  An fp32 value x, scaled by a fixed power of two S, is stored as
//     hi = fp16(S x),  lo = fp16(S x - hi)          (S x = hi + lo up to 2^-22 |S x|: 11 + 11 significand bits)
// and a contraction is evaluated as A_lo W_hi + A_hi W_lo + A_hi W_hi on v_mfma_f32_16x16x32_f16 (fp16 x fp16 products are
// exact in fp32, accumulation is fp32), then multiplied by 2^-(SA + SW) -- exact.  Same bytes, same three matrix products
// per operand pair as the bf16 hi/lo split of rounds 1-4 (which kept 8 + 8 bits: ~5e-6 of the output scale against
// float64), at the accuracy of the exact-fp32 kernels (~1e-6 of scale = fp32 accumulation noise;
// tools/ubench/f16split_probe.hip).  What fp16 lacks is exponent range: activations are scaled by 2^5 (|x| < 2047;
// full 22 bits down to |x| = 2^-8, below that an absolute error <= 2^-30), weights by 2^7 (|w| < 511; 2^-32).  A value
// outside the range does NOT pass silently: every split that writes memory checks it and raises a sticky flag that
// `df3d_split_overflow()` reads (the Python layer raises on it); `DF3D_CONV_PRECISION=split3` (three bf16 parts, fp32's
// exponent range, twice the matrix work) is the mode for such data.  MFMA and the converters honour fp16 subnormals.
#define DF3D_SA_EXP 5
#define DF3D_SW_EXP 7
#define DF3D_SA_SCALE 32.0f                        // 2^SA: activations
#define DF3D_SW_SCALE 128.0f                       // 2^SW: weights
#define DF3D_SA_INV 0.03125f
#define DF3D_ACC_UNSCALE 0.000244140625f           // 2^-(SA + SW): accumulators of a two-part product -> fp32 values
#define DF3D_AA_UNSCALE 0.0009765625f              // 2^-(2 SA): products of two activation operands
typedef __bf16 df3d_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 df3d_f16x2 __attribute__((ext_vector_type(2)));
typedef float df3d_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 df3d_f16x8 __attribute__((ext_vector_type(8)));

static __device__ unsigned g_split_overflow_tu;    // one per translation unit (no relocatable device code): see split_overflow_*

// dropout keep decision of element i under a call's 64-bit seed (s0 = low, s1 = high word): a counter-based hash, the same mask
// whatever the launch shape; thr = p * 2^24 against 24 uniform bits (df3d_relu_dropout, df3d_dropout_add_layernorm,
// df3d_cross_attention_train)
__device__ __forceinline__ unsigned rd_mix(unsigned x) {
  x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
  return x;
}
__device__ __forceinline__ bool rd_keep(unsigned long long i, unsigned s0, unsigned s1, unsigned thr) {
  const unsigned h = rd_mix(rd_mix((unsigned)i ^ s0) + (unsigned)(i >> 32) * 0x9E3779B9u + s1);
  return (h >> 8) >= thr;
}

// a pair of fp32 values -> packed bf16 (round to nearest even, x0 in the low 16 bits): the operand format of the bf16 mode
__device__ __forceinline__ unsigned bf16_pair(float x0, float x1) {
  df3d_f32x2 v = {x0, x1};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, df3d_bf16x2));
}
// hi/lo bf16 split of two fp32 values (v_cvt_pk_bf16_f32): the building block of the three-part split
__device__ __forceinline__ void split_pair_bf16_ref(float x0, float x1, unsigned &hi, unsigned &lo) {
  df3d_f32x2 v = {x0, x1};
  df3d_bf16x2 h = __builtin_convertvector(v, df3d_bf16x2);
  df3d_f32x2 r = v - __builtin_convertvector(h, df3d_f32x2);
  df3d_bf16x2 l = __builtin_convertvector(r, df3d_bf16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}
// hi/lo fp16 split of two fp32 values scaled by `s` (a power of two); each result packs the pair (x0 in the low 16 bits)
template <bool CHECK>
__device__ __forceinline__ void split_pair_f16_ref(float x0, float x1, float s, unsigned &hi, unsigned &lo) {
  df3d_f32x2 v = {x0 * s, x1 * s};
  if (CHECK) {
    // (also true for NaN / inf that an unchecked split upstream let through)
    if (__builtin_expect(!(__builtin_fabsf(v[0]) <= 65504.f) || !(__builtin_fabsf(v[1]) <= 65504.f), 0))
      atomicOr(&g_split_overflow_tu, 1u);
  }
  df3d_f16x2 h = __builtin_convertvector(v, df3d_f16x2);
  df3d_f32x2 r = v - __builtin_convertvector(h, df3d_f32x2);
  df3d_f16x2 l = __builtin_convertvector(r, df3d_f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}
// outputs may be vector elements (which cannot bind to references)
#define DF3D_SPLIT_PAIR_(x0, x1, S, CHECK, HI, LO)                          \
  do {                                                                      \
    unsigned sp_h__, sp_l__;                                                \
    df3d::split_pair_f16_ref<CHECK>(x0, x1, S, sp_h__, sp_l__);             \
    (HI) = sp_h__;                                                          \
    (LO) = sp_l__;                                                          \
  } while (0)
// activations (checked: for splits that go to memory; _nc: in-register operands inside a matrix loop, whose out-of-range
// values turn into inf / NaN and are caught by the next checked split downstream)
#define split_pair(x0, x1, HI, LO) DF3D_SPLIT_PAIR_(x0, x1, DF3D_SA_SCALE, true, HI, LO)
#define split_pair_nc(x0, x1, HI, LO) DF3D_SPLIT_PAIR_(x0, x1, DF3D_SA_SCALE, false, HI, LO)
// checked without a branch per pair: the caller keeps the largest |value| it has split (AMAX, one v_max3 per pair; the values
// are finite fp32 numbers -- NaN can only follow an overflow that was not flagged) and calls split_range_flag(AMAX) once
#define split_pair_acc(x0, x1, HI, LO, AMAX)                                                  \
  do {                                                                                        \
    (AMAX) = fmaxf((AMAX), fmaxf(__builtin_fabsf(x0), __builtin_fabsf(x1)));                  \
    DF3D_SPLIT_PAIR_(x0, x1, DF3D_SA_SCALE, false, HI, LO);                                   \
  } while (0)
#define split_range_flag(AMAX)                                                                \
  do {                                                                                        \
    if (__builtin_expect(!((AMAX) * DF3D_SA_SCALE <= 65504.f), 0)) atomicOr(&df3d::g_split_overflow_tu, 1u); \
  } while (0)
// (AMAX already carries the operand's scale)
#define split_range_flag_scaled(AMAX)                                                         \
  do {                                                                                        \
    if (__builtin_expect(!((AMAX) <= 65504.f), 0)) atomicOr(&df3d::g_split_overflow_tu, 1u);  \
  } while (0)
// weights / filters (packed once per parameter version)
#define split_pair_w(x0, x1, HI, LO) DF3D_SPLIT_PAIR_(x0, x1, DF3D_SW_SCALE, true, HI, LO)
#define split_pair_bf16(x0, x1, HI, LO)                  \
  do {                                                   \
    unsigned sp_h__, sp_l__;                             \
    df3d::split_pair_bf16_ref(x0, x1, sp_h__, sp_l__);   \
    (HI) = sp_h__;                                       \
    (LO) = sp_l__;                                       \
  } while (0)
// one fp32 value -> the 16-bit patterns of its activation split
__device__ __forceinline__ void split_one(float x, unsigned &hi, unsigned &lo) {
  unsigned h, l;
  split_pair_f16_ref<true>(x, 0.f, DF3D_SA_SCALE, h, l);
  hi = h & 0xffffu;
  lo = l & 0xffffu;
}
// packed fp16 pair of a split row -> the two fp32 values of that PART (hi or lo), unscaled
__device__ __forceinline__ df3d_f32x2 split_part_to_f32(unsigned packed) {
  df3d_f32x2 v = __builtin_convertvector(__builtin_bit_cast(df3d_f16x2, packed), df3d_f32x2);
  return v * DF3D_SA_INV;
}
// hi + lo of a packed pair -> fp32 values (exact: both parts and their sum are fp32 numbers)
__device__ __forceinline__ df3d_f32x2 split_to_f32(unsigned hi, unsigned lo) {
  df3d_f32x2 h = __builtin_convertvector(__builtin_bit_cast(df3d_f16x2, hi), df3d_f32x2);
  df3d_f32x2 l = __builtin_convertvector(__builtin_bit_cast(df3d_f16x2, lo), df3d_f32x2);
  return (h + l) * DF3D_SA_INV;
}
#define DF3D_MFMA_F16(A, B, C) \
  __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(df3d_f16x8, A), __builtin_bit_cast(df3d_f16x8, B), C, 0, 0, 0)
#endif

// sticky overflow flags of the fp16 operand splits: every translation unit with device code that splits registers the
// address of its flag (common.hip keeps the list); see df3d_split_overflow()
void split_overflow_register(const void *symbol, const char *tu);
#ifdef __HIPCC__
#define DF3D_SPLIT_OVERFLOW_TU(name)                                                                       \
  namespace {                                                                                              \
  struct SplitOverflowReg_##name {                                                                         \
    SplitOverflowReg_##name() { df3d::split_overflow_register((const void *)&df3d::g_split_overflow_tu, #name); } \
  } g_split_overflow_reg_##name;                                                                           \
  }
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
