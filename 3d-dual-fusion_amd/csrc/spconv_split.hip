// Sparse convolution on the 16-bit matrix cores with fp32-grade accuracy ("split" path) for gfx950.
//
// fp32 MFMA on gfx950 runs at 1/16 of the fp16 / bf16 rate (157 vs 2500 TFLOP/s).  An fp32 value x is split
// exactly once, where it is PRODUCED, into two fp16 numbers of the value scaled by a fixed power of two
//     hi = fp16(S x),  lo = fp16(S x - hi)            (S x = hi + lo up to 2^-22 |S x|; csrc/common.h)
// and the contraction is evaluated as  A_lo*W_hi + A_hi*W_lo + A_hi*W_hi  with v_mfma_f32_16x16x32_f16
// (fp16 x fp16 products are exact in fp32; accumulation is fp32; the epilogue multiplies by 2^-(SA + SW)).  The
// dropped lo*lo term and the split residuals are <= 2^-21 relative per product: the result sits within fp32
// accumulation noise of the exact fp32 contraction (~1e-6 of the output scale against float64, like the exact-fp32
// MFMA kernels of spconv.hip; rounds 1-4 split into bf16 hi + lo: 16 bits, ~5e-6) at 3/16 of the fp32-MFMA time.
// That moves the C >= 64 layers from the MFMA roof onto the gather (L1/L2/HBM) roof, which is where
// BASELINE.json's north_star measures them.  Values outside fp16's range raise the flag df3d_split_overflow() reads.
//
// Data formats (both produced by kernels in this file or by the conv epilogue):
//   split rows   [n][C/8][ hi 8 x fp16 | lo 8 x fp16 ]   32 B per 8 channels, same bytes as fp32
//   packed W     [K][wave 4][kb][ct][hi|lo][lane 64][8 x fp16]: exactly the B operand of every lane, so the
//                per-offset weight fetch is 16 B per lane, coalesced
// (NP = 3: three bf16 parts per value, six products, fp32's exponent range; NP = 1: plain bf16 rows and filters)
//
// Kernel structure = spconv_pair_kernel (spconv.hip): one workgroup per CU, rulebook pairs of a row tile
// compacted per kernel offset, 16-pair MFMA chunks, per-wave column slice of W in registers, LDS
// accumulators, fused epilogue.  Differences: 16x16x32 bf16 MFMAs (3 per fp32 product block), gathers run
// 3 items ahead (items are ~5x shorter than in the fp32 kernel), weights rotate between two register sets by
// name, the item list is built by 32 lanes instead of one, and the epilogue can emit the split rows of its
// output for the next convolution.
#include <stdlib.h>
#include <algorithm>
#include <string.h>

#include "common.h"

namespace df3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

int timing_rec_begin(int cin, int cout, int kvol, int n_out, const int32_t *nbr, int split,
                     hipStream_t stream);   // spconv.hip
void timing_rec_end(int rec, hipStream_t stream);

DF3D_SPLIT_OVERFLOW_TU(spconv_split)

// three-way split (round 4, "split3" precision): x = hi + mid + lo EXACTLY for finite normal x (8 + 8 + 8 significand bits),
// every part a bf16 number; pairs packed like split_pair
__device__ __forceinline__ void split3_pair_ref(float x0, float x1, unsigned &hi, unsigned &mid, unsigned &lo) {
  unsigned h, m;
  split_pair_bf16(x0, x1, h, m);                                   // h = bf16(x), m = bf16(x - h)
  const float r0 = (x0 - __uint_as_float(h << 16)) - __uint_as_float(m << 16);
  const float r1 = (x1 - __uint_as_float(h & 0xffff0000u)) - __uint_as_float(m & 0xffff0000u);
  unsigned l, unused;
  split_pair_bf16(r0, r1, l, unused);
  hi = h, mid = m, lo = l;
}

// (outputs may be vector elements, which cannot bind to references)
#define split3_pair(x0, x1, HI, MID, LO)                         \
  do {                                                           \
    unsigned s3_h__, s3_m__, s3_l__;                             \
    df3d::split3_pair_ref(x0, x1, s3_h__, s3_m__, s3_l__);       \
    (HI) = s3_h__;                                               \
    (MID) = s3_m__;                                              \
    (LO) = s3_l__;                                               \
  } while (0)

__device__ __forceinline__ float bf16lo_to_f32(unsigned packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf16hi_to_f32(unsigned packed) { return __uint_as_float(packed & 0xffff0000u); }

// ---- fp32 rows -> bf16 rows (round to nearest even; the `hi` half of the split) --------------------------
__global__ __launch_bounds__(256) void bf16_rows_kernel(const float *__restrict__ x, size_t nblk, u32x4 *__restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nblk) return;
  f32x4 a = ((const f32x4 *)x)[2 * i], b = ((const f32x4 *)x)[2 * i + 1];
  out[i] = (u32x4){bf16_pair(a[0], a[1]), bf16_pair(a[2], a[3]), bf16_pair(b[0], b[1]), bf16_pair(b[2], b[3])};
}

// bf16 rows -> fp32 rows
__global__ __launch_bounds__(256) void bf16_rows_to_f32_kernel(const u32x4 *__restrict__ x, size_t nblk, float *__restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nblk) return;
  const u32x4 v = x[i];
  ((f32x4 *)out)[2 * i] = (f32x4){bf16lo_to_f32(v[0]), bf16hi_to_f32(v[0]), bf16lo_to_f32(v[1]), bf16hi_to_f32(v[1])};
  ((f32x4 *)out)[2 * i + 1] = (f32x4){bf16lo_to_f32(v[2]), bf16hi_to_f32(v[2]), bf16lo_to_f32(v[3]), bf16hi_to_f32(v[3])};
}

// W[K][CIN][COUT] fp32 -> bf16 B operands of the output-stationary kernel (NP = 1): [k][kb][ct][lane], same lane ->
// (column, channel) map as layout 1 of pack_weights_kernel, hi parts only
__global__ __launch_bounds__(256) void pack_weights_bf16_kernel(const float *__restrict__ w, int K, int cin, int cout,
                                                                u32x4 *__restrict__ out) {
  const int KB = cin / 32;
  size_t total = (size_t)K * cin * cout / 8;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int lane = (int)(i & 63);
  size_t r = i >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int CW = cout > 128 ? 128 : cout, CT = CW / 16;
  const int ct = (int)(r % CT); r /= CT;
  const int kb = (int)(r % KB); r /= KB;
  const int k = (int)(r % K);
  const int col = (int)(r / K) * CW + n * CT + ct;
  const int ch0 = kb * 32 + g * 8;
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    o[e] = bf16_pair(w[((size_t)k * cin + ch0 + 2 * e) * cout + col], w[((size_t)k * cin + ch0 + 2 * e + 1) * cout + col]);
  out[i] = o;
}

// ---- fp32 rows -> split rows ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_rows_kernel(const float *__restrict__ x, size_t nblk,
                                                         u32x4 *__restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nblk) return;
  f32x4 a = ((const f32x4 *)x)[2 * i], b = ((const f32x4 *)x)[2 * i + 1];
  u32x4 ho, lo;
  split_pair(a[0], a[1], ho[0], lo[0]);
  split_pair(a[2], a[3], ho[1], lo[1]);
  split_pair(b[0], b[1], ho[2], lo[2]);
  split_pair(b[2], b[3], ho[3], lo[3]);
  out[2 * i] = ho;
  out[2 * i + 1] = lo;
}

// ---- gradient rows -> split rows under a per-tensor power-of-two scale (round 5, training) -----------------------------------
// The fp16 pair format holds activations (|x| < 2047 at the fixed scale 2^5) but not gradients, whose magnitude is set by the loss:
// rounds 4-5 ran the input-gradient convolutions on three bf16 parts (six products, ~3x the two-part kernel's time on the
// 64-channel layers).  A gradient TENSOR has a narrow range relative to its own largest value, though: scaled by the power of two
// that puts its largest |value| into [512, 1024), every entry >= 2^-19 of that maximum keeps its 22 bits and smaller ones are
// off by <= 2^-40 of the maximum -- fp32-grade "of scale", which is what the parity bounds measure.  rows_absmax_kernel reduces
// |x| into one word (fp32 bit patterns of non-negative values order like unsigned integers), pow2_scale_kernel turns it into
// s = 2^k and a [channels] vector of 1 / s for the convolution's epilogue, split_rows_scaled_kernel splits s * x.
// inf / NaN: the scale stays 1 and the checked split raises the range flag.
// per-workgroup maxima (no atomics, nothing to clear): part[blockIdx.x] = max |x| over the workgroup's strided share
__global__ __launch_bounds__(256) void rows_absmax_kernel(const float *__restrict__ x, size_t n4, size_t n, float *__restrict__ part) {
  float m = 0.f;
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  auto upd = [&](f32x4 v) {
    m = fmaxf(fmaxf(m, fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1]))), fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3])));
    if (!(v[0] == v[0]) || !(v[1] == v[1]) || !(v[2] == v[2]) || !(v[3] == v[3])) m = __builtin_inff();   // NaN must not hide
  };
  for (; i + 3 * stride < n4; i += 4 * stride) {                 // four loads in flight
    const f32x4 v0 = ((const f32x4 *)x)[i], v1 = ((const f32x4 *)x)[i + stride], v2 = ((const f32x4 *)x)[i + 2 * stride],
                v3 = ((const f32x4 *)x)[i + 3 * stride];
    upd(v0), upd(v1), upd(v2), upd(v3);
  }
  for (; i < n4; i += stride) upd(((const f32x4 *)x)[i]);
  if (blockIdx.x == 0)
    for (size_t t = n4 * 4 + threadIdx.x; t < n; t += 256) m = fmaxf(m, __builtin_fabsf(x[t]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
}

// s = 2^k from the per-workgroup maxima (every workgroup of this launch reduces them itself: at most POW2_PARTS values), and the
// [channels] vector of 1 / s
constexpr int POW2_PARTS = 1024;
__global__ __launch_bounds__(256) void pow2_scale_kernel(const float *__restrict__ part, int nparts, int channels,
                                                         float *__restrict__ scale, float *__restrict__ inv) {
  float m = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) m = fmaxf(m, part[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  const float amax = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
  float s = 1.f;
  if (amax > 0.f && amax < __builtin_inff()) {
    // exponent e of amax (amax in [2^e, 2^(e+1))): s = 2^(9 - e) puts it into [512, 1024); clamped to fp32's normal range
    int k = 9 - ((int)((__float_as_uint(amax) >> 23) & 0xffu) - 127);
    k = k > 120 ? 120 : (k < -120 ? -120 : k);
    s = __uint_as_float((unsigned)(k + 127) << 23);
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) scale[0] = s, scale[1] = amax;
  if (i < channels) inv[i] = 1.f / s;
}

__global__ __launch_bounds__(256) void split_rows_scaled_kernel(const float *__restrict__ x, size_t nblk, const float *__restrict__ scale,
                                                                u32x4 *__restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nblk) return;
  const float s = *scale;
  f32x4 a = ((const f32x4 *)x)[2 * i] * s, b = ((const f32x4 *)x)[2 * i + 1] * s;
  u32x4 ho, lo;
  split_pair(a[0], a[1], ho[0], lo[0]);
  split_pair(a[2], a[3], ho[1], lo[1]);
  split_pair(b[0], b[1], ho[2], lo[2]);
  split_pair(b[2], b[3], ho[3], lo[3]);
  out[2 * i] = ho;
  out[2 * i + 1] = lo;
}

// fp32 rows -> three-part rows [n][C/8][hi 8 x bf16 | mid | lo] (48 B per 8 channels)
__global__ __launch_bounds__(256) void split3_rows_kernel(const float *__restrict__ x, size_t nblk, u32x4 *__restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nblk) return;
  f32x4 a = ((const f32x4 *)x)[2 * i], b = ((const f32x4 *)x)[2 * i + 1];
  u32x4 ho, mo, lo;
  split3_pair(a[0], a[1], ho[0], mo[0], lo[0]);
  split3_pair(a[2], a[3], ho[1], mo[1], lo[1]);
  split3_pair(b[0], b[1], ho[2], mo[2], lo[2]);
  split3_pair(b[2], b[3], ho[3], mo[3], lo[3]);
  out[3 * i] = ho;
  out[3 * i + 1] = mo;
  out[3 * i + 2] = lo;
}

// W -> packed B operands of the output-stationary kernel with THREE parts: [half][k][kb][ct][hi|mid|lo][lane] (layout 1)
__global__ __launch_bounds__(256) void pack_weights3_kernel(const float *__restrict__ w, int K, int cin, int cout,
                                                            u32x4 *__restrict__ out) {
  const int KB = cin / 32;
  size_t total = (size_t)K * cin * cout / 8 * 3;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int lane = (int)(i & 63);
  size_t r = i >> 6;
  int part = (int)(r % 3); r /= 3;
  int n = lane & 15, g = lane >> 4;
  const int CW = cout > 128 ? 128 : cout, CT = CW / 16;
  int ct = (int)(r % CT); r /= CT;
  int kb = (int)(r % KB); r /= KB;
  int k = (int)(r % K);
  int col = (int)(r / K) * CW + n * CT + ct;
  int ch0 = kb * 32 + g * 8;
  unsigned v[8];
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    unsigned h, m, l;
    split3_pair(w[((size_t)k * cin + ch0 + e) * cout + col], w[((size_t)k * cin + ch0 + e + 1) * cout + col], h, m, l);
    v[e >> 1] = part == 0 ? h : (part == 1 ? m : l);
  }
  out[i] = (u32x4){v[0], v[1], v[2], v[3]};
}

// ---- W[K][CIN][COUT] fp32 -> packed B operands -----------------------------------------------------------
// layout 0 (spconv_split_kernel, pair-compacted):   [k][wave 4][kb][ct][hi|lo][lane], lane (n,g) -> column
//          wave*COUT/4 + CT*n + ct, channels g*CIN/4 + kb*8 + e
// layout 1 (spconv_os_split_kernel, output-stationary): [k][kb][ct][hi|lo][lane], lane (n,g) -> column n*CT + ct,
//          channels kb*32 + g*8 + e.  256-column filters are two such 128-column halves back to back
//          ([half][k][kb][ct 8][hi|lo][lane], column half*128 + n*8 + ct): a workgroup computes one half.
__global__ __launch_bounds__(256) void pack_weights_kernel(const float *__restrict__ w, int K, int cin, int cout,
                                                           int layout, u32x4 *__restrict__ out) {
  const int KB = cin / 32;
  size_t total = (size_t)K * cin * cout / 4;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int lane = (int)(i & 63);
  size_t r = i >> 6;
  int part = (int)(r & 1); r >>= 1;
  int n = lane & 15, g = lane >> 4;
  int k, kb, col, ch0;
  if (layout == 0) {
    const int CS = cout / 4, CT = CS / 16;
    int ct = (int)(r % CT); r /= CT;
    kb = (int)(r % KB); r /= KB;
    int wave = (int)(r & 3);
    k = (int)(r >> 2);
    col = wave * CS + CT * n + ct;
    ch0 = g * (cin / 4) + kb * 8;
  } else {
    const int CW = cout > 128 ? 128 : cout, CT = CW / 16;
    int ct = (int)(r % CT); r /= CT;
    kb = (int)(r % KB); r /= KB;
    k = (int)(r % K);
    col = (int)(r / K) * CW + n * CT + ct;     // lane n owns CT consecutive output columns -> vector epilogue
    ch0 = kb * 32 + g * 8;
  }
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    unsigned hi, lo;
    split_pair_w(w[((size_t)k * cin + ch0 + 2 * e) * cout + col], w[((size_t)k * cin + ch0 + 2 * e + 1) * cout + col], hi, lo);
    o[e] = part ? lo : hi;
  }
  out[i] = o;
}

struct SplitConvArgs {
  const u32x4 *feat;     // split rows of the input
  const u32x4 *w;        // packed weights
  const int32_t *nbr;
  const float *bias, *scale, *shift, *residual;
  float *out;
  u32x4 *out_split;      // optional split rows of the output
  int n_in, n_out, K, relu;
  int dbg;               // tuning experiments (DF3D_OS_DBG): 1 = no gathers, 2 = no W staging, 4 = no MFMAs
  // grouped / strided rows (output-stationary kernel only; blockIdx.y = column block or group):
  int ldi;               // u32x4 per split input row
  int in_goff;           // u32x4 added to the input row per blockIdx.y (0: every block reads the same channels)
  int ldo;               // floats per output row
  int gy;                // gridDim.y
  const int32_t *cols;   // optional [gy][2] = (first output column, valid columns) of a block: compact fp32 stores
  const int32_t *order;  // optional [n_out]: tile t of the output-stationary kernel owns rows order[t*TM .. t*TM+TM-1]
  unsigned long long *trace;   // builds with -DDF3D_OS_TRACE: [workgroup][wave][8] s_memtime stamps (tuning aid)
  int cbw;               // loader / consumer kernel: column blocks (or groups) one workgroup walks (0 = 1), gridDim.y = ceil(gy / cbw)
};

#ifdef DF3D_OS_TRACE
#define OS_STAMP(slot)                                                                                          \
  do {                                                                                                          \
    if (a.trace && (threadIdx.x & 63) == 0)                                                                     \
      a.trace[((size_t)(blockIdx.x + gridDim.x * blockIdx.y) * 16 + (threadIdx.x >> 6)) * 8 + (slot)] =        \
          __builtin_amdgcn_s_memtime();                                                                         \
  } while (0)
static unsigned long long *g_os_trace = nullptr;
extern "C" void df3d_debug_set_os_trace(void *p) { g_os_trace = (unsigned long long *)p; }
#else
#define OS_STAMP(slot) do { } while (0)
#endif

// tuning experiments of the output-stationary kernel (DF3D_OS_DBG bits: 1 no gathers, 2 no W staging, 4 no MFMAs,
// 8 raised wave priority around the MFMAs, 16 MFMAs without B-operand LDS reads, 32 no output stores) exist only in
// builds with -DDF3D_OS_EXPERIMENTS: as run-time flags they cost the production kernel ~20 % (uniform branches
// inside the MFMA batches)
// DF3D_OS_QGATHER (build flag, round-3 experiment): the output-stationary kernel issues its gathers quad-coalesced and permutes
#ifdef DF3D_OS_QGATHER
#define OS_QROW (lane >> 2)
#define OS_QSUB (lane & 3)
#else
#define OS_QROW n
#define OS_QSUB g
#endif
#ifdef DF3D_OS_EXPERIMENTS
#define OS_DBG(bit) ((a.dbg & (bit)) != 0)
#else
#define OS_DBG(bit) false
#endif

#define DF3D_MFMA_BF16(A, B, C) \
  __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), C, 0, 0, 0)
// the matrix instruction of a precision mode: two-part operands are fp16 pairs (common.h), one / three parts bf16
template <int NP>
__device__ __forceinline__ f32x4 mfma_parts(const u32x4 &A, const u32x4 &B, const f32x4 &C) {
  if constexpr (NP == 2) return DF3D_MFMA_F16(A, B, C);
  else return DF3D_MFMA_BF16(A, B, C);
}
// accumulator -> fp32 value
template <int NP>
__device__ __forceinline__ constexpr float acc_unscale() { return NP == 2 ? DF3D_ACC_UNSCALE : 1.0f; }

template <int CIN, int COUT>
__global__ __launch_bounds__(256, 1) void spconv_split_kernel(SplitConvArgs a, int TM,
                                                               const int32_t *__restrict__ tile_rows) {
  constexpr int KB = CIN / 32;       // 32-deep MFMA k-blocks; lane group g owns channels [g*CIN/4, (g+1)*CIN/4)
  constexpr int CS = COUT / 4;       // output columns per wave
  constexpr int CT = CS / 16;        // 16-wide column tiles per wave (1 or 2)
  constexpr int NCH = 2 / CT;        // chunks per item -> always 2 independent accumulators
  constexpr int RQ = CIN / 4;        // u32x4 per split row
  static_assert(CT == 1 || CT == 2, "COUT must be 64 or 128");
  static_assert(CIN % 32 == 0, "CIN must be a multiple of 32");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *accL = (float *)smem;                                  // [4][TM+1][CS]; row TM = trash row
  int *idxL = (int *)(smem + (size_t)4 * (TM + 1) * CS * 4);    // [K][TM]: nbr tile, compacted in place to input rows
  unsigned short *listL = (unsigned short *)(idxL + a.K * TM);  // [K][TM]: matching output rows (tile-local)
  __shared__ int cntL[DF3D_MAX_KVOL];
  __shared__ int segL[DF3D_MAX_KVOL + 1];                       // first item of each ACTIVE offset; [nact] = T
  __shared__ int actL[DF3D_MAX_KVOL + 1];                       // active offsets in order
  __shared__ int nactL;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;
  const int cs0 = wave * CS;
  const int K = a.K;
  const int range0 = tile_rows ? tile_rows[blockIdx.x] : blockIdx.x * TM;
  const int range1 = tile_rows ? tile_rows[blockIdx.x + 1] : min(range0 + TM, a.n_out);
  for (int row0 = range0; row0 < range1; row0 += TM) {
    const int row_end = min(row0 + TM, range1);
    __syncthreads();

    for (int e = tid; e < K * TM; e += 256) {
      int k = e / TM, r = e - k * TM;
      int row = row0 + r;
      idxL[e] = (row < row_end) ? a.nbr[(size_t)k * a.n_out + row] : -1;
    }
    float *myacc = accL + (size_t)wave * (TM + 1) * CS;
    for (int e = lane; e < TM * CS / 4; e += 64) ((f32x4 *)myacc)[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    // ---- compact the valid pairs of every offset (offsets striped over the waves) ----
    for (int k = wave; k < K; k += 4) {
      int cnt = 0;
      for (int r0 = 0; r0 < TM; r0 += 64) {
        int r = r0 + lane;
        int v = r < TM ? idxL[k * TM + r] : -1;
        bool valid = v >= 0;
        unsigned long long m = __ballot(valid);
        int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
        __builtin_amdgcn_wave_barrier();
        if (valid) {
          idxL[k * TM + pos] = v;
          listL[k * TM + pos] = (unsigned short)r;
        }
        cnt += __popcll(m);
      }
      if (lane == 0) cntL[k] = cnt;
    }
    __syncthreads();

    // ---- item list: one entry per (offset, chunk group): k | grp << 8 | cnt << 16, built by one lane per offset ----
    int *itemL = (int *)(listL + (size_t)K * TM + (((size_t)K * TM) & 1));     // 4-byte aligned, [T+3]
    auto ngroups = [&](int cnt) { return (((cnt + 15) >> 4) + NCH - 1) / NCH; };
    if (tid < 32) {
      int cnt = tid < K ? cntL[tid] : 0;
      int ng = ngroups(cnt);
      int start = 0, rank = 0, tot = 0, nact = 0;
      for (int j = 0; j < K; ++j) {                // K <= 32: every lane scans the counts once
        int c = cntL[j];
        int gj = ngroups(c);
        if (j < tid) { start += gj; rank += gj > 0; }
        tot += gj;
        nact += gj > 0;
      }
      if (tid < K && ng > 0) {
        segL[rank] = start;
        actL[rank] = tid;
        for (int gq = 0; gq < ng; ++gq) itemL[start + gq] = tid | (gq << 8) | (cnt << 16);
      }
      if (tid == 0) {
        segL[nact] = tot;
        actL[nact] = 0;
        nactL = nact;
      }
    }
    __syncthreads();
    const int nact = nactL;
    const int T = segL[nact];
    if (tid < 3 && T > 0) itemL[T + tid] = itemL[T - 1];   // sentinels: the look-ahead re-reads the last item
    __syncthreads();

    u32x4 bc[KB][CT][2], bn[KB][CT][2];
    auto load_b = [&](int k, u32x4 (&dst)[KB][CT][2]) {
      const u32x4 *wk = a.w + ((size_t)(k * 4 + wave) * (KB * CT * 2)) * 64 + lane;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int p = 0; p < 2; ++p) dst[kb][ct][p] = wk[((kb * CT + ct) * 2 + p) * 64];
    };
    auto read_idx = [&](int item, int (&idx)[NCH]) {
      int k = item & 0xff, grp = (item >> 8) & 0xff, cnt = item >> 16;
#pragma unroll
      for (int h = 0; h < NCH; ++h) {
        int p = (grp * NCH + h) * 16 + n;
        p = p < cnt ? p : cnt - 1;
        idx[h] = idxL[k * TM + p];
      }
    };
    auto read_rows = [&](int item, int (&rl)[NCH][4]) {
      int k = item & 0xff, grp = (item >> 8) & 0xff, cnt = item >> 16;
#pragma unroll
      for (int h = 0; h < NCH; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int p = (grp * NCH + h) * 16 + 4 * g + r;
          int pp = p < cnt ? p : cnt - 1;
          int t = listL[k * TM + pp];
          rl[h][r] = p < cnt ? t : TM;              // invalid slots -> trash row
        }
    };
    // A fragments of items t .. t+3 (the gathers run 3 items ahead): [slot][chunk][k-block][hi|lo]
    u32x4 s0[NCH][KB][2], s1[NCH][KB][2], s2[NCH][KB][2], s3[NCH][KB][2];
    auto load_a = [&](const int (&idx)[NCH], u32x4 (&dst)[NCH][KB][2]) {
#pragma unroll
      for (int h = 0; h < NCH; ++h) {
        const u32x4 *src = a.feat + (size_t)idx[h] * RQ + (size_t)g * KB * 2;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          dst[h][kb][0] = src[kb * 2];
          dst[h][kb][1] = src[kb * 2 + 1];
        }
      }
    };
    f32x4 acc[NCH][CT], aprev[NCH][CT];
    float2 fo2[NCH][4];
    float fo1[NCH][4];
    auto flush_read = [&](const int (&rl)[NCH][4]) {
#pragma unroll
      for (int h = 0; h < NCH; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (CT == 2) fo2[h][r] = *(const float2 *)(myacc + (size_t)rl[h][r] * CS + 2 * n);
          else fo1[h][r] = myacc[(size_t)rl[h][r] * CS + n];
        }
    };
    auto flush_write = [&](const int (&rl)[NCH][4], f32x4 (&v)[NCH][CT]) {
#pragma unroll
      for (int h = 0; h < NCH; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (CT == 2) {
            float2 o = fo2[h][r];
            o.x += v[h][0][r];
            o.y += v[h][CT - 1][r];
            *(float2 *)(myacc + (size_t)rl[h][r] * CS + 2 * n) = o;
          } else {
            myacc[(size_t)rl[h][r] * CS + n] = fo1[h][r] + v[h][0][r];
          }
        }
    };

    int idxn[NCH], rl_cur[NCH][4], rl_prev[NCH][4];
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) aprev[h][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) rl_prev[h][r] = TM;       // first flush goes to the trash row
    }
    if (T > 0) {
      load_b(actL[0], bn);
      read_idx(itemL[0], idxn);
      load_a(idxn, s0);
      read_idx(itemL[1], idxn);
      load_a(idxn, s1);
      read_idx(itemL[2], idxn);
      load_a(idxn, s2);
    }
    // One item = NCH chunks of 16 pairs = 3*KB*NCH*CT MFMAs.  The items of the whole tile form ONE software
    // pipeline (no bubble at offset boundaries), unrolled four times so that the A ring slots are fixed
    // registers.  When the offset changes, the prefetched weights move bn -> bc (once per ~T/K items) and the
    // fetch of the following offset's weights starts.
    int kcur = -1, ai = -1;
    auto step = [&](int t, u32x4 (&cur)[NCH][KB][2], u32x4 (&tgt)[NCH][KB][2]) {
      const int it0 = __builtin_amdgcn_readfirstlane(itemL[t]);
      const int it3 = __builtin_amdgcn_readfirstlane(itemL[t + 3]);
      if ((it0 & 0xff) != kcur) {
        kcur = it0 & 0xff;
        ++ai;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int p = 0; p < 2; ++p) bc[kb][ct][p] = bn[kb][ct][p];
        if (ai + 1 < nact) load_b(actL[ai + 1], bn);
      }
      read_idx(it3, idxn);
      read_rows(it0, rl_cur);
      flush_read(rl_prev);
#pragma unroll
      for (int h = 0; h < NCH; ++h)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[h][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int term = 0; term < 3; ++term)          // lo*hi, hi*lo first (small), hi*hi last
#pragma unroll
          for (int h = 0; h < NCH; ++h)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
              const int pa = term == 0 ? 1 : 0, pb = term == 1 ? 1 : 0;
              acc[h][ct] = DF3D_MFMA_F16(cur[h][kb][pa], bc[kb][ct][pb], acc[h][ct]);
            }
      load_a(idxn, tgt);                                  // gathers of item t+3 (into the slot item t-1 freed)
      flush_write(rl_prev, aprev);                        // item t-1's LDS update
#pragma unroll
      for (int h = 0; h < NCH; ++h) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) aprev[h][ct] = acc[h][ct];
#pragma unroll
        for (int r = 0; r < 4; ++r) rl_prev[h][r] = rl_cur[h][r];
      }
    };
    for (int t = 0; t < T; t += 4) {
      step(t, s0, s3);
      if (t + 1 < T) step(t + 1, s1, s0);
      if (t + 2 < T) step(t + 2, s2, s1);
      if (t + 3 < T) step(t + 3, s3, s2);
    }
    flush_read(rl_prev);
    flush_write(rl_prev, aprev);
    __builtin_amdgcn_wave_barrier();

    // ---- epilogue: this wave's CS columns of every row of the tile.  LDS column c of the wave's slice holds
    //      output column cs0 + c (the packed weights put tile ct of lane n at column CT*n + ct) ----
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
      if (rl < TM && row < row_end) {
        f32x4 v = *(const f32x4 *)(myacc + (size_t)rl * CS + lc);
        v = (v * DF3D_ACC_UNSCALE + bi) * sc + sh;
        size_t o = (size_t)row * COUT + cs0 + lc;
        if (a.residual) v += *(const f32x4 *)(a.residual + o);
        if (a.relu) {
          v[0] = fmaxf(v[0], 0.f);
          v[1] = fmaxf(v[1], 0.f);
          v[2] = fmaxf(v[2], 0.f);
          v[3] = fmaxf(v[3], 0.f);
        }
        *(f32x4 *)(a.out + o) = v;
        if (a.out_split) {
          unsigned h[2], l[2];
          split_pair(v[0], v[1], h[0], l[0]);
          split_pair(v[2], v[3], h[1], l[1]);
          // 8-channel block = [hi 16 B | lo 16 B]; this lane owns 4 of the 8 channels
          char *blk = (char *)a.out_split + (o >> 3) * 32 + ((o >> 2) & 1) * 8;
          *(u32x2 *)blk = (u32x2){h[0], h[1]};
          *(u32x2 *)(blk + 16) = (u32x2){l[0], l[1]};
        }
      }
    }
  }  // passes over the row range
}

// (offset, channel block) of a step of the output-stationary kernels, advanced one step at a time from the set of the
// tile's active offsets.  Everything lives in scalar registers: looking the offset of a step up in LDS put one (and with
// the neighbour index two) exposed LDS round trips into every step's instruction stream, right behind the MFMA batch --
// an in-order wave cannot hide them, and the barrier re-aligns the two waves of a SIMD every step.
// Past the last step the cursor stays on it (a valid weight tile) with live = false (zero A rows).
// Workgroups go to the eight XCDs round-robin by their linear id; this maps id -> tile so that every XCD (= every L2)
// owns one contiguous range of tiles, for any tile count.
__device__ __forceinline__ int xcd_tile(int bid, int nt) {
  const int q = nt >> 3, r = nt & 7, x = bid & 7;
  return x * q + (x < r ? x : r) + (bid >> 3);
}

// 32-bit LDS byte address of a __shared__ object (operand of the inline-asm ds_read_b128 below)
__device__ __forceinline__ unsigned lds_addr(const void *p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char *)p;
}

struct StepCursor {
  unsigned m;
  int k, kb;
  bool live;
  __device__ __forceinline__ void init(unsigned mask) {
    m = mask;
    live = mask != 0u;
    k = live ? __builtin_ctz(mask) : 0;
    kb = 0;
  }
  template <int KB>
  __device__ __forceinline__ void next() {
    if (!live) return;
    if (++kb == KB) {
      const unsigned m2 = m & (m - 1u);
      if (m2) {
        m = m2;
        k = __builtin_ctz(m2);
        kb = 0;
      } else {
        live = false;
        kb = KB - 1;
      }
    }
  }
};

// rows without a neighbour gather this all-zero split row (keeps the gathers branch-free, so that the
// compiler can count its vmcnt waits instead of draining the whole load queue at every step)
__device__ u32x4 g_zero_row[192];          // up to 512 input channels, three parts

// ---------------------------------------------------------------------------------------------------------
// Output-stationary variant: a wave owns 16*RT output rows and all COUT columns, accumulators in registers,
// the packed B operands of one (offset, 32-channel block) step staged through LDS (double buffered, one
// barrier per step), A fragments (32 B of split row per lane) straight from L2/HBM one step ahead.  MFMAs are
// also issued for rows without a neighbour at an offset (zero operands) -- at 3/16 of the fp32 cost that waste
// is cheaper than the pair compaction -- and several workgroups per CU hide the gather latency.
// NP = precision parts of the operands: 2 = fp32 rows split into bf16 hi + lo (three MFMA products per pair, fp32-grade
// result), 1 = plain bf16 rows and bf16 weights (one product, fp32 accumulate; rows are [N][C] bf16, the packed filter
// bank holds the hi parts only, `out_split` receives bf16 rows and `residual` is read as bf16 rows).
// KS = wave groups per workgroup that split the OFFSETS of the tile between them (each with its own weight stages, the
// accumulators meet in LDS before the epilogue): the 90 x 90 maps of the BEV neck have 127 row tiles x 2 column halves for
// 256 CUs -- one wave per SIMD, every step an exposed chain of barrier, fragment reads and matrix instructions (a step costs
// ~1500 clocks whatever its MFMA count); three groups put three waves on every SIMD without a second pass over memory.
// MG = masked gathers (round 5): a lane whose row has no neighbour at the offset issues NO request (exec-masked loads into
// zeroed registers) instead of reading the all-zero row.  The sparse 3 x 3 x 3 layers have ~45 % such lanes, and every lane
// of a gather instruction costs the texture path a slot whatever it reads (tools/ubench/qperm_probe.hip: gathers + MFMAs
// of a 64-channel K = 27 layer 52 -> 41 us); dense maps (neck, head) have none and keep the branch-free form.
// (Round 5, measured and dropped: capping the 64-column two-part kernel at 80 registers for six waves per SIMD -- three 8-wave
// workgroups per CU, all 519 tiles of a 66 k-row layer resident at once instead of 512 + a round of 7 -- spills 18 dwords
// into the step loop: 61 -> 173 us.)
template <int CIN, int COUT, int RT, int NW, int KPS, int NP = 2, int KS = 1, bool MG = false>
__global__ __launch_bounds__(NW * KS * 64) void spconv_os_split_kernel(SplitConvArgs a) {
  // KPS = 32-channel blocks per step (one barrier per step)
  // COUT = 256 is computed as two 128-column halves by different workgroups (blockIdx.y): twice the workgroups for
  // the small dense maps of the BEV neck and half the accumulator registers; the gathers of the second half hit L2
  constexpr int CW = COUT > 128 ? 128 : COUT;
  constexpr int KB = CIN / 32 / KPS, CT = CW / 16, TM = 16 * RT * NW, WROWS = 16 * RT, RQ = CIN / 4;
  constexpr int WQ = KPS * CT * NP * 64;          // u32x4 per W step tile
  constexpr int NT = NW * 64;
  constexpr int WPT = (WQ + NT - 1) / NT;
  // Round 5: THREE weight stages where the tile is small (one wave group, <= 4 staging loads per thread): the stage of step
  // s + 1 is then complete one barrier EARLIER, so a wave reads the first B fragments of step s + 1 behind the last matrix
  // instructions of step s -- before the barrier -- and issues MFMAs right after it.  With two stages every step began with an
  // exposed LDS round trip (ISA of round 4: s_barrier, 4 x ds_read_b128, s_waitcnt, MFMA ...; the compiler does not move LDS
  // reads across a barrier, and a step of 12 MFMAs is ~190 clocks of matrix work against ~120 of that round trip).
  constexpr bool RING3 = KS == 1 && WPT <= 4;
  constexpr int NST = RING3 ? 3 : 2;
  __shared__ u32x4 Wl[KS][NST][WQ];
  __shared__ int nbrL[DF3D_MAX_KVOL][TM];
  __shared__ int rowL[TM];
  __shared__ unsigned wg_mask;

  // (with KS groups: `tid` / `wave` are group-local below the neighbour-table pass, `grp` is uniform per wave)
  const int wtid = threadIdx.x, lane = wtid & 63;
  const int grp = KS > 1 ? __builtin_amdgcn_readfirstlane(wtid / NT) : 0;
  const int tid = KS > 1 ? wtid - grp * NT : wtid, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;
  int nt = gridDim.x, bid = blockIdx.x, tile = bid;
  tile = xcd_tile(bid, nt);                       // consecutive tiles stay on one XCD / L2
  const int row0 = tile * TM;
  const int col0 = blockIdx.y * CW;

  OS_STAMP(0);
  if (wtid == 0) wg_mask = 0u;
  __syncthreads();
  // neighbour tile -> LDS; a wave covers 64 rows (or TM) of one offset per pass, so the set of offsets with at
  // least one neighbour in the tile falls out of the same pass (one ballot per load)
  {
    constexpr int KSTEP = NT * KS / TM;          // offsets covered per pass (NT = 4*TM/RT... >= 1)
    static_assert(NT % TM == 0 && KSTEP >= 1, "tile shape");
    const int r = wtid % TM, k0 = wtid / TM;
    // the tile's rows: consecutive, or -- with a tiling order -- whatever rows the order puts next to each other
    // (rows that are neighbours in space and share their neighbour pattern: fewer active offsets per tile, gathers
    // that stay inside one XCD's L2); n_out marks the padding of the last tile
    int row = row0 + r;
    row = row < a.n_out ? (a.order ? a.order[row] : row) : a.n_out;
    if (k0 == 0) rowL[r] = row;
    unsigned mine = 0u;
    int v[(DF3D_MAX_KVOL + KSTEP - 1) / KSTEP];
#pragma unroll
    for (int i = 0; i < (DF3D_MAX_KVOL + KSTEP - 1) / KSTEP; ++i) {
      int k = k0 + i * KSTEP;
      v[i] = (k < a.K && row < a.n_out) ? a.nbr[(size_t)k * a.n_out + row] : -1;
    }
#pragma unroll
    for (int i = 0; i < (DF3D_MAX_KVOL + KSTEP - 1) / KSTEP; ++i) {
      int k = k0 + i * KSTEP;
      if (k < a.K) nbrL[k][r] = v[i];
      if (TM >= 64) {                            // the whole wave looks at one offset
        if (__ballot(v[i] >= 0) != 0ull) mine |= 1u << (k & 31);
      } else if (v[i] >= 0) {
        atomicOr(&wg_mask, 1u << k);
      }
    }
    if (TM >= 64 && lane == 0 && mine) atomicOr(&wg_mask, mine);
  }
  __syncthreads();
  unsigned gmask = __builtin_amdgcn_readfirstlane(wg_mask);
  int nact = __popc(gmask);
  if constexpr (KS > 1) {
    // this group's share of the active offsets: a contiguous run of ceil(nact / KS) of them; every group walks as many
    // steps as the longest share (the barriers are the workgroup's), the others multiply zero rows at the end
    const int per = (nact + KS - 1) / KS;
    unsigned rest = gmask, mine = 0u;
    for (int i = 0; rest; ++i, rest &= rest - 1u)
      if (i / per == grp) mine |= rest & (0u - rest);
    gmask = mine;
    nact = per;
  }
  const int steps = nact * KB;

  f32x4 acc[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Every step issues the same loads, unconditionally: W tile of step min(s+2, last) and the A fragments of step
  // s+3 (the zero row past the end or where a row has no neighbour).  Steps are padded to a multiple of 3
  // (the A ring rotates by name); padding steps multiply zeros.
  // The W tile of step s+4 is fetched at step s and parked in one of three register sets until step s+3 stores it
  // to LDS: three whole steps cover the L2 / HBM latency (with a single set the store waited for a load issued one
  // step earlier, which made every step at least one memory round trip long).  WD = 1 where a set is too large.
  // Which (offset, channel block) a fetch belongs to comes from two scalar cursors (cw for the weight stream, ca for
  // the gathers), each advanced once per fetch.
  constexpr int WD = WPT <= 4 ? 3 : 1;
  u32x4 w0[WPT], w1[WPT], w2[WPT];
  StepCursor cw, ca;
  cw.init(gmask);
  ca.init(gmask);
  auto load_w = [&](u32x4 (&wreg)[WPT]) {          // the next tile of the weight stream
    const u32x4 *src = a.w + ((size_t)blockIdx.y * a.K * KB + (size_t)(cw.k * KB + cw.kb)) * WQ;   // KPS consecutive [kb] tiles
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      int e = tid + NT * i;
      wreg[i] = src[(WQ % NT == 0 || e < WQ) ? e : 0];
    }
    cw.template next<KB>();
  };
  auto store_w = [&](int buf, u32x4 (&wreg)[WPT]) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      int e = tid + NT * i;
      if (WQ % NT == 0 || e < WQ) Wl[grp][buf][e] = wreg[i];
    }
  };
  // A fragments of the next AD steps in a register ring (AD sets, rotated by name: the step loop is unrolled AD times).
  // The gathers of a step are issued AD - 1 whole steps before its matrix instructions need them.  Measured on MI355X
  // (tools/ubench/sk_probe.py, -DDF3D_OS_ADEPTH=6): depth 6 changes nothing (conv4 93.4 vs 91.6 us) -- the gathers are
  // not what a step waits for -- and costs 24 registers, so the ring stays at 3.
#ifndef DF3D_OS_ADEPTH
#define DF3D_OS_ADEPTH 3
#endif
  constexpr int AD = (RT * KPS * NP <= 4) ? DF3D_OS_ADEPTH : 3;
  static_assert(AD % 3 == 0, "the W register sets rotate with period 3");
  u32x4 ar[AD][RT][KPS][NP];
  // the gathers of the next step of the A stream, in two halves: peek_a reads the neighbour indices from LDS (issued at
  // the TOP of a step, so that the round trip runs under the step's matrix instructions), issue_a turns them into
  // addresses and issues the loads (at the END of the step, into the fragment registers the step has just freed)
  int idxn[RT];
  auto peek_a = [&]() {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) idxn[rt] = nbrL[ca.k][wave * WROWS + rt * 16 + OS_QROW];
  };
  auto issue_a = [&](u32x4 (&dst)[RT][KPS][NP]) {
    const int kb = ca.kb * KPS;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int idx = idxn[rt];
      if constexpr (MG) {
#pragma unroll
        for (int j = 0; j < KPS; ++j)
#pragma unroll
          for (int q = 0; q < NP; ++q) dst[rt][j][q] = (u32x4){0u, 0u, 0u, 0u};
        if (ca.live && idx >= 0) {
          const u32x4 *p = a.feat + (size_t)idx * a.ldi + blockIdx.y * a.in_goff + (kb * 4 + OS_QSUB) * NP;
#pragma unroll
          for (int j = 0; j < KPS; ++j)
#pragma unroll
            for (int q = 0; q < NP; ++q) dst[rt][j][q] = p[j * 4 * NP + q];
        }
        continue;
      }
      const u32x4 *p = (ca.live && idx >= 0 && !OS_DBG(1)) ? a.feat + (size_t)idx * a.ldi + blockIdx.y * a.in_goff
                                                             : g_zero_row;
      p += (kb * 4 + OS_QSUB) * NP;
#pragma unroll
      for (int j = 0; j < KPS; ++j) {
#pragma unroll
        for (int q = 0; q < NP; ++q) dst[rt][j][q] = p[j * 4 * NP + q];
      }
    }
    ca.template next<KB>();
  };
  // Step s: barrier; W(s+1) (fetched during step s-1) -> LDS; fetch W(s+2); MFMAs of step s; fetch A(s+3).
#ifdef DF3D_OS_QGATHER
  // quad-coalesced gathers (lane l loaded sub-block l & 3 of row l >> 2): brought into the MFMA operand shape by ds_bpermute
  // ONE STEP AHEAD of their use, into a second register set, so that the permute's round trip runs under the barrier and
  // the weight staging of the next step
  u32x4 rdy[RT][KPS][NP];
  auto to_operand = [&](u32x4 (&raw)[RT][KPS][NP]) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int j = 0; j < KPS; ++j)
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
          for (int d = 0; d < 4; ++d)
            rdy[rt][j][q][d] = (unsigned)__builtin_amdgcn_ds_bpermute((4 * n + g) * 4, (int)raw[rt][j][q][d]);
  };
#endif
  // RING3: the first B fragments of the next step are read before its barrier -- into fragment slot 0 itself when a step has
  // an even number of column-pair batches (its last batch reads slot 1), else into a set of their own
  constexpr int NBATCH_ = KPS * CT / 2;
  constexpr bool BNEXT_IN_PLACE = RING3 && NBATCH_ % 2 == 0;
  u32x4 bnext[BNEXT_IN_PLACE ? 1 : 2 * NP];
  u32x4 bq[2][2 * NP];
  auto step = [&](int s, u32x4 (&raw)[RT][KPS][NP], u32x4 (&rawn)[RT][KPS][NP], u32x4 (&wset)[WPT]) {
#ifdef DF3D_OS_QGATHER
    u32x4 (&cur)[RT][KPS][NP] = rdy;
#else
    u32x4 (&cur)[RT][KPS][NP] = raw;
#endif
    __syncthreads();
    // the next step's weight tile goes to LDS (and the one after the ring is fetched) in the shadow of the first MFMA
    // batch instead of in front of it: right after the barrier a wave should do nothing but fetch B fragments and issue
    // matrix instructions (the other buffer is not read by anybody during this step)
    auto stage_w = [&]() {
      if (!OS_DBG(2)) {
        if constexpr (RING3) {
          store_w((s + 2) % 3, wset);              // W(s + 2) into the stage step s - 1 has left (everybody is past the barrier)
          load_w(wset);                            // W(s + 5)
        } else if constexpr (WD == 3) {
          store_w((s + 1) & 1, wset);
          load_w(wset);                            // W(s + 4)
        } else {
          store_w((s + 1) & 1, w0);
          load_w(w0);                              // W(s + 2)
        }
      }
    };
    peek_a();                                      // neighbour indices of step s + AD
    constexpr int WPOS = (KPS * CT / 2) > 1 ? 1 : 0;
    if (WPOS == 0) stage_w();
    const u32x4 *wb = Wl[grp][RING3 ? s % 3 : (s & 1)] + lane;
#ifdef DF3D_OS_PRODUCT_MAJOR
    // experiment: the three products of a step run product-major over groups of G column tiles, so that two matrix
    // instructions on the same accumulator are G instructions apart (the default order keeps them 2 apart, and the
    // compiler schedules them back to back)
    if constexpr (NP == 2 && RT == 1 && KPS == 1 && CT >= 4) {
      constexpr int G = 4;
      u32x4 bg[2][G][2];
#pragma unroll
      for (int c = 0; c < G; ++c) {
        bg[0][c][0] = wb[(c * 2 + 0) * 64];
        bg[0][c][1] = wb[(c * 2 + 1) * 64];
      }
#pragma unroll
      for (int b = 0; b < CT / G; ++b) {
        if (b == 0 && WPOS > 0) stage_w();
        if (b + 1 < CT / G) {
#pragma unroll
          for (int c = 0; c < G; ++c) {
            bg[(b + 1) & 1][c][0] = wb[(((b + 1) * G + c) * 2 + 0) * 64];
            bg[(b + 1) & 1][c][1] = wb[(((b + 1) * G + c) * 2 + 1) * 64];
          }
        }
#pragma unroll
        for (int c = 0; c < G; ++c) acc[0][b * G + c] = mfma_parts<NP>(cur[0][0][1], bg[b & 1][c][0], acc[0][b * G + c]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < G; ++c) acc[0][b * G + c] = mfma_parts<NP>(cur[0][0][0], bg[b & 1][c][1], acc[0][b * G + c]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < G; ++c) acc[0][b * G + c] = mfma_parts<NP>(cur[0][0][0], bg[b & 1][c][0], acc[0][b * G + c]);
        __builtin_amdgcn_sched_barrier(0);
      }
      issue_a(cur);
      return;
    }
#endif
    // B fragments of the next column pair are fetched from LDS while the MFMAs of the current pair run
    constexpr int NBATCH = KPS * CT / 2;
    if constexpr (BNEXT_IN_PLACE) {
    } else if constexpr (RING3) {
#pragma unroll
      for (int q = 0; q < 2 * NP; ++q) bq[0][q] = bnext[q];
    } else if (!OS_DBG(16)) {
#pragma unroll
      for (int q = 0; q < 2 * NP; ++q) bq[0][q] = wb[q * 64];
    }
#pragma unroll
    for (int i = 0; i < NBATCH; ++i) {
      const int j = i / (CT / 2), c2 = (i % (CT / 2)) * 2;
      if (WPOS > 0 && i == WPOS) stage_w();
      if (i + 1 < NBATCH && !OS_DBG(16)) {
#pragma unroll
        for (int q = 0; q < 2 * NP; ++q) bq[(i + 1) & 1][q] = wb[((i + 1) * 2 * NP + q) * 64];
      }
      if constexpr (NP == 3) {
        // six products per operand pair, smallest terms first: (lo,hi) (mid,mid) (hi,lo) (mid,hi) (hi,mid) (hi,hi);
        // the dropped ones are <= 2^-24 of the product
        const u32x4 *bp = bq[i & 1];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const u32x4 a0 = cur[rt][j][0], a1 = cur[rt][j][1], a2 = cur[rt][j][2];
          f32x4 c0 = acc[rt][c2], c1 = acc[rt][c2 + 1];
          c0 = mfma_parts<NP>(a2, bp[0], c0);
          c1 = mfma_parts<NP>(a2, bp[3], c1);
          c0 = mfma_parts<NP>(a1, bp[1], c0);
          c1 = mfma_parts<NP>(a1, bp[4], c1);
          c0 = mfma_parts<NP>(a0, bp[2], c0);
          c1 = mfma_parts<NP>(a0, bp[5], c1);
          c0 = mfma_parts<NP>(a1, bp[0], c0);
          c1 = mfma_parts<NP>(a1, bp[3], c1);
          c0 = mfma_parts<NP>(a0, bp[1], c0);
          c1 = mfma_parts<NP>(a0, bp[4], c1);
          c0 = mfma_parts<NP>(a0, bp[0], c0);
          c1 = mfma_parts<NP>(a0, bp[3], c1);
          acc[rt][c2] = c0, acc[rt][c2 + 1] = c1;
        }
        continue;
      }
      if constexpr (NP == 1) {
        const u32x4 b0 = bq[i & 1][0], b1 = bq[i & 1][1];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          acc[rt][c2] = mfma_parts<NP>(cur[rt][j][0], b0, acc[rt][c2]);
          acc[rt][c2 + 1] = mfma_parts<NP>(cur[rt][j][0], b1, acc[rt][c2 + 1]);
        }
        continue;
      }
      u32x4 bh0 = bq[i & 1][0], bl0 = bq[i & 1][NP >= 2 ? 1 : 0], bh1 = bq[i & 1][NP >= 2 ? 2 : 0], bl1 = bq[i & 1][NP >= 2 ? 3 : 0];
      if (OS_DBG(16)) {                    // experiment: MFMAs without the LDS reads of their B operands
        bh0 = cur[0][j][0];
        bl0 = cur[0][j][NP - 1];
        bh1 = cur[0][j][0];
        bl1 = cur[0][j][NP - 1];
      }
      if (OS_DBG(4)) continue;
      if (OS_DBG(8)) __builtin_amdgcn_s_setprio(3);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        acc[rt][c2] = mfma_parts<NP>(cur[rt][j][NP - 1], bh0, acc[rt][c2]);
        acc[rt][c2 + 1] = mfma_parts<NP>(cur[rt][j][NP - 1], bh1, acc[rt][c2 + 1]);
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        acc[rt][c2] = mfma_parts<NP>(cur[rt][j][0], bl0, acc[rt][c2]);
        acc[rt][c2 + 1] = mfma_parts<NP>(cur[rt][j][0], bl1, acc[rt][c2 + 1]);
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        acc[rt][c2] = mfma_parts<NP>(cur[rt][j][0], bh0, acc[rt][c2]);
        acc[rt][c2 + 1] = mfma_parts<NP>(cur[rt][j][0], bh1, acc[rt][c2 + 1]);
      }
      if (OS_DBG(8)) __builtin_amdgcn_s_setprio(0);
    }
    if constexpr (RING3) {                         // stage (s + 1) % 3 was stored during step s - 1: complete since this step's barrier
      const u32x4 *wn = Wl[grp][(s + 1) % 3] + lane;
#pragma unroll
      for (int q = 0; q < 2 * NP; ++q) (BNEXT_IN_PLACE ? bq[0][q] : bnext[q]) = wn[q * 64];
    }
    issue_a(raw);
#ifdef DF3D_OS_QGATHER
    to_operand(rawn);                              // the next step's fragments (gathered two steps ago)
#else
    (void)rawn;
#endif
  };

  OS_STAMP(1);
  if (steps > 0) {
    if constexpr (RING3) {
      load_w(w0);                                  // W(0), W(1): the first two stages
      load_w(w1);
      store_w(0, w0);
      store_w(1, w1);
      load_w(w1);                                  // W(2), W(3), W(4)
      load_w(w2);
      load_w(w0);
    } else {
    load_w(w0);                                    // W(0)
    store_w(0, w0);
    if constexpr (WD == 3) {
      load_w(w1);                                  // W(1), W(2), W(3)
      load_w(w2);
      load_w(w0);
    } else {
      load_w(w0);                                  // W(1)
    }
    }
#pragma unroll
    for (int j = 0; j < AD; ++j) {
      peek_a();
      issue_a(ar[j]);
    }
    if constexpr (RING3) {                         // step 0's first fragments: behind a barrier of their own
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 2 * NP; ++q) (BNEXT_IN_PLACE ? bq[0][q] : bnext[q]) = Wl[grp][0][lane + q * 64];
    }
    OS_STAMP(2);
#ifdef DF3D_OS_QGATHER
    to_operand(ar[0]);
#endif
    for (int s = 0; s < steps; s += AD) {
#pragma unroll
      for (int j = 0; j < AD; j += 3) {
        step(s + j, ar[j], ar[(j + 1) % AD], w1);
        step(s + j + 1, ar[j + 1], ar[(j + 2) % AD], w2);
        step(s + j + 2, ar[j + 2], ar[(j + 3) % AD], w0);
      }
    }
  }
  OS_STAMP(3);
#ifdef DF3D_OS_TRACE
  if (a.trace && (threadIdx.x & 63) == 0) a.trace[((size_t)(blockIdx.x + gridDim.x * blockIdx.y) * 16 + (threadIdx.x >> 6)) * 8 + 5] = steps;
#endif

  if constexpr (KS > 1) {
    // the groups' partial sums meet in LDS (over the weight stages, which nobody reads any more); group 0 goes on alone
    static_assert((KS - 1) * NW * RT * CT * 64 <= KS * 2 * WQ, "the partial sums must fit the weight stages");
    f32x4 *red = (f32x4 *)&Wl[0][0][0];
    __syncthreads();
    if (grp > 0) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) red[((((grp - 1) * NW + wave) * RT + rt) * CT + ct) * 64 + lane] = acc[rt][ct];
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int q = 1; q < KS; ++q)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] += red[((((q - 1) * NW + wave) * RT + rt) * CT + ct) * 64 + lane];
  }
  // ---- epilogue: bias, folded BN, residual, ReLU; optional split rows of the result.  Lane n owns the CT
  //      consecutive columns n*CT .. n*CT+CT-1 of its rows (packed-weight layout 1): 16-byte stores ----
  static_assert(CT == 2 || CT == 4 || CT == 8, "COUT must be 32, 64, 128 or 256");
  if constexpr (CT == 2) {
    // 32 columns per block: lane n owns columns 2n, 2n+1 (8-byte stores; four lanes share a split block)
    const int col = col0 + n * 2;
    const float2 bi = a.bias ? *(const float2 *)(a.bias + col) : make_float2(0.f, 0.f);
    const float2 sc = a.scale ? *(const float2 *)(a.scale + col) : make_float2(1.f, 1.f);
    const float2 sh = a.shift ? *(const float2 *)(a.shift + col) : make_float2(0.f, 0.f);
    const int oc0 = a.cols ? a.cols[2 * blockIdx.y] : 0, ocnt = a.cols ? a.cols[2 * blockIdx.y + 1] : 0;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rowL[wave * WROWS + rt * 16 + 4 * g + r];
        if (row >= a.n_out) continue;
        float2 v = make_float2((acc[rt][0][r] * acc_unscale<NP>() + bi.x) * sc.x + sh.x,
                               (acc[rt][1][r] * acc_unscale<NP>() + bi.y) * sc.y + sh.y);
        if (a.cols) {                        // compact layout: only the block's valid columns exist in the output
          if (a.relu) {
            v.x = fmaxf(v.x, 0.f);
            v.y = fmaxf(v.y, 0.f);
          }
          float *dst = a.out + (size_t)row * a.ldo + oc0 + n * 2;
          if (n * 2 < ocnt) dst[0] = v.x;
          if (n * 2 + 1 < ocnt) dst[1] = v.y;
          continue;
        }
        const size_t o = (size_t)row * a.ldo + col;
        if (a.residual) {
          if constexpr (NP == 1) {
            const unsigned rr = *(const unsigned *)((const char *)a.residual + o * 2);
            v.x += bf16lo_to_f32(rr);
            v.y += bf16hi_to_f32(rr);
          } else {
            const float2 rr = *(const float2 *)(a.residual + o);
            v.x += rr.x;
            v.y += rr.y;
          }
        }
        if (a.relu) {
          v.x = fmaxf(v.x, 0.f);
          v.y = fmaxf(v.y, 0.f);
        }
        if (a.out) *(float2 *)(a.out + o) = v;
        if (a.out_split) {
          if constexpr (NP == 3) {
            unsigned hp, mp, lp;
            split3_pair(v.x, v.y, hp, mp, lp);
            char *blk = (char *)a.out_split + (o >> 3) * 48 + (n & 3) * 4;   // 8-channel block = [hi | mid | lo] 16 B each
            *(unsigned *)blk = hp;
            *(unsigned *)(blk + 16) = mp;
            *(unsigned *)(blk + 32) = lp;
            continue;
          }
          if constexpr (NP == 1) {
            *(unsigned *)((char *)a.out_split + o * 2) = bf16_pair(v.x, v.y);
            continue;
          }
          unsigned hp, lp;
          split_pair(v.x, v.y, hp, lp);
          char *blk = (char *)a.out_split + (o >> 3) * 32 + (n & 3) * 4;     // 8-channel block = [hi 16 B | lo 16 B]
          *(unsigned *)blk = hp;
          *(unsigned *)(blk + 16) = lp;
        }
      }
    }
    OS_STAMP(4);
    return;
  } else {
  f32x4 bi[CT / 4], sc[CT / 4], sh[CT / 4];
#pragma unroll
  for (int q = 0; q < CT / 4; ++q) {
    const int col = col0 + n * CT + q * 4;
    bi[q] = a.bias ? *(const f32x4 *)(a.bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
    sc[q] = a.scale ? *(const f32x4 *)(a.scale + col) : (f32x4){1.f, 1.f, 1.f, 1.f};
    sh[q] = a.shift ? *(const f32x4 *)(a.shift + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rowL[wave * WROWS + rt * 16 + 4 * g + r];
      if (row >= a.n_out) continue;
      if (OS_DBG(32) && acc[rt][0][r] != 1234.567f) continue;      // experiment: no output stores
      const size_t o = (size_t)row * a.ldo + col0 + n * CT;
      unsigned h[CT / 2], l[CT / 2], m3[CT / 2];         // packed pairs
#pragma unroll
      for (int q = 0; q < CT / 4; ++q) {
        f32x4 v = (f32x4){acc[rt][q * 4][r], acc[rt][q * 4 + 1][r], acc[rt][q * 4 + 2][r], acc[rt][q * 4 + 3][r]};
        v = (v * acc_unscale<NP>() + bi[q]) * sc[q] + sh[q];
        if (a.residual) {
          if constexpr (NP == 1) {
            const u32x2 rr = *(const u32x2 *)((const char *)a.residual + (o + q * 4) * 2);
            v += (f32x4){bf16lo_to_f32(rr[0]), bf16hi_to_f32(rr[0]), bf16lo_to_f32(rr[1]), bf16hi_to_f32(rr[1])};
          } else {
            v += *(const f32x4 *)(a.residual + o + q * 4);
          }
        }
        if (a.relu) {
          v[0] = fmaxf(v[0], 0.f);
          v[1] = fmaxf(v[1], 0.f);
          v[2] = fmaxf(v[2], 0.f);
          v[3] = fmaxf(v[3], 0.f);
        }
        if (a.out) *(f32x4 *)(a.out + o + q * 4) = v;
        if (a.out_split) {
          if constexpr (NP == 3) {
            split3_pair(v[0], v[1], h[q * 2], m3[q * 2], l[q * 2]);
            split3_pair(v[2], v[3], h[q * 2 + 1], m3[q * 2 + 1], l[q * 2 + 1]);
          } else if constexpr (NP == 1) {
            h[q * 2] = bf16_pair(v[0], v[1]);
            h[q * 2 + 1] = bf16_pair(v[2], v[3]);
          } else {
            split_pair(v[0], v[1], h[q * 2], l[q * 2]);
            split_pair(v[2], v[3], h[q * 2 + 1], l[q * 2 + 1]);
          }
        }
      }
      if constexpr (NP == 3) {
        if (a.out_split) {
          char *blk = (char *)a.out_split + (o >> 3) * 48;            // 8-channel block = [hi | mid | lo] 16 B each
          if constexpr (CT == 8) {
            *(u32x4 *)blk = (u32x4){h[0], h[1], h[2], h[3]};
            *(u32x4 *)(blk + 16) = (u32x4){m3[0], m3[1], m3[2], m3[3]};
            *(u32x4 *)(blk + 32) = (u32x4){l[0], l[1], l[2], l[3]};
          } else {
            blk += (n & 1) * 8;
            *(u32x2 *)blk = (u32x2){h[0], h[1]};
            *(u32x2 *)(blk + 16) = (u32x2){m3[0], m3[1]};
            *(u32x2 *)(blk + 32) = (u32x2){l[0], l[1]};
          }
        }
        continue;
      }
      if (a.out_split && NP == 1) {                                   // bf16 rows: this lane's CT columns
        char *dst = (char *)a.out_split + o * 2;
        if constexpr (CT == 8) *(u32x4 *)dst = (u32x4){h[0], h[1], h[2], h[3]};
        else *(u32x2 *)dst = (u32x2){h[0], h[1]};
      } else if (a.out_split) {
        char *blk = (char *)a.out_split + (o >> 3) * 32;             // 8-channel block = [hi 16 B | lo 16 B]
        if constexpr (CT == 8) {
          *(u32x4 *)blk = (u32x4){h[0], h[1], h[2], h[3]};
          *(u32x4 *)(blk + 16) = (u32x4){l[0], l[1], l[2], l[3]};
        } else {
          blk += (n & 1) * 8;                                        // two lanes share a block
          *(u32x2 *)blk = (u32x2){h[0], h[1]};
          *(u32x2 *)(blk + 16) = (u32x2){l[0], l[1]};
        }
      }
    }
  }
  OS_STAMP(4);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Output-stationary kernel, second structure: four matrix waves and eight loader waves per workgroup, both operands by
// LDS-DMA in whole cache lines.
//
// What the measurements of the kernel above and of two intermediate structures (a two-group LDS-DMA ping-pong with and
// without "helper" DMAs from the group in its matrix phase, a 256-row tile with the offsets split over workgroups) say
// (tools/ubench/cu_ingest.hip, consumer_loop.hip, lc_trace.py, ablate_probe.py on MI355X):
//   * fragment-shaped register gathers cost 64 line requests per wave instruction; an LDS-DMA of eight whole 128-byte
//     lines (8 rows x one 32-channel block, hi | lo) costs one;
//   * one wave gets an LDS-DMA instruction of 1 KB accepted only every 110-180 clocks (6-9 B/clk), a compute unit as a
//     whole takes 47-50 B/clk from L2 once eight or more waves issue (16 B/clk for lines from the Infinity Cache,
//     12 B/clk from HBM); a 128 x 128 x 32 step needs 32 KB per 768 MFMA clocks = 42 B/clk;
//   * in a ping-pong of two wave groups every phase begins with an exposed LDS round trip (~300 clocks before the first
//     MFMA), the DMA issue stalls of a wave sit in ITS instruction stream in front of its own next MFMAs, and MFMA time,
//     LDS start-up and DMA issue time simply add up (34 % + ~15 % + ~30 % of the kernel);
//   * splitting the offsets of a tile over workgroups needs device-scope fences per workgroup: slower than the idle CUs
//     it fills (128 -> 128 K = 27, 29k rows: 158 us without, 176 us with the split).
// Here the roles never change: waves 0-3 (one per SIMD) own 32 rows x all columns of the block each and do nothing but
// fragment reads and MFMAs, waves 4-11 do nothing but address resolution and LDS-DMA (two row pieces + one or two
// filter pieces per step each).  The operands go through a ring of four stages; one s_barrier per step.  At the barrier
// that ends step s the loaders guarantee that the stages of steps <= s + 2 are complete (each has waited for its own
// pieces) and the matrix waves that they have left the stage of step s, which the loaders then refill with step s + 4.
// A matrix wave therefore reads the first fragments of step s + 1 BEFORE that barrier, behind the MFMAs of the last
// batches of step s, and issues MFMAs back to back across steps.  Its fragment reads are inline asm on purpose: the
// compiler puts a full s_waitcnt vmcnt(0) in front of every LDS read it can see that may alias an LDS-DMA destination,
// and it clusters the reads of a batch -- four ds_read_b128 in a row hold the wave's issue slot for ~32 clocks in which
// the matrix pipe of its SIMD runs dry (consumer_loop.hip: 460 -> 386 ns per step).  Every read is issued alone in an
// MFMA gap, one batch ahead of its use, so the only wait is a full lgkmcnt(0) at a batch boundary.
template <int CIN, int CW>
__global__ __launch_bounds__(768) void spconv_os_lc_kernel(SplitConvArgs a) {
  constexpr int NP = 2, TM = 128, KB = CIN / 32, NS = 4;
  constexpr int RT = 2, CT = CW / 16;             // per matrix wave: 2 row tiles x CT column tiles of 16 x 16
  static_assert(CW == 128, "column block (a 64-column variant of this structure was measured: 24 KB per 384 MFMA clocks "
                           "makes the loaders the pole, 64 -> 64 K = 27 on 66k rows 85 us against 66 us of the kernel above)");
  constexpr int WQ = CT * NP * 64;                // u32x4 per (offset, 32-channel block) filter tile
  constexpr int AQ = TM * 8;                      // u32x4 per A stage: 128 rows x 128 B
  constexpr int NLOAD = 8, APL = 16 / NLOAD, WPL = (WQ / 64) / NLOAD;      // row / filter pieces per loader wave and step
  __shared__ u32x4 Al[NS][AQ];
  __shared__ u32x4 Wl[NS][WQ];
  __shared__ int nbrL[DF3D_MAX_KVOL][TM];
  __shared__ int rowL[TM];
  __shared__ unsigned wg_mask;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);                 // consecutive tiles stay on one XCD / L2
  const int row0 = tile * TM;
  // Round 3: a workgroup walks `cbw` column blocks of its row tile without leaving the step pipeline -- the neighbour
  // table, the ring's fill and the wait for the last stores at the end of the workgroup are paid once per row tile
  // instead of once per (row tile, block) (head middle convolutions, 18 blocks of 18 steps: prologue 7 % + epilogue 23 %
  // of a workgroup's life, one workgroup per CU so nothing overlapped them: tools/ubench/lc_trace_head.py).
  const int cbw = a.cbw > 0 ? a.cbw : 1;
  const int cb0 = blockIdx.y * cbw;
  const int ncb = min(cbw, a.gy - cb0);
  OS_STAMP(6);

  if (tid == 0) wg_mask = 0u;
  __syncthreads();
  {
    constexpr int KSTEP = 768 / TM;              // offsets covered per pass
    constexpr int NPASS = (DF3D_MAX_KVOL + KSTEP - 1) / KSTEP;
    const int r = tid % TM, k0 = tid / TM;
    int row = row0 + r;
    row = row < a.n_out ? (a.order ? a.order[row] : row) : a.n_out;
    if (k0 == 0) rowL[r] = row;
    unsigned mine = 0u;
    int v[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      int k = k0 + i * KSTEP;
      v[i] = (k < a.K && row < a.n_out) ? a.nbr[(size_t)k * a.n_out + row] : -1;
    }
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      int k = k0 + i * KSTEP;
      if (k < a.K) nbrL[k][r] = v[i];
      if (__ballot(v[i] >= 0) != 0ull) mine |= 1u << (k & 31);       // TM >= 64: the whole wave looks at one offset
    }
    if (lane == 0 && mine) atomicOr(&wg_mask, mine);
  }
  __syncthreads();
  const unsigned gmask = __builtin_amdgcn_readfirstlane(wg_mask);
  const int steps = __popc(gmask) * KB;             // per column block
  const int total = steps * ncb;
  // XOR swizzle of a row's eight 16-byte units, by the row's position in its 16-row MFMA tile (applied to the SOURCE
  // address of the DMA: an LDS-DMA writes lane-linearly): makes the four lane groups of a ds_read_b128 (rows n, units
  // g*2 + q) hit 16 distinct bank columns (brute-forced; linear over GF(2))
  auto swz = [](int nn) { return ((nn >> 2) & 1) | (((nn >> 1) & 1) << 2); };
  OS_STAMP(0);
#ifdef DF3D_OS_TRACE
  unsigned long long t_a = 0, t_b = 0, t_c = 0, t_x = 0;
#define LC_T0() t_x = __builtin_amdgcn_s_memtime()
#define LC_ACC(v) do { unsigned long long t_y = __builtin_amdgcn_s_memtime(); v += t_y - t_x; t_x = t_y; } while (0)
#define LC_DUMP()                                                                                                   \
  do {                                                                                                              \
    if (a.trace && lane == 0) {                                                                                     \
      unsigned long long *t_ = a.trace + ((size_t)(blockIdx.x + gridDim.x * blockIdx.y) * 16 + wave) * 8;           \
      t_[1] = t_a, t_[2] = t_b, t_[3] = t_c, t_[5] = steps;                                                         \
    }                                                                                                               \
  } while (0)
#else
#define LC_T0() do { } while (0)
#define LC_ACC(v) do { } while (0)
#define LC_DUMP() do { } while (0)
#endif

  if (wave >= 4) {
    // ------------------------------------------------ loader waves ------------------------------------------------
    // (they are the pole: above the matrix waves in priority)
    __builtin_amdgcn_s_setprio(3);
    const int lw = wave - 4;
    StepCursor cu;                                // the step that is fetched next
    cu.init(gmask);
    // The channel blocks of one offset are consecutive steps and gather the SAME rows: the neighbour-table lookups and
    // the address arithmetic happen once per offset (`abase`, per-lane block stride 0 for the all-zero row), a step only
    // adds its block offset.
    const u32x4 *abase[APL], *wbase = nullptr;
    int astep[APL];
    int cbl = cb0;                                // column block of the step that is fetched next
    auto issue = [&](int t) {
      const int st = t & (NS - 1);
      if (cu.kb == 0) {
        int idx[APL];
#pragma unroll
        for (int i = 0; i < APL; ++i) idx[i] = nbrL[cu.k][(lw * APL + i) * 8 + (lane >> 3)];
#pragma unroll
        for (int i = 0; i < APL; ++i) {
          const int r = (lw * APL + i) * 8 + (lane >> 3);
          const int unit = (lane & 7) ^ swz(r & 15);
          abase[i] = idx[i] >= 0 ? a.feat + (size_t)idx[i] * a.ldi + cbl * a.in_goff + unit : g_zero_row + (lane & 7);
          astep[i] = idx[i] >= 0 ? 8 : 0;
        }
        wbase = a.w + ((size_t)cbl * a.K * KB + (size_t)cu.k * KB) * WQ + (lw * WPL) * 64 + lane;
      }
      if (!OS_DBG(64)) {                          // (experiment: no DMAs)
#pragma unroll
        for (int i = 0; i < APL; ++i)
          __builtin_amdgcn_global_load_lds(abase[i] + cu.kb * astep[i],
                                           (__attribute__((address_space(3))) void *)&Al[st][(lw * APL + i) * 64], 16, 0, 0);
#pragma unroll
        for (int i = 0; i < WPL; ++i)
          __builtin_amdgcn_global_load_lds(wbase + (size_t)cu.kb * WQ + i * 64,
                                           (__attribute__((address_space(3))) void *)&Wl[st][(lw * WPL + i) * 64], 16, 0, 0);
      }
      cu.template next<KB>();
      if (!cu.live) {                             // the block's last step: the same offsets again, next block's filters
        ++cbl;
        cu.init(gmask);
      }
    };
    constexpr int PPS = APL + WPL;                // pieces per loader wave and step
    LC_T0();
    for (int t = 0; t < NS && t < total; ++t) issue(t);
    LC_ACC(t_a);
    for (int s = -1; s < total; ++s) {
      // steps <= s + 2 have landed; the pieces of step s + 3 (the newest issued) may still be in flight
      if (s + 3 < total) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PPS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      LC_ACC(t_b);
      asm volatile("s_barrier" ::: "memory");
      LC_ACC(t_c);
      if (s >= 0 && s + NS < total) issue(s + NS);                   // into the stage step s has just left
      LC_ACC(t_a);
    }
    LC_DUMP();
    return;
  }

  // -------------------------------------------------- matrix waves --------------------------------------------------
  __builtin_amdgcn_s_setprio(1);
  const int wr = wave;                            // rows wr * 32 .. + 31
  f32x4 acc[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const unsigned a_lds = lds_addr(&Al[0][0]) + (unsigned)((wr * 32 + n) * 8) * 16u;
  const unsigned a_addr0 = a_lds + (unsigned)((g * 2) ^ swz(n)) * 16u;
  const unsigned a_addr1 = a_lds + (unsigned)((g * 2 + 1) ^ swz(n)) * 16u;
  const unsigned w_addr = lds_addr(&Wl[0][0]) + (unsigned)lane * 16u;
#define LC_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define LC_SB() __builtin_amdgcn_sched_barrier(0)
#define LC_M(rt, c, ap, br, buf, slot) acc[rt][c] = DF3D_MFMA_F16(af[buf][rt][ap], bq[slot][br], acc[rt][c])
  // Batch i of a step = the two column tiles 2i, 2i + 1 x both row tiles x the three products (lo*hi, hi*lo, hi*hi:
  // the summation order of the kernel above): 12 MFMAs on B slot `slot`; RA .. RH = single fragment reads issued behind
  // MFMAs 0 .. 7
#define LC_BATCH(i, buf, slot, RA, RB, RC, RD, RE, RF, RG, RH)                    \
  do {                                                                            \
    if (OS_DBG(4)) break;                                                         \
    LC_M(0, (i) * 2, 1, 0, buf, slot); LC_SB(); RA; LC_SB();                      \
    LC_M(0, (i) * 2 + 1, 1, 2, buf, slot); LC_SB(); RB; LC_SB();                  \
    LC_M(1, (i) * 2, 1, 0, buf, slot); LC_SB(); RC; LC_SB();                      \
    LC_M(1, (i) * 2 + 1, 1, 2, buf, slot); LC_SB(); RD; LC_SB();                  \
    LC_M(0, (i) * 2, 0, 1, buf, slot); LC_SB(); RE; LC_SB();                      \
    LC_M(0, (i) * 2 + 1, 0, 3, buf, slot); LC_SB(); RF; LC_SB();                  \
    LC_M(1, (i) * 2, 0, 1, buf, slot); LC_SB(); RG; LC_SB();                      \
    LC_M(1, (i) * 2 + 1, 0, 3, buf, slot); LC_SB(); RH; LC_SB();                  \
    LC_M(0, (i) * 2, 0, 0, buf, slot);                                            \
    LC_M(0, (i) * 2 + 1, 0, 2, buf, slot);                                        \
    LC_M(1, (i) * 2, 0, 0, buf, slot);                                            \
    LC_M(1, (i) * 2 + 1, 0, 2, buf, slot);                                        \
    LC_SB();                                                                      \
  } while (0)
#define LC_BATCH_X(...) LC_BATCH(__VA_ARGS__)       /* expands LC_READ_B4 / LC_READ_A4 into four arguments first */
#define LC_WAITALL(buf)                                                                                                \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                                  \
               : "+v"(af[buf][0][0]), "+v"(af[buf][0][1]), "+v"(af[buf][1][0]), "+v"(af[buf][1][1]), "+v"(bq[0][0]),   \
                 "+v"(bq[0][1]), "+v"(bq[0][2]), "+v"(bq[0][3]), "+v"(bq[1][0]), "+v"(bq[1][1]), "+v"(bq[1][2]), "+v"(bq[1][3]))
#define LC_NONE do { } while (0)
#ifdef DF3D_LC_NOB   /* experiment: no B-fragment reads (64 of the 80 KB a step reads from LDS) */
#define LC_READ_B4(slot, wa, i) LC_NONE, LC_NONE, LC_NONE, LC_NONE
#else
#define LC_READ_B4(slot, wa, i)  /* the four fragments of batch i as statements RA .. RD */                            \
  LC_READ(bq[slot][0], wa, ((i) * 4 + 0) * 1024), LC_READ(bq[slot][1], wa, ((i) * 4 + 1) * 1024),                      \
      LC_READ(bq[slot][2], wa, ((i) * 4 + 2) * 1024), LC_READ(bq[slot][3], wa, ((i) * 4 + 3) * 1024)
#endif
#define LC_READ_A4(buf, aa0, aa1)                                                                                      \
  LC_READ(af[buf][0][0], aa0, 0), LC_READ(af[buf][0][1], aa1, 0), LC_READ(af[buf][1][0], aa0, 16 * 128),               \
      LC_READ(af[buf][1][1], aa1, 16 * 128)
  // one step: on entry the reads of its A fragments (buffer `buf`) and of B batch 0 (slot 0) are in flight; on exit the
  // same holds for step s + 1 (buffer buf ^ 1) -- its stage was complete at the previous barrier already (behind the
  // last step the same reads fetch a stale stage and nobody uses them: no branch in the loop body)
#define LC_STEP(s, buf)                                                                                                \
  do {                                                                                                                 \
    const unsigned wt = w_addr + (unsigned)((s) & (NS - 1)) * (WQ * 16u);                                              \
    const unsigned wn = w_addr + (unsigned)(((s) + 1) & (NS - 1)) * (WQ * 16u);                                        \
    const unsigned an0 = a_addr0 + (unsigned)(((s) + 1) & (NS - 1)) * (AQ * 16u);                                      \
    const unsigned an1 = a_addr1 + (unsigned)(((s) + 1) & (NS - 1)) * (AQ * 16u);                                      \
    LC_WAITALL(buf);                                                                                                   \
    LC_BATCH_X(0, buf, 0, LC_READ_B4(1, wt, 1), LC_NONE, LC_NONE, LC_NONE, LC_NONE);                                   \
    LC_WAITALL(buf);                                                                                                   \
    LC_BATCH_X(1, buf, 1, LC_READ_B4(0, wt, 2), LC_NONE, LC_NONE, LC_NONE, LC_NONE);                                   \
    LC_WAITALL(buf);                                                                                                   \
    LC_BATCH_X(2, buf, 0, LC_READ_B4(1, wt, 3), LC_READ_A4((buf) ^ 1, an0, an1));                                      \
    LC_WAITALL(buf);                                                                                                   \
    LC_BATCH_X(3, buf, 1, LC_READ_B4(0, wn, 0), LC_NONE, LC_NONE, LC_NONE, LC_NONE);                                   \
    LC_ACC(t_a);                                                                                                       \
    asm volatile("s_barrier" ::: "memory");                                                                            \
    LC_ACC(t_c);                                                                                                       \
  } while (0)

  // ---- epilogue of one column block: bias, folded BN, residual, ReLU; optional split rows of the result.  Lane n owns the
  //      CT consecutive columns n * CT .. n * CT + CT - 1 (packed-weight layout 1).  Leaves the accumulators at zero.
  //      Round 3: the arguments live in locals (the step loop's asm memory clobbers made every row re-load them from the
  //      kernel arguments) and the per-column vectors pass through an asm after their one wait -- the compiler could not
  //      prove those loads complete at the loop joins and put an s_waitcnt vmcnt(0) in front of EVERY row, i.e. it also
  //      waited for the previous row's stores: eight store round trips in a row, ~8000 clocks of a one-workgroup-per-CU
  //      kernel (lc_trace_head.py).  Now the stores of a block queue up behind each other ----
  const float *e_res = a.residual;
  float *e_out = a.out;
  char *e_split = (char *)a.out_split;
  const float *e_bias = a.bias, *e_scale = a.scale, *e_shift = a.shift;
  const int e_relu = a.relu, e_ldo = a.ldo, e_nout = a.n_out;
  static_assert(CT == 8, "two f32x4 per lane and vector below");
  float e_amax = 0.f;                            // largest |value| written as split rows (range check, once per workgroup)
  auto epilogue = [&](int cb) {
    const int col0 = cb * CW;
    f32x4 bi[2], sc[2], sh[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int col = col0 + n * CT + q * 4;
      bi[q] = e_bias ? *(const f32x4 *)(e_bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
      sc[q] = e_scale ? *(const f32x4 *)(e_scale + col) : (f32x4){1.f, 1.f, 1.f, 1.f};
      sh[q] = e_shift ? *(const f32x4 *)(e_shift + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bi[0]), "+v"(bi[1]), "+v"(sc[0]), "+v"(sc[1]), "+v"(sh[0]), "+v"(sh[1]));
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rowL[wr * 32 + rt * 16 + 4 * g + r];
        if (row >= e_nout) continue;
        const size_t o = (size_t)row * e_ldo + col0 + n * CT;
        unsigned hh[CT / 2], ll[CT / 2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          f32x4 v = (f32x4){acc[rt][q * 4][r], acc[rt][q * 4 + 1][r], acc[rt][q * 4 + 2][r], acc[rt][q * 4 + 3][r]};
          v = (v * DF3D_ACC_UNSCALE + bi[q]) * sc[q] + sh[q];
          if (e_res) v += *(const f32x4 *)(e_res + o + q * 4);
          if (e_relu) {
            v[0] = fmaxf(v[0], 0.f);
            v[1] = fmaxf(v[1], 0.f);
            v[2] = fmaxf(v[2], 0.f);
            v[3] = fmaxf(v[3], 0.f);
          }
          if (e_out) *(f32x4 *)(e_out + o + q * 4) = v;
          if (e_split) {
            split_pair_acc(v[0], v[1], hh[q * 2], ll[q * 2], e_amax);
            split_pair_acc(v[2], v[3], hh[q * 2 + 1], ll[q * 2 + 1], e_amax);
          }
        }
        if (e_split) {
          char *blk = e_split + (o >> 3) * 32;                       // 8-channel block = [hi 16 B | lo 16 B]
          *(u32x4 *)blk = (u32x4){hh[0], hh[1], hh[2], hh[3]};
          *(u32x4 *)(blk + 16) = (u32x4){ll[0], ll[1], ll[2], ll[3]};
        }
      }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };

  u32x4 af[2][RT][NP];
  u32x4 bq[2][2 * NP];
  asm volatile("s_barrier" ::: "memory");         // steps 0 and 1 are in their stages
  {
    LC_READ(af[0][0][0], a_addr0, 0);
    LC_READ(af[0][0][1], a_addr1, 0);
    LC_READ(af[0][1][0], a_addr0, 16 * 128);
    LC_READ(af[0][1][1], a_addr1, 16 * 128);
    LC_READ(bq[0][0], w_addr, 0 * 1024);
    LC_READ(bq[0][1], w_addr, 1 * 1024);
    LC_READ(bq[0][2], w_addr, 2 * 1024);
    LC_READ(bq[0][3], w_addr, 3 * 1024);
  }
  LC_T0();
  // behind a block's last step: the fragment reads of the next step are in flight.  They land, the epilogue runs without
  // them (they would cost it 32 live registers: spills), and the same reads are issued again -- the stage of step s + 1
  // is not refilled before the barrier that ends it.  The epilogue's stores drain behind the next block's MFMAs.
  static_assert(KB % 2 == 0, "a block's steps come in pairs: the fragment buffers restart at 0 with every block");
  int t = 0;
  for (int cb = cb0; cb < cb0 + ncb && steps > 0; ++cb) {
    for (int s = 0; s < steps; s += 2, t += 2) {
      LC_STEP(t, 0);
      LC_STEP(t + 1, 1);
    }
    LC_WAITALL(0);
    __builtin_amdgcn_s_setprio(0);
    epilogue(cb);
    __builtin_amdgcn_s_setprio(1);
    if (cb + 1 < cb0 + ncb) {
      const unsigned rn0 = a_addr0 + (unsigned)(t & (NS - 1)) * (AQ * 16u);
      const unsigned rn1 = a_addr1 + (unsigned)(t & (NS - 1)) * (AQ * 16u);
      const unsigned rwn = w_addr + (unsigned)(t & (NS - 1)) * (WQ * 16u);
      LC_READ(af[0][0][0], rn0, 0);
      LC_READ(af[0][0][1], rn1, 0);
      LC_READ(af[0][1][0], rn0, 16 * 128);
      LC_READ(af[0][1][1], rn1, 16 * 128);
      LC_READ(bq[0][0], rwn, 0 * 1024);
      LC_READ(bq[0][1], rwn, 1 * 1024);
      LC_READ(bq[0][2], rwn, 2 * 1024);
      LC_READ(bq[0][3], rwn, 3 * 1024);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (steps == 0)                                // a tile without a single neighbour: bias / shift rows
    for (int cb = cb0; cb < cb0 + ncb; ++cb) epilogue(cb);
  split_range_flag(e_amax);
#undef LC_READ
#undef LC_SB
#undef LC_M
#undef LC_BATCH
#undef LC_BATCH_X
#undef LC_WAITALL
#undef LC_NONE
#undef LC_READ_B4
#undef LC_READ_A4
#undef LC_STEP
  __builtin_amdgcn_s_setprio(0);
  LC_DUMP();
#undef LC_T0
#undef LC_ACC
#undef LC_DUMP
  OS_STAMP(4);

  OS_STAMP(7);
}

static int g_num_cu = 0;
static int num_cu() {
  if (!g_num_cu) {
    hipDeviceProp_t p;
    g_num_cu = (hipGetDeviceProperties(&p, 0) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
  }
  return g_num_cu;
}

// The loader / consumer kernel serves 128-column blocks from ~190 workgroups on (below that the 64-row tiles of the
// kernel above fill the chip better); DF3D_OS_LC=0 / 1 forces it off / on.
template <int CIN>
static int launch_os_lc(const SplitConvArgs &a_, hipStream_t stream) {
  SplitConvArgs a = a_;
  // column blocks per workgroup: the split into ny workgroups per row tile with the fewest (rounds over the CUs) x (blocks
  // walked + ~half a block of prologue and drain per workgroup); DF3D_LC_CBW overrides (tuning aid)
  const int tiles = cdiv(a.n_out, 128), ncu = num_cu();
  int best_ny = a.gy;
  double best = 1e30;
  for (int ny = 1; ny <= a.gy; ++ny) {
    const double cost = (double)cdiv((long long)tiles * ny, ncu) * (cdiv(a.gy, ny) + 0.5);
    if (cost < best - 1e-9) best = cost, best_ny = ny;
  }
  a.cbw = cdiv(a.gy, best_ny);
  static const char *force = getenv("DF3D_LC_CBW");
  if (force && atoi(force) > 0) a.cbw = atoi(force) < a.gy ? atoi(force) : a.gy;
  hipLaunchKernelGGL((spconv_os_lc_kernel<CIN, 128>), dim3(tiles, cdiv(a.gy, a.cbw)), dim3(768), 0, stream, a);
  return DF3D_OK;
}

static bool use_lc(const SplitConvArgs &a) {
  const char *t = getenv("DF3D_OS_LC");         // read per call: the tests switch it inside one process
  if (a.cols) return false;
  if (t && t[0] == '0') return false;
  if (t && t[0] == '1') return true;
  // (one offset = a linear layer over rows: the ring's fill and drain would be most of a two-step tile)
  return a.K > 1 && (long long)cdiv(a.n_out, 128) * a.gy >= 190;
}

// (Round 3's staged-range and weight-stationary experiment kernels -- measured, parity-tested, never faster than the kernels
// here -- moved out of the library in round 5: tools/ubench/attic/spconv_halo.h, spconv_ws.h)

template <int CIN, int COUT>
static int launch_os_split(const SplitConvArgs &a, hipStream_t stream) {
  // measured on MI355X (tools/conv_probe.py, DF3D_OS_CFG sweep): 8 waves x 1 row tile wins at the nuScenes layer
  // sizes (30k-70k rows).  The W step tiles are re-read from L2 by every workgroup (n_out/TM x 4*K*CIN*COUT
  // bytes per launch, more than the gathers), so more rows per workgroup = less L2 traffic; bigger register
  // tiles (RT 2) leave too few waves on the 256 CUs at these row counts.
  int rt = 1, nw = a.n_out >= 16 * 1024 ? 8 : 2, kps = 1;
  if constexpr (COUT == 128 && CIN % 32 == 0) {
    if (use_lc(a)) return launch_os_lc<CIN>(a, stream);
  }
  static const char *cfg = getenv("DF3D_OS_CFG");       // tuning aid: "RT,NW[,KPS]"
  if (cfg && cfg[0] && cfg[1] == ',') {
    rt = cfg[0] - '0';
    nw = atoi(cfg + 2);
    const char *c2 = strchr(cfg + 2, ',');
    if (c2) kps = atoi(c2 + 1);
  }
  constexpr int KMAX = CIN >= 64 ? 2 : 1;
  if (kps > KMAX) kps = KMAX;
#define DF3D_OS_LAUNCH(RT, NW, KPS)                                                                      \
  hipLaunchKernelGGL((spconv_os_split_kernel<CIN, COUT, RT, NW, KPS>), dim3(cdiv(a.n_out, 16 * RT * NW), a.gy), \
                     dim3(NW * 64), 0, stream, a)
  if (kps == 2) {
    if (rt == 2) DF3D_OS_LAUNCH(2, 4, KMAX);
    else if (nw == 2) DF3D_OS_LAUNCH(1, 2, KMAX);
    else if (nw == 8) DF3D_OS_LAUNCH(1, 8, KMAX);
    else DF3D_OS_LAUNCH(1, 4, KMAX);
  } else if (nw == 16) DF3D_OS_LAUNCH(1, 16, 1);
  else if (rt == 2 && nw == 8) DF3D_OS_LAUNCH(2, 8, 1);
  else if (nw == 8) {
    // sparse 3-D rulebooks: masked gathers (DF3D_OS_MASKED=0 / 1 forces them off / on; read per call for the A/B)
    const char *mg = getenv("DF3D_OS_MASKED");
    // measured (tools/conv_probe.py, MI355X): 32 -> 32 35.3 -> 33.5 us, 64 -> 64 63.5 -> 68.5 us (the branch costs the wider
    // kernel more than the requests it saves) -- default on for 32 input channels only
    const bool masked = mg ? mg[0] == '1' : (a.K == 27 && !a.cols && CIN == 32);
    if (masked)
      hipLaunchKernelGGL((spconv_os_split_kernel<CIN, COUT, 1, 8, 1, 2, 1, true>), dim3(cdiv(a.n_out, 128), a.gy), dim3(512), 0,
                         stream, a);
    else DF3D_OS_LAUNCH(1, 8, 1);
  }
  else if (rt == 2 && nw == 4) DF3D_OS_LAUNCH(2, 4, 1);
  else if (rt == 2 && nw == 2) DF3D_OS_LAUNCH(2, 2, 1);
  else if (rt == 1 && nw == 4) DF3D_OS_LAUNCH(1, 4, 1);
  else DF3D_OS_LAUNCH(1, 2, 1);
#undef DF3D_OS_LAUNCH
  return DF3D_OK;
}

// 256-channel shapes (dense BEV neck).  Workgroup rows by map size so that the small (90 x 90) maps still fill the
// 256 CUs; DF3D_OS_WIDE="RT,NW" overrides (tuning aid).
template <int CIN, int COUT>
static int launch_os_split_wide(const SplitConvArgs &a, hipStream_t stream) {
  const int CS = a.gy;
  if constexpr (COUT % 128 == 0) {
    if (use_lc(a)) return launch_os_lc<CIN>(a, stream);
  }
  int rt = 1, nw = (long long)a.n_out * CS >= 128 * 384 ? 8 : (long long)a.n_out * CS >= 64 * 192 ? 4 : 2;
  static const char *cfg = getenv("DF3D_OS_WIDE");
  if (cfg && cfg[0] && cfg[1] == ',') {
    rt = cfg[0] - '0';
    nw = atoi(cfg + 2);
  }
  // Small maps (B = 1 at 90 x 90: 127 row tiles x 2 column halves of 4 waves = one wave per SIMD): three wave groups per
  // workgroup share the offsets of the tile (KS = 3).  DF3D_OS_KSPLIT=0 turns it off (A/B; read per call).
  if constexpr (CIN <= 256 && COUT % 128 == 0) {
    const char *ks = getenv("DF3D_OS_KSPLIT");
    const long long waves = (long long)cdiv(a.n_out, 64) * CS * 4;
    if (!(cfg && cfg[0]) && nw == 4 && a.K >= 3 && !(ks && ks[0] == '0') && waves <= 6LL * num_cu()) {
      hipLaunchKernelGGL((spconv_os_split_kernel<CIN, COUT, 1, 4, 1, 2, 3>), dim3(cdiv(a.n_out, 64), CS), dim3(768), 0,
                         stream, a);
      return DF3D_OK;
    }
  }
#define DF3D_OS_LAUNCH(RT, NW)                                                                                  \
  hipLaunchKernelGGL((spconv_os_split_kernel<CIN, COUT, RT, NW, 1>), dim3(cdiv(a.n_out, 16 * RT * NW), CS), \
                     dim3(NW * 64), 0, stream, a)
  if (rt == 2 && nw == 4) DF3D_OS_LAUNCH(2, 4);
  else if (rt == 2) DF3D_OS_LAUNCH(2, 2);
  else if (nw == 8) DF3D_OS_LAUNCH(1, 8);
  else if (nw == 4) DF3D_OS_LAUNCH(1, 4);
  else DF3D_OS_LAUNCH(1, 2);
#undef DF3D_OS_LAUNCH
  return DF3D_OK;
}


template <int CIN, int COUT>
static int launch_split(const SplitConvArgs &a, const int32_t *tile_rows, int ntiles, hipStream_t stream) {
  const size_t per_row = (size_t)COUT * 4 + (size_t)a.K * 4 + (size_t)a.K * 2;
  int tm_max = (int)((144 * 1024) / per_row) & ~3;
  if (tm_max > 4095) tm_max = 4092;                    // chunk-group index has 8 bits: (TM/16) < 256
  int slots = num_cu();
  int m = cdiv(a.n_out, (long long)slots * tm_max);
  int TM = (cdiv(a.n_out, (long long)slots * m) + 3) & ~3;
  if (TM < 16) TM = 16;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute((const void *)spconv_split_kernel<CIN, COUT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS) failed: %s", hipGetErrorString(e));
      return DF3D_EHIP;
    }
    configured = true;
  }
  int nt = cdiv(a.n_out, TM);
  if (tile_rows && ntiles > 0) {
    TM = tm_max;
    nt = ntiles;
  } else {
    tile_rows = nullptr;
  }
  size_t lds = (size_t)(TM + 1) * COUT * 4 + (size_t)TM * a.K * 6 + ((size_t)a.K * (TM / 16 + 2) + 8) * 4 + 64;
  hipLaunchKernelGGL((spconv_split_kernel<CIN, COUT>), dim3(nt), dim3(256), lds, stream, a, TM, tile_rows);
  return DF3D_OK;
}

// which kernel serves a shape: 1 = output-stationary (layout 1), 0 = pair-compacted (layout 0)
static int split_layout(int cin, int cout) {
  static const char *force = getenv("DF3D_SPLIT_KERNEL");
  if (force && force[0] == 'p') return 0;
  if (force && force[0] == 'o') return 1;
  (void)cin;
  (void)cout;
  return 1;
}

static bool split_shape_wide(int cin, int cout) {      // served by the output-stationary kernel only
  return (cin == 256 && (cout == 128 || cout == 256)) || (cin == 128 && cout == 256);
}

static bool split_shape_head(int cin, int cout) {      // detection heads: shared conv 512 -> 64 | 128, final convs -> <= 32
  return (cin == 512 && (cout == 64 || cout == 128)) || ((cin == 64 || cin == 128) && cout == 32) ||
         (cin == 128 && cout == 64);                       // + the second linear of a 64-channel transformer FFN
}

// output-stationary launch of any served (cin, cout per column block) pair
static int launch_os_any(int cin, int cout, const SplitConvArgs &a, hipStream_t stream);

static bool split_shape_ok(int cin, int cout) {
  return (cout == 128 && (cin == 128 || cin == 64)) || (cout == 64 && (cin == 64 || cin == 32)) ||
         ((split_shape_wide(cin, cout) || split_shape_head(cin, cout) || (cout == 32 && cin == 32)) &&
          split_layout(cin, cout) == 1);
}

// bf16 rows / bf16 weights (NP = 1): one configuration per shape (8 waves, or 2 for small row counts)
template <int CIN, int COUT>
static int launch_os_bf16(const SplitConvArgs &a, hipStream_t stream) {
  if ((long long)a.n_out * a.gy >= 16 * 1024)
    hipLaunchKernelGGL((spconv_os_split_kernel<CIN, COUT, 1, 8, 1, 1>), dim3(cdiv(a.n_out, 128), a.gy), dim3(512), 0, stream, a);
  else
    hipLaunchKernelGGL((spconv_os_split_kernel<CIN, COUT, 1, 2, 1, 1>), dim3(cdiv(a.n_out, 32), a.gy), dim3(128), 0, stream, a);
  return DF3D_OK;
}

static bool bf16_shape_ok(int cin, int cout) {
  // (128 -> 64 and 64 -> 32: the input gradients of the strided 64 -> 128 / 32 -> 64 layers -- round 6: they fell back to the
  // exact-fp32 kernel, 649 us of the TransFusion training step)
  return (cin == 32 && (cout == 32 || cout == 64)) || (cin == 64 && (cout == 32 || cout == 64 || cout == 128)) ||
         (cin == 128 && (cout == 64 || cout == 128 || cout == 256)) || (cin == 256 && (cout == 128 || cout == 256));
}

static int launch_os_bf16_any(int cin, int cout, const SplitConvArgs &a, hipStream_t stream) {
  if (cin == 32 && cout == 32) return launch_os_bf16<32, 32>(a, stream);
  if (cin == 32 && cout == 64) return launch_os_bf16<32, 64>(a, stream);
  if (cin == 64 && cout == 32) return launch_os_bf16<64, 32>(a, stream);
  if (cin == 64 && cout == 64) return launch_os_bf16<64, 64>(a, stream);
  if (cin == 64 && cout == 128) return launch_os_bf16<64, 128>(a, stream);
  if (cin == 128 && cout == 64) return launch_os_bf16<128, 64>(a, stream);
  if (cin == 128 && cout == 128) return launch_os_bf16<128, 128>(a, stream);
  if (cin == 128 && cout == 256) return launch_os_bf16<128, 256>(a, stream);
  if (cin == 256 && cout == 128) return launch_os_bf16<256, 128>(a, stream);
  if (cin == 256 && cout == 256) return launch_os_bf16<256, 256>(a, stream);
  set_error("no bf16 kernel for cin=%d cout=%d", cin, cout);
  return DF3D_EINVAL;
}

static int launch_os_any(int cin, int cout, const SplitConvArgs &a, hipStream_t stream) {
  if (cin == 256 && cout == 256) return launch_os_split_wide<256, 256>(a, stream);
  if (cin == 256 && cout == 128) return launch_os_split_wide<256, 128>(a, stream);
  if (cin == 128 && cout == 256) return launch_os_split_wide<128, 256>(a, stream);
  if (cin == 512 && cout == 64) return launch_os_split_wide<512, 64>(a, stream);
  if (cin == 512 && cout == 128) return launch_os_split_wide<512, 128>(a, stream);
  if (cin == 128 && cout == 32) return launch_os_split<128, 32>(a, stream);
  if (cin == 128 && cout == 64) return launch_os_split<128, 64>(a, stream);
  if (cin == 128 && cout == 128) return launch_os_split<128, 128>(a, stream);
  if (cin == 64 && cout == 128) return launch_os_split<64, 128>(a, stream);
  if (cin == 64 && cout == 64) return launch_os_split<64, 64>(a, stream);
  if (cin == 32 && cout == 64) return launch_os_split<32, 64>(a, stream);
  if (cin == 64 && cout == 32) return launch_os_split<64, 32>(a, stream);
  if (cin == 32 && cout == 32) return launch_os_split<32, 32>(a, stream);
  set_error("no output-stationary split kernel for cin=%d cout=%d", cin, cout);
  return DF3D_EINVAL;
}

// three-part operands (NP = 3, six products): one configuration per shape -- 8 waves per workgroup on the large maps, 4 / 2 on
// the small ones (the same rule as the bf16 launches); the loader / consumer kernel has no three-part form
template <int CIN, int COUT>
static int launch_os_p3(const SplitConvArgs &a, hipStream_t stream) {
  const long long work = (long long)a.n_out * a.gy;
  if (work >= 32 * 1024)
    hipLaunchKernelGGL((spconv_os_split_kernel<CIN, COUT, 1, 8, 1, 3>), dim3(cdiv(a.n_out, 128), a.gy), dim3(512), 0, stream, a);
  else if (work >= 12 * 1024)
    hipLaunchKernelGGL((spconv_os_split_kernel<CIN, COUT, 1, 4, 1, 3>), dim3(cdiv(a.n_out, 64), a.gy), dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL((spconv_os_split_kernel<CIN, COUT, 1, 2, 1, 3>), dim3(cdiv(a.n_out, 32), a.gy), dim3(128), 0, stream, a);
  return DF3D_OK;
}

static int launch_os_p3_any(int cin, int cout, const SplitConvArgs &a, hipStream_t stream) {
  if (cin == 256 && cout == 256) return launch_os_p3<256, 256>(a, stream);
  if (cin == 256 && cout == 128) return launch_os_p3<256, 128>(a, stream);
  if (cin == 128 && cout == 256) return launch_os_p3<128, 256>(a, stream);
  if (cin == 512 && cout == 64) return launch_os_p3<512, 64>(a, stream);
  if (cin == 512 && cout == 128) return launch_os_p3<512, 128>(a, stream);
  if (cin == 128 && cout == 32) return launch_os_p3<128, 32>(a, stream);
  if (cin == 128 && cout == 64) return launch_os_p3<128, 64>(a, stream);
  if (cin == 128 && cout == 128) return launch_os_p3<128, 128>(a, stream);
  if (cin == 64 && cout == 128) return launch_os_p3<64, 128>(a, stream);
  if (cin == 64 && cout == 64) return launch_os_p3<64, 64>(a, stream);
  if (cin == 32 && cout == 64) return launch_os_p3<32, 64>(a, stream);
  if (cin == 64 && cout == 32) return launch_os_p3<64, 32>(a, stream);
  if (cin == 32 && cout == 32) return launch_os_p3<32, 32>(a, stream);
  set_error("no three-part split kernel for cin=%d cout=%d", cin, cout);
  return DF3D_EINVAL;
}

}  // namespace df3d

using namespace df3d;

// ---- "split3" precision (round 4): operands as THREE bf16 parts (hi + mid + lo = the fp32 value exactly), six MFMA products
//      per operand pair, fp32 accumulate: fp32-grade results (dropped terms <= 2^-24 of a product) at 1/6 of the bf16 rate --
//      2.6x the fp32 matrix rate.  Same shapes as the output-stationary split kernels. ----
extern "C" size_t df3d_conv_packed_weight_bytes3(int kvol, int cin, int cout) {
  if (!split_shape_ok(cin, cout) || split_layout(cin, cout) != 1 || kvol <= 0 || kvol > DF3D_MAX_KVOL) return 0;
  return (size_t)kvol * cin * cout * 6;
}

extern "C" int df3d_conv_pack_weights3(const float *filters, int groups, int kvol, int cin, int cout, void *packed, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(filters && packed && groups >= 1, "conv_pack_weights3: bad argument");
  DF3D_CHECK_ARG(df3d_conv_packed_weight_bytes3(kvol, cin, cout) != 0 && (groups == 1 || cout <= 128),
                 "conv_pack_weights3: shape K=%d cin=%d cout=%d has no three-part kernel", kvol, cin, cout);
  // (G filter banks back to back are one bank of G * K offsets, as in df3d_conv_pack_weights_groups)
  size_t total = (size_t)groups * kvol * cin * cout / 8 * 3;
  hipLaunchKernelGGL(pack_weights3_kernel, dim3(cdiv((long long)total, 256)), dim3(256), 0, stream, filters, groups * kvol, cin,
                     cout, (u32x4 *)packed);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_split_rows3(const float *features, long long n, int c, void *split3, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(features && split3, "split_rows3: null argument");
  DF3D_CHECK_ARG(c > 0 && c % 8 == 0 && n >= 0, "split_rows3: channels must be a multiple of 8 (got %d)", c);
  size_t nblk = (size_t)n * c / 8;
  if (nblk == 0) return DF3D_OK;
  hipLaunchKernelGGL(split3_rows_kernel, dim3(cdiv((long long)nblk, 256)), dim3(256), 0, stream, features, nblk, (u32x4 *)split3);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

// df3d_conv_rows_split on three-part rows / filters (in_split3 rows of in_channels * 6 bytes; out_split3 likewise); with
// groups = 1, in_group_stride = 0 and out_channels = cout it is the sparse convolution itself; `residual` fp32 rows or NULL
extern "C" int df3d_conv_rows_split3(const void *in_split3, int n_in, int in_channels, int cin, int in_group_stride,
                                     const void *packed3, int kvol, int cout, int groups, const int32_t *nbr, int n_out,
                                     const float *bias, const float *scale, const float *shift, const float *residual,
                                     int relu, float *out, int out_channels, const int32_t *out_cols, void *out_split3,
                                     void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(in_split3 && packed3 && nbr && (out || out_split3), "conv_rows_split3: null argument");
  DF3D_CHECK_ARG(df3d_conv_packed_weight_bytes3(kvol, cin, cout) != 0, "conv_rows_split3: K=%d cin=%d cout=%d has no three-part kernel",
                 kvol, cin, cout);
  DF3D_CHECK_ARG(groups >= 1 && groups <= 65535, "conv_rows_split3: groups");
  DF3D_CHECK_ARG(in_channels % 8 == 0 && in_group_stride % 8 == 0 && in_group_stride >= 0 &&
                     (long long)(groups - 1) * in_group_stride + cin <= in_channels,
                 "conv_rows_split3: input columns of the groups must lie inside the %d-channel rows", in_channels);
  DF3D_CHECK_ARG(!residual || (groups == 1 && out_channels == cout), "conv_rows_split3: a residual needs one group and dense output rows");
  const int blocks = groups * (cout > 128 ? cout / 128 : 1);
  if (out_cols) {
    DF3D_CHECK_ARG(cout == 32 && !out_split3, "conv_rows_split3: compact output columns need cout = 32 and no split output");
  } else {
    DF3D_CHECK_ARG(out_channels % 8 == 0 && (long long)groups * cout <= out_channels,
                   "conv_rows_split3: %d x %d output columns do not fit %d-channel rows", groups, cout, out_channels);
  }
  if (n_out == 0) return DF3D_OK;
  SplitConvArgs a = {(const u32x4 *)in_split3, (const u32x4 *)packed3, nbr, bias, scale, shift, residual,
                     out, (u32x4 *)out_split3, n_in, n_out, kvol, relu, 0,
                     in_channels / 8 * 3, in_group_stride / 8 * 3, out_channels, blocks, out_cols};
  int rec = timing_rec_begin(cin, cout * groups, kvol, n_out, nbr, 3, stream);
  int rc = launch_os_p3_any(cin, cout, a, stream);
  if (rc) return rc;
  timing_rec_end(rec, stream);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" size_t df3d_conv_packed_weight_bytes(int kvol, int cin, int cout) {
  if (!split_shape_ok(cin, cout) || kvol <= 0 || kvol > DF3D_MAX_KVOL) return 0;
  return (size_t)kvol * cin * cout * 4;
}

extern "C" int df3d_conv_pack_weights(const float *filters, int kvol, int cin, int cout, void *packed, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(filters && packed, "conv_pack_weights: null argument");
  DF3D_CHECK_ARG(df3d_conv_packed_weight_bytes(kvol, cin, cout) != 0,
                 "conv_pack_weights: shape K=%d cin=%d cout=%d has no split-precision kernel", kvol, cin, cout);
  size_t total = (size_t)kvol * cin * cout / 4;     // 16-byte groups
  hipLaunchKernelGGL(pack_weights_kernel, dim3(cdiv((long long)total, 256)), dim3(256), 0, stream, filters, kvol, cin,
                     cout, split_layout(cin, cout), (u32x4 *)packed);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_conv_pack_weights_groups(const float *filters, int groups, int kvol, int cin, int cout, void *packed,
                                             void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(filters && packed && groups >= 1, "conv_pack_weights_groups: bad argument");
  DF3D_CHECK_ARG(df3d_conv_packed_weight_bytes(kvol, cin, cout) != 0 && cout <= 128,
                 "conv_pack_weights_groups: shape K=%d cin=%d cout=%d has no grouped split-precision kernel", kvol, cin, cout);
  // one column block per filter bank: the packed image is offset-major, so G banks back to back are one bank of G * K offsets
  size_t total = (size_t)groups * kvol * cin * cout / 4;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(cdiv((long long)total, 256)), dim3(256), 0, stream, filters, groups * kvol,
                     cin, cout, split_layout(cin, cout), (u32x4 *)packed);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

// scale: DF3D_POW2_SCALE_FLOATS floats -- [0] = s, [1] = the largest |value|, the rest the reduction's per-workgroup maxima
static int pow2_scale(const float *x, size_t ne, float *scale, float *inv, int inv_channels, hipStream_t stream) {
  int wgs = 1;
  if (ne) {
    wgs = (int)std::min<size_t>(POW2_PARTS, cdiv((long long)(ne / 4), 256 * 4) + 1);
    hipLaunchKernelGGL(rows_absmax_kernel, dim3(wgs), dim3(256), 0, stream, x, ne / 4, ne, scale + 2);
  } else {
    DF3D_HIP(hipMemsetAsync(scale + 2, 0, sizeof(float), stream));
  }
  hipLaunchKernelGGL(pow2_scale_kernel, dim3(std::max(1, cdiv(inv_channels, 256))), dim3(256), 0, stream, scale + 2, wgs,
                     inv_channels, scale, inv ? inv : scale);
  return DF3D_OK;
}

extern "C" int df3d_pow2_scale_floats(void) { return 2 + POW2_PARTS; }

extern "C" int df3d_split_rows_scaled(const float *features, long long n, int c, void *split, float *scale, float *inv_scale,
                                      int inv_channels, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(features && split && scale && inv_scale, "split_rows_scaled: null argument");
  DF3D_CHECK_ARG(c > 0 && c % 8 == 0 && n >= 0 && inv_channels >= 1, "split_rows_scaled: channels must be a multiple of 8 (got %d)", c);
  DF3D_CHECK_ARG((size_t)features % 16 == 0, "split_rows_scaled: rows must be 16-byte aligned");
  const size_t ne = (size_t)n * c, nblk = ne / 8;
  int rc = pow2_scale(features, ne, scale, inv_scale, inv_channels, stream);
  if (rc) return rc;
  if (nblk)
    hipLaunchKernelGGL(split_rows_scaled_kernel, dim3(cdiv((long long)nblk, 256)), dim3(256), 0, stream, features, nblk, scale,
                       (u32x4 *)split);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_rows_pow2_scale(const float *x, long long n_elems, float *scale, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(x && scale && n_elems >= 0, "rows_pow2_scale: bad argument");
  DF3D_CHECK_ARG((size_t)x % 16 == 0, "rows_pow2_scale: rows must be 16-byte aligned");
  int rc = pow2_scale(x, (size_t)n_elems, scale, nullptr, 0, stream);
  if (rc) return rc;
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_split_rows(const float *features, long long n, int c, void *split, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(features && split, "split_rows: null argument");
  DF3D_CHECK_ARG(c > 0 && c % 8 == 0 && n >= 0, "split_rows: channels must be a multiple of 8 (got %d)", c);
  size_t nblk = (size_t)n * c / 8;
  if (nblk == 0) return DF3D_OK;
  hipLaunchKernelGGL(split_rows_kernel, dim3(cdiv((long long)nblk, 256)), dim3(256), 0, stream, features, nblk,
                     (u32x4 *)split);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_sparse_conv_split(const void *features_split, int n_in, int cin, const void *packed_filters,
                                      int kvol, int cout, const int32_t *nbr, int n_out, const float *bias,
                                      const float *scale, const float *shift, const float *residual, int relu,
                                      float *out, void *out_split, const int32_t *tile_rows, int ntiles,
                                      void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(features_split && packed_filters && nbr && (out || out_split), "sparse_conv_split: null argument");
  DF3D_CHECK_ARG(kvol > 0 && kvol <= DF3D_MAX_KVOL, "sparse_conv_split: kernel volume %d unsupported", kvol);
  DF3D_CHECK_ARG(split_shape_ok(cin, cout), "sparse_conv_split: cin=%d cout=%d has no split-precision kernel", cin,
                 cout);
  DF3D_CHECK_ARG(out || split_layout(cin, cout) == 1, "sparse_conv_split: the pair-compacted kernel always writes fp32 rows");
  if (n_out == 0) return DF3D_OK;
  SplitConvArgs a = {(const u32x4 *)features_split, (const u32x4 *)packed_filters, nbr, bias, scale, shift, residual,
                     out, (u32x4 *)out_split, n_in, n_out, kvol, relu,
                     getenv("DF3D_OS_DBG") ? atoi(getenv("DF3D_OS_DBG")) : 0,
                     cin / 4, 0, cout, cout > 128 ? cout / 128 : 1, nullptr};
  if (ntiles == -1 && tile_rows) a.order = tile_rows;       // tiling order of the output-stationary kernel
#ifdef DF3D_OS_TRACE
  a.trace = g_os_trace;
#endif
  int rec = timing_rec_begin(cin, cout, kvol, n_out, nbr, 1 | (out && out_split ? 8 : 0), stream);   // bit 3: both row formats written
  int rc;
  if (split_layout(cin, cout) == 1) {
    rc = launch_os_any(cin, cout, a, stream);
  } else if (cout == 128) {
    rc = cin == 128 ? launch_split<128, 128>(a, tile_rows, ntiles, stream)
                    : launch_split<64, 128>(a, tile_rows, ntiles, stream);
  } else {
    rc = cin == 64 ? launch_split<64, 64>(a, tile_rows, ntiles, stream)
                   : launch_split<32, 64>(a, tile_rows, ntiles, stream);
  }
  if (rc) return rc;
  timing_rec_end(rec, stream);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_conv_rows_split(const void *in_split, int n_in, int in_channels, int cin, int in_group_stride,
                                    const void *packed_filters, int kvol, int cout, int groups, const int32_t *nbr,
                                    int n_out, const float *bias, const float *scale, const float *shift, int relu,
                                    float *out, int out_channels, const int32_t *out_cols, void *out_split,
                                    void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(in_split && packed_filters && nbr && (out || out_split), "conv_rows_split: null argument");
  DF3D_CHECK_ARG(kvol > 0 && kvol <= DF3D_MAX_KVOL, "conv_rows_split: kernel volume %d unsupported", kvol);
  DF3D_CHECK_ARG(split_shape_ok(cin, cout) && split_layout(cin, cout) == 1,
                 "conv_rows_split: cin=%d cout=%d has no output-stationary split kernel", cin, cout);
  DF3D_CHECK_ARG(groups >= 1 && groups <= 65535, "conv_rows_split: groups");
  DF3D_CHECK_ARG(in_channels % 8 == 0 && in_group_stride % 8 == 0 && in_group_stride >= 0 &&
                     (long long)(groups - 1) * in_group_stride + cin <= in_channels,
                 "conv_rows_split: input columns [g*%d, g*%d+%d) must lie inside the %d-channel rows", in_group_stride,
                 in_group_stride, cin, in_channels);
  const int blocks = groups * (cout > 128 ? cout / 128 : 1);
  if (out_cols) {
    DF3D_CHECK_ARG(cout == 32 && !out_split, "conv_rows_split: compact output columns need cout = 32 and no split output");
  } else {
    DF3D_CHECK_ARG(out_channels % 8 == 0 && (long long)groups * cout <= out_channels,
                   "conv_rows_split: %d x %d output columns do not fit %d-channel rows", groups, cout, out_channels);
  }
  if (n_out == 0) return DF3D_OK;
  SplitConvArgs a = {(const u32x4 *)in_split, (const u32x4 *)packed_filters, nbr, bias, scale, shift, nullptr,
                     out, (u32x4 *)out_split, n_in, n_out, kvol, relu,
                     getenv("DF3D_OS_DBG") ? atoi(getenv("DF3D_OS_DBG")) : 0, in_channels / 4, in_group_stride / 4, out_channels, blocks, out_cols};
#ifdef DF3D_OS_TRACE
  a.trace = g_os_trace;
#endif
  int rec = timing_rec_begin(cin, cout * groups, kvol, n_out, nbr, 1, stream);
  int rc = launch_os_any(cin, cout, a, stream);
  if (rc) return rc;
  timing_rec_end(rec, stream);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

// ---- bf16 rows / bf16 weights, fp32 accumulate (BASELINE configs[2]: "bf16, fp32 accumulate") -------------------
extern "C" int df3d_rows_to_bf16(const float *features, long long n, int c, void *rows_bf16, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(c > 0 && c % 8 == 0 && n >= 0, "rows_to_bf16: channels must be a multiple of 8 (got %d)", c);
  size_t nblk = (size_t)n * c / 8;
  if (nblk == 0) return DF3D_OK;
  DF3D_CHECK_ARG(features && rows_bf16, "rows_to_bf16: null argument");
  hipLaunchKernelGGL(bf16_rows_kernel, dim3(cdiv((long long)nblk, 256)), dim3(256), 0, stream, features, nblk,
                     (u32x4 *)rows_bf16);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_rows_from_bf16(const void *rows_bf16, long long n, int c, float *features, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(c > 0 && c % 8 == 0 && n >= 0, "rows_from_bf16: channels must be a multiple of 8 (got %d)", c);
  size_t nblk = (size_t)n * c / 8;
  if (nblk == 0) return DF3D_OK;
  DF3D_CHECK_ARG(features && rows_bf16, "rows_from_bf16: null argument");
  hipLaunchKernelGGL(bf16_rows_to_f32_kernel, dim3(cdiv((long long)nblk, 256)), dim3(256), 0, stream,
                     (const u32x4 *)rows_bf16, nblk, features);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" size_t df3d_conv_packed_weight_bytes_bf16(int kvol, int cin, int cout) {
  if (!bf16_shape_ok(cin, cout) || kvol <= 0 || kvol > DF3D_MAX_KVOL) return 0;
  return (size_t)kvol * cin * cout * 2;
}

extern "C" int df3d_conv_pack_weights_bf16(const float *filters, int kvol, int cin, int cout, void *packed,
                                           void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(filters && packed, "conv_pack_weights_bf16: null argument");
  DF3D_CHECK_ARG(df3d_conv_packed_weight_bytes_bf16(kvol, cin, cout) != 0,
                 "conv_pack_weights_bf16: shape K=%d cin=%d cout=%d has no bf16 kernel", kvol, cin, cout);
  size_t total = (size_t)kvol * cin * cout / 8;
  hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3(cdiv((long long)total, 256)), dim3(256), 0, stream, filters, kvol,
                     cin, cout, (u32x4 *)packed);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_sparse_conv_bf16(const void *features_bf16, int n_in, int cin, const void *packed_filters, int kvol,
                                     int cout, const int32_t *nbr, int n_out, const float *bias, const float *scale,
                                     const float *shift, const void *residual_bf16, int relu, float *out,
                                     void *out_bf16, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(features_bf16 && packed_filters && nbr && (out || out_bf16), "sparse_conv_bf16: null argument");
  DF3D_CHECK_ARG(kvol > 0 && kvol <= DF3D_MAX_KVOL, "sparse_conv_bf16: kernel volume %d unsupported", kvol);
  DF3D_CHECK_ARG(bf16_shape_ok(cin, cout), "sparse_conv_bf16: cin=%d cout=%d has no bf16 kernel", cin, cout);
  if (n_out == 0) return DF3D_OK;
  SplitConvArgs a = {(const u32x4 *)features_bf16, (const u32x4 *)packed_filters, nbr, bias, scale, shift,
                     (const float *)residual_bf16, out, (u32x4 *)out_bf16, n_in, n_out, kvol, relu, 0,
                     cin / 8, 0, cout, cout > 128 ? cout / 128 : 1, nullptr};
  int rec = timing_rec_begin(cin, cout, kvol, n_out, nbr, 2, stream);
  int rc = launch_os_bf16_any(cin, cout, a, stream);
  if (rc) return rc;
  timing_rec_end(rec, stream);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}
