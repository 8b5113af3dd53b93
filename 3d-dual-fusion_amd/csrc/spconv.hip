// Fused sparse convolution for gfx950: output-stationary implicit GEMM on fp32 MFMA.
//
// The reference runs, per conv, 27 x {gather kernel -> GEMM -> scatter-add kernel} with a
// D2H sync (TF/mmdet3d/ops/spconv/include/spconv/spconv_ops.h:260-361, reordering.cu.h:22-158)
// and BN / ReLU / residual as separate elementwise passes (CP/det3d/models/backbones/scn.py:
// 76-94).  That moves ~6x the algorithmic bytes.  Here ONE kernel per conv:
//
//   workgroup (4 waves) owns TM = 64*RT consecutive output rows and all COUT columns;
//   for every kernel offset k that has at least one neighbour in the tile
//     for every 32-wide slice of CIN
//        W[k][slice][:] is staged global -> LDS once per workgroup (double buffered, one
//          barrier per step), rows permuted/swizzled so the B-fragment reads are
//          conflict-free ds_read_b32;
//        each lane loads its A fragment (KS contiguous floats of the gathered input row,
//          16-byte vector loads straight from HBM/L2 - no LDS round trip for activations);
//        v_mfma_f32_16x16x4_f32 accumulates into registers (exact fp32, k-ordered fmaf chain);
//   epilogue: + bias, folded BatchNorm (scale, shift), + residual, ReLU, store.
//
// Each output row is written exactly once; each input row is read once per (output, offset)
// pair that uses it: algorithmic bytes = R*CIN*4 + N_out*COUT*4 + 4*K*N_out (nbr table)
// + K*CIN*COUT*4.  The reduction index order inside MFMA is a permutation
// of cin (lane group g owns cin [g*KS, (g+1)*KS) of the slice); fp32 accumulate.
// Roofline: HBM for C <= 32 (AI 3-14 flop/B), fp32 MFMA (157 TF) for C >= 64.
#include <stdlib.h>

#include <vector>

#include <mutex>

#include "common.h"

namespace df3d {

DF3D_SPLIT_OVERFLOW_TU(spconv)

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
  const float *feat;
  const float *w;
  const int32_t *nbr;
  const float *bias, *scale, *shift, *residual;
  float *out;
  int n_in, n_out, K, cin, cout, relu;
  // MFMA kernel only (df3d_sparse_conv_grouped): floats per input / output row (the operands may be column slices of
  // wider rows), and the column offset between the slices of two consecutive groups (blockIdx.y; filters, bias, scale
  // and shift of a group follow the previous group's)
  int ldi, ldo, gi, go;
  void *out_split;        // spconv_small_lists_kernel only: also the operand split of the result (hi | lo per 8 channels), or null
};

template <int KS>
struct AFrag {
  float v[KS];
};

// CINP: cin rounded up to a multiple of KC.  KC: cin slice per step (8, 16 or 32).
// RT: 16-row tiles per wave.  VEC: cin % 4 == 0 (16-byte A loads).
template <int CINP, int COUT, int KC, int RT, bool VEC>
__global__ __launch_bounds__(256) void spconv_mfma_kernel(ConvArgs a) {
  constexpr int KS = KC / 4;          // k-steps per slice = floats per lane per row
  constexpr int NCH = CINP / KC;      // slices
  constexpr int CT = COUT / 16;       // column tiles
  constexpr int TM = 64 * RT;         // rows per workgroup
  constexpr int WROWS = 16 * RT;      // rows per wave
  constexpr int WT = KC * COUT;       // floats per W tile
  constexpr int WPT = (WT / 4 + 255) / 256;  // float4 per thread for staging
  constexpr bool SWZ = COUT >= 32;

  __shared__ float Wl[2][WT];
  __shared__ int nbrL[DF3D_MAX_KVOL][TM];
  __shared__ unsigned wg_mask;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;
  if (gridDim.y > 1) {                          // grouped launch: this workgroup's slice of the operands
    const int grp = blockIdx.y;
    a.feat += (size_t)grp * a.gi;
    a.w += (size_t)grp * a.K * a.cin * COUT;
    a.out += (size_t)grp * a.go;
    if (a.bias) a.bias += grp * COUT;
    if (a.scale) a.scale += grp * COUT;
    if (a.shift) a.shift += grp * COUT;
  }

  // XCD-aware tile order: consecutive tiles (which share neighbour rows) stay on one XCD/L2
  int nt = gridDim.x;
  int bid = blockIdx.x;
  int tile = bid;
  if ((nt & 7) == 0) tile = (bid & 7) * (nt >> 3) + (bid >> 3);
  const int row0 = tile * TM;

  if (tid == 0) wg_mask = 0u;
  for (int e = tid; e < a.K * TM; e += 256) {
    int k = e / TM, r = e - k * TM;
    int row = row0 + r;
    nbrL[k][r] = (row < a.n_out) ? a.nbr[(size_t)k * a.n_out + row] : -1;
  }
  __syncthreads();
  // per-wave activity mask over k (wave-uniform), workgroup mask = OR
  unsigned wmask = 0u;
  for (int k = 0; k < a.K; ++k) {
    bool any = false;
#pragma unroll
    for (int q = 0; q < (WROWS + 63) / 64; ++q) {
      int r = q * 64 + lane;
      int v = (r < WROWS) ? nbrL[k][wave * WROWS + r] : -1;
      any |= (__ballot(v >= 0) != 0ull);
    }
    if (any) wmask |= (1u << k);
  }
  if (lane == 0 && wmask) atomicOr(&wg_mask, wmask);
  __syncthreads();
  const unsigned gmask = wg_mask;
  const int nact = __popc(gmask);
  const int steps = nact * NCH;

  f32x4 acc[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- helpers -------------------------------------------------------------------
  auto kth_active = [&](int ai) -> int {  // index of the ai-th set bit of gmask
    unsigned m = gmask;
    for (int i = 0; i < ai; ++i) m &= m - 1;
    return __ffs((int)m) - 1;
  };
  f32x4 wreg[WPT];
  auto load_w = [&](int k, int ch) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      int e = tid + 256 * i;
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (e < WT / 4) {
        int r = e / (COUT / 4), c4 = e - r * (COUT / 4);
        int ci = ch * KC + r;
        if (ci < a.cin) v = *(const f32x4 *)(a.w + ((size_t)k * a.cin + ci) * COUT + c4 * 4);
      }
      wreg[i] = v;
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      int e = tid + 256 * i;
      if (e < WT / 4) {
        int r = e / (COUT / 4), c4 = e - r * (COUT / 4);
        int rp = 4 * (r % KS) + (r / KS);  // global row g*KS+j -> LDS row 4j+g
        int col = c4 * 4;
        if (SWZ) col ^= ((rp & 1) << 4);
        *(f32x4 *)(&Wl[buf][rp * COUT + col]) = wreg[i];
      }
    }
  };
  AFrag<KS> anext[RT], acur[RT];
  auto load_a = [&](int k, int ch) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      int idx = nbrL[k][wave * WROWS + rt * 16 + n];
      int c0 = ch * KC + g * KS;
      if (VEC) {
#pragma unroll
        for (int q = 0; q < KS / 4; ++q) {
          f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (idx >= 0) v = *(const f32x4 *)(a.feat + (size_t)idx * a.ldi + c0 + q * 4);
          anext[rt].v[q * 4 + 0] = v[0];
          anext[rt].v[q * 4 + 1] = v[1];
          anext[rt].v[q * 4 + 2] = v[2];
          anext[rt].v[q * 4 + 3] = v[3];
        }
      } else {
#pragma unroll
        for (int j = 0; j < KS; ++j) {
          float v = 0.f;
          if (idx >= 0 && c0 + j < a.cin) v = a.feat[(size_t)idx * a.ldi + c0 + j];
          anext[rt].v[j] = v;
        }
      }
    }
  };

  // ---- software pipeline over (active offset, cin slice) ---------------------------
  int k_next = 0;
  if (steps > 0) {
    k_next = kth_active(0);
    load_w(k_next, 0);
    store_w(0);
    if ((wmask >> k_next) & 1u) load_a(k_next, 0);
  }
  for (int s = 0; s < steps; ++s) {
    const int k_cur = k_next;
    const bool wave_on = (wmask >> k_cur) & 1u;
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acur[rt] = anext[rt];
    const bool more = s + 1 < steps;
    int ch_n = 0;
    if (more) {
      int s1 = s + 1;
      int ai = s1 / NCH;
      ch_n = s1 - ai * NCH;
      k_next = (ch_n == 0) ? kth_active(ai) : k_cur;
      load_w(k_next, ch_n);
      if ((wmask >> k_next) & 1u) load_a(k_next, ch_n);
    }
    if (wave_on) {
      const float *wb = Wl[s & 1];
#pragma unroll
      for (int j = 0; j < KS; ++j) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          int rp = 4 * j + g;
          int col = ct * 16 + n;
          if (SWZ) col ^= ((g & 1) << 4);
          float b = wb[rp * COUT + col];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[rt].v[j], b, acc[rt][ct], 0, 0, 0);
        }
      }
    }
    if (more) store_w((s + 1) & 1);
  }

  // ---- epilogue: bias, folded BN, residual, ReLU -----------------------------------
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    int col = ct * 16 + n;
    float bi = a.bias ? a.bias[col] : 0.f;
    float sc = a.scale ? a.scale[col] : 1.f;
    float sh = a.shift ? a.shift[col] : 0.f;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = row0 + wave * WROWS + rt * 16 + 4 * g + r;
        if (row < a.n_out) {
          float v = (acc[rt][ct][r] + bi) * sc + sh;
          if (a.residual) v += a.residual[(size_t)row * COUT + col];
          if (a.relu) v = fmaxf(v, 0.f);
          a.out[(size_t)row * a.ldo + col] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// v2 for the compute-bound layers (COUT >= 64): pair-compacted MFMA rows, LDS accumulators,
// one workgroup per CU.
//
// The output-stationary kernel above issues MFMAs for every (row, offset) of its tile, also
// where the neighbour is absent (42 % of the slots at conv4's occupancy).  Here
//   * the grid is sized to the machine: n_out rows are cut into 256*m equal tiles of TM rows
//     (TM <= ~200-380, chosen at launch), one resident workgroup per CU, no tail wave;
//   * accumulators live in LDS ([wave][TM][COUT/4] fp32), so MFMA rows need not be output
//     rows: for each kernel offset the VALID (output row, input row) pairs of the tile are
//     compacted (ballot + popcount into a per-wave list) and processed 16 at a time; only the
//     last chunk of an offset is partially filled (~2-6 % waste instead of ~70 %);
//   * each of the 4 waves owns a quarter of the output columns and keeps the matching
//     [CIN x COUT/4] slice of W[k] in REGISTERS as MFMA B operands (8-byte loads from L2,
//     double-buffered one offset ahead); waves never synchronise inside the main loop;
//   * A fragments (CIN/4 contiguous floats of the gathered input row per lane) come straight
//     from HBM/L2 with 16-byte loads, one chunk group ahead of the MFMAs;
//   * per chunk the 16 x COUT/4 product is added into the LDS accumulators (8-byte RMW; an
//     output row appears at most once per offset, so no atomics), one chunk behind the MFMAs;
//   * epilogue from LDS: bias, folded BN, residual, ReLU, 16-byte stores.
template <int CIN, int COUT>
__global__ __launch_bounds__(256, 1) void spconv_pair_kernel(ConvArgs a, int TM, const int32_t *__restrict__ tile_rows) {
  constexpr int KS = CIN / 4;        // k-steps = floats of one input row held by a lane
  constexpr int CS = COUT / 4;       // output columns per wave
  constexpr int CT = CS / 16;        // 16-wide column tiles per wave (1 or 2)
  constexpr int NCH = 2 / CT;        // chunks in flight -> always 2 independent accumulators
  static_assert(CT == 1 || CT == 2, "COUT must be 64 or 128");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float *accL = (float *)smem;                                  // [4][TM+1][CS]; row TM = trash row
  int *idxL = (int *)(smem + (size_t)4 * (TM + 1) * CS * 4);    // [K][TM]: nbr tile, compacted in place to input rows
  unsigned short *listL = (unsigned short *)(idxL + a.K * TM);  // [K][TM]: matching output rows (tile-local)
  __shared__ int cntL[DF3D_MAX_KVOL];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;
  const int cs0 = wave * CS;
  const int K = a.K;
  // Row range of this workgroup: equal row counts, or - when the caller provides them - boundaries
  // that balance the number of rulebook PAIRS per workgroup (df3d_conv_tiles; occupancy varies 1.4x
  // between tiles of a LiDAR sweep and the slowest tile sets the kernel time).  Ranges longer than the
  // LDS tile are walked in passes of TM rows.
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
  // ---- compact the valid pairs of every offset once per tile (offsets striped over the waves):
  //      idxL[k][j] = input row, listL[k][j] = output row of the j-th valid pair (in place: j <= r) ----
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

  // ---- item list: one entry per (offset, chunk group) in processing order: k | grp << 8 | cnt << 16 ----
  int *itemL = (int *)(listL + (size_t)K * TM + (((size_t)K * TM) & 1));     // 4-byte aligned, [T+2]
  __shared__ int segL[DF3D_MAX_KVOL + 1];                                     // first item of each offset; [K] = T
  auto ngroups = [&](int cnt) { return (((cnt + 15) >> 4) + NCH - 1) / NCH; };
  if (tid == 0) {
    int t = 0;
    for (int k = 0; k < K; ++k) {
      segL[k] = t;
      int cnt = cntL[k];
      int ng = ngroups(cnt);
      for (int gq = 0; gq < ng; ++gq) itemL[t++] = k | (gq << 8) | (cnt << 16);
    }
    segL[K] = t;
    int last = t > 0 ? itemL[t - 1] : 0;
    itemL[t] = last;          // two sentinels: the pipeline's look-ahead past the end re-reads the last item
    itemL[t + 1] = last;
  }
  __syncthreads();
  const int T = segL[K];

  float bcur[KS][CT], bnext[KS][CT];
  auto load_b = [&](int k, float (&dst)[KS][CT]) {
    const float *wk = a.w + ((size_t)k * CIN + (size_t)g * KS) * COUT + cs0;
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      if (CT == 2) {
        float2 v = *(const float2 *)(wk + (size_t)j * COUT + 2 * n);
        dst[j][0] = v.x;
        dst[j][CT - 1] = v.y;
      } else {
        dst[j][0] = wk[(size_t)j * COUT + n];
      }
    }
  };
  // Branch-free gathers: slots past the end of an offset's pair list re-read its last pair; their
  // products land in the trash row at flush time (MFMA rows are independent of each other).
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
  float a0[NCH][KS], a1[NCH][KS], a2[NCH][KS];     // A fragments of items t, t+1, t+2 (gathers run 2 items ahead)
  auto load_a = [&](const int (&idx)[NCH], float (&dst)[NCH][KS]) {
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      const float *src = a.feat + (size_t)idx[h] * CIN + g * KS;
#pragma unroll
      for (int q = 0; q < KS / 4; ++q) {
        f32x4 v = *(const f32x4 *)(src + q * 4);
        dst[h][q * 4 + 0] = v[0];
        dst[h][q * 4 + 1] = v[1];
        dst[h][q * 4 + 2] = v[2];
        dst[h][q * 4 + 3] = v[3];
      }
    }
  };
  f32x4 acc[NCH][CT], aprev[NCH][CT];
  // LDS accumulate: plain read-add-write, the two halves in different MFMA shadows (LDS float atomics
  // cost ~150 cycles per wave instruction on gfx950: measured 2x slower overall).  Rows of one group
  // are distinct (trash-row duplicates do not matter).
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
    load_b(itemL[0] & 0xff, bnext);
    read_idx(itemL[0], idxn);
    load_a(idxn, a1);
    read_idx(itemL[1], idxn);
    load_a(idxn, a2);
  }
  for (int k = 0; k < K; ++k) {
    const int t0 = segL[k], t1 = segL[k + 1];
    if (t0 == t1) continue;
#pragma unroll
    for (int j = 0; j < KS; ++j)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) bcur[j][ct] = bnext[j][ct];
    if (t1 < T) load_b(itemL[t1] & 0xff, bnext);          // next active offset's weights, one offset ahead
    for (int t = t0; t < t1; ++t) {
      // ---- one item = NCH chunks of 16 pairs.  Branch-free body, software-scheduled.  The A ring
      //      (items t, t+1, t+2) rotates by NAME: three copies of the body selected by t % 3, so no
      //      register moves are spent on it. ----
      auto body = [&](float (&cur)[NCH][KS], float (&tgt)[NCH][KS]) {
        const int it0 = itemL[t], it2 = itemL[t + 2];
        read_idx(it2, idxn);
        read_rows(it0, rl_cur);
        flush_read(rl_prev);
#pragma unroll
        for (int h = 0; h < NCH; ++h)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) acc[h][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KS; ++j)
#pragma unroll
          for (int h = 0; h < NCH; ++h)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
              acc[h][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[h][j], bcur[j][ct], acc[h][ct], 0, 0, 0);
        load_a(idxn, tgt);                                  // gathers of item t+2 (into the slot item t frees)
        flush_write(rl_prev, aprev);                        // item t-1's LDS update
#pragma unroll
        for (int h = 0; h < NCH; ++h) {
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) aprev[h][ct] = acc[h][ct];
#pragma unroll
          for (int r = 0; r < 4; ++r) rl_prev[h][r] = rl_cur[h][r];
        }
        // The wave is alone on its SIMD and issue is in order: every non-MFMA instruction must sit in
        // the 32-cycle shadow of an MFMA.  Ask the scheduler for "1 MFMA, then a few others" groups.
#pragma unroll
        for (int i = 0; i < KS * NCH * CT; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               // 1 MFMA
          if (i < 6) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);    // LDS reads first (indices, rows, acc)
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);               // up to 3 VALU
          if (i >= 8 && i < 8 + 2 * NCH * (KS / 4) && (i & 1) == 0)
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);             // one gather load every other MFMA
          if (i >= KS * NCH * CT / 2 && i < KS * NCH * CT / 2 + 4 * NCH)
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);             // LDS writes of the flush
        }
      };
      // NOTE: the MFMAs of item t cannot overwrite `cur` before they have read it: load_a(tgt == cur's
      // successor slot) targets the slot of item t+2 == slot of item t-1, which is dead.
      switch (t % 3) {
        case 0: body(a1, a0); break;     // slots: t -> a1, t+1 -> a2, t+2 -> a0
        case 1: body(a2, a1); break;
        default: body(a0, a2); break;
      }
    }
  }
  flush_read(rl_prev);
  flush_write(rl_prev, aprev);
  __builtin_amdgcn_wave_barrier();

  // ---- epilogue: this wave's CS columns of every row of the tile ----
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
      v = (v + bi) * sc + sh;
      size_t o = (size_t)row * COUT + cs0 + lc;
      if (a.residual) v += *(const f32x4 *)(a.residual + o);
      if (a.relu) {
        v[0] = fmaxf(v[0], 0.f);
        v[1] = fmaxf(v[1], 0.f);
        v[2] = fmaxf(v[2], 0.f);
        v[3] = fmaxf(v[3], 0.f);
      }
      *(f32x4 *)(a.out + o) = v;
    }
  }
  }  // passes over the row range
}

template <int CIN, int COUT>
static int launch_pair(const ConvArgs &a, const int32_t *tile_rows, int ntiles, hipStream_t stream) {
  // LDS per row: accumulators (COUT floats) + nbr tile (K ints) + 4 list entries
  const size_t per_row = (size_t)COUT * 4 + (size_t)a.K * 4 + (size_t)a.K * 2;   // + one trash row below
  int tm_max = (int)((144 * 1024) / per_row) & ~3;
  static int num_cu = 0;
  if (!num_cu) {
    hipDeviceProp_t p;
    num_cu = (hipGetDeviceProperties(&p, 0) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
  }
  static const int wgs_per_cu = getenv("DF3D_PAIR_WGS") ? atoi(getenv("DF3D_PAIR_WGS")) : 1;
  if (wgs_per_cu > 1) tm_max = (int)((144 * 1024 / wgs_per_cu) / per_row) & ~3;
  int slots = num_cu * wgs_per_cu;
  int m = cdiv(a.n_out, (long long)slots * tm_max);
  int TM = (cdiv(a.n_out, (long long)slots * m) + 3) & ~3;
  if (TM < 16) TM = 16;
  size_t items = (size_t)a.K * (TM / 16 + 2) + 8;   // upper bound on (offset, group) items
  size_t lds = (size_t)TM * per_row + (size_t)COUT * 4 + items * 4 + 64;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute((const void *)spconv_pair_kernel<CIN, COUT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS) failed: %s", hipGetErrorString(e));
      return DF3D_EHIP;
    }
    configured = true;
  }
  int nt = cdiv(a.n_out, TM);
  if (tile_rows && ntiles > 0) {
    // pair-balanced ranges: sparse regions get more rows than the equal split, so give the LDS tile
    // the full budget; longer ranges take extra passes
    TM = tm_max;
    lds = (size_t)TM * per_row + (size_t)COUT * 4 + ((size_t)a.K * (TM / 16 + 2) + 8) * 4 + 64;
    nt = ntiles;
  } else {
    tile_rows = nullptr;
  }
  hipLaunchKernelGGL((spconv_pair_kernel<CIN, COUT>), dim3(nt), dim3(256), lds, stream, a, TM, tile_rows);
  return DF3D_OK;
}

static int pair_ntiles(int n_out, int cin, int cout);

// returns 1 if handled, 0 if not applicable, <0 on error
static int dispatch_pair(const ConvArgs &a, const int32_t *tile_rows, int ntiles, hipStream_t stream) {
  int rc = 0;
  if (pair_ntiles(a.n_out, a.cin, a.cout) == 0) return 0;
  if (a.cout == 128) {
    if (a.cin == 128) rc = launch_pair<128, 128>(a, tile_rows, ntiles, stream);
    else if (a.cin == 64) rc = launch_pair<64, 128>(a, tile_rows, ntiles, stream);
    else return 0;
  } else if (a.cout == 64) {
    if (a.cin == 64) rc = launch_pair<64, 64>(a, tile_rows, ntiles, stream);
    else if (a.cin == 32) rc = launch_pair<32, 64>(a, tile_rows, ntiles, stream);
    else return 0;
  } else {
    return 0;
  }
  return rc < 0 ? rc : 1;
}

// Any-shape fallback (correctness path for channel counts outside the tuned table).
__global__ __launch_bounds__(256) void spconv_generic_kernel(ConvArgs a) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)a.n_out * a.cout) return;
  int o = (int)(t / a.cout), co = (int)(t - (size_t)o * a.cout);
  float acc = 0.f;
  for (int k = 0; k < a.K; ++k) {
    int idx = a.nbr[(size_t)k * a.n_out + o];
    if (idx < 0) continue;
    const float *f = a.feat + (size_t)idx * a.cin;
    const float *w = a.w + (size_t)k * a.cin * a.cout + co;
    for (int ci = 0; ci < a.cin; ++ci) acc = fmaf(f[ci], w[(size_t)ci * a.cout], acc);
  }
  float v = (acc + (a.bias ? a.bias[co] : 0.f)) * (a.scale ? a.scale[co] : 1.f) + (a.shift ? a.shift[co] : 0.f);
  if (a.residual) v += a.residual[t];
  if (a.relu) v = fmaxf(v, 0.f);
  a.out[t] = v;
}

// ---------------------------------------------------------------------------------------------------------
// C <= 16 input channels and 16 output channels (conv_input 5 -> 16, the conv1 blocks 16 -> 16): vector-ALU kernel.
// These layers are 0.06 GFLOP on ~38 k rows with ~3 neighbours per row -- nothing for the matrix cores to do.  The
// MFMA kernel above ran them as 27 barrier-separated steps of four tiny matrix instructions with a one-step gather
// prefetch: ~0.8 us per step of pure latency, 22-29 us per launch for ~15 MB of traffic.
// Here a row is owned by COUT / 4 adjacent lanes (4 output channels each), the whole filter bank sits in LDS (27.6 KB,
// fetched once per persistent workgroup, read as broadcast float4), the 27 neighbour lookups are one batch of
// independent loads and only the present neighbours are gathered; no barrier after the weights are in.
// Measured on MI355X: 16 -> 16 29.5 -> 17.5 us, 5 -> 16 26 -> 16 us per launch.  (Tried and slower: splitting the
// INPUT channels over the lanes with branch-free gathers, 36 us; the same kernel for 16 -> 32, 40-68 us vs 38.)
// The accumulation order is the MFMA kernel's (offsets ascending; inside an offset the channels in the order the
// 16x16x4 k-groups consumed them: ci = g * KS + j for j outer, g inner), one fmaf per product: bit-identical results.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void spconv_small_kernel(ConvArgs a) {
  constexpr int LPR = COUT / 4;                   // lanes per row
  constexpr int RPB = 256 / LPR;                  // rows per workgroup pass
  constexpr int CINP = CIN <= 8 ? 8 : 16;         // the padded channel count the MFMA kernel worked on
  constexpr int KS = CINP / 4;
  constexpr int WSTR = CIN * COUT + 32;           // floats between two offsets' filters: rows that stand on different
                                                  // offsets read different bank halves
  extern __shared__ float Wl[];                   // [K][WSTR] filters, then [RPB][K] compacted (offset, input row) lists
  const int tid = threadIdx.x;
  for (int e = tid; e < a.K * CIN * COUT / 4; e += 256) {
    const int k = e / (CIN * COUT / 4), r = e - k * (CIN * COUT / 4);
    *(f32x4 *)(Wl + (size_t)k * WSTR + r * 4) = ((const f32x4 *)a.w)[e];
  }
  int *lists = (int *)(Wl + (size_t)a.K * WSTR);
  __syncthreads();
  const int c4 = (tid % LPR) * 4, rl = tid / LPR;
  int *mine = lists + rl * DF3D_MAX_KVOL;
  // persistent workgroups: the filter bank is fetched once per workgroup, not once per 64 rows
  for (int row = blockIdx.x * RPB + rl; row < a.n_out; row += gridDim.x * RPB) {
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the row's present neighbours, compacted in offset order (every lane of the row computes the same list, lane 0
    // writes it): the loop below then runs max-over-the-wave(list length) times, not |union of the rows' offsets| times --
    // a stride-2 layer has 2 of 27 offsets per row, and 8-16 rows share a wave
    int idx[DF3D_MAX_KVOL];
#pragma unroll
    for (int k = 0; k < DF3D_MAX_KVOL; ++k) idx[k] = k < a.K ? a.nbr[(size_t)k * a.n_out + row] : -1;
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < DF3D_MAX_KVOL; ++k) {
      if (idx[k] >= 0) {
        if (c4 == 0) mine[cnt] = idx[k] | (k << 26);
        ++cnt;
      }
    }
    for (int j = 0; j < cnt; ++j) {
      const int pk = mine[j];                     // written by this row's lane 0 of the same wave, in program order
      const int k = (unsigned)pk >> 26;
      const float *f = a.feat + (size_t)(pk & 0x3ffffff) * CIN;
      float x[CINP];
      if constexpr (CIN % 4 == 0) {
#pragma unroll
        for (int q = 0; q < CIN / 4; ++q) {
          const f32x4 v = *(const f32x4 *)(f + q * 4);
          x[q * 4] = v[0], x[q * 4 + 1] = v[1], x[q * 4 + 2] = v[2], x[q * 4 + 3] = v[3];
        }
      } else {
#pragma unroll
        for (int ci = 0; ci < CINP; ++ci) x[ci] = ci < CIN ? f[ci] : 0.f;
      }
      const float *wk = Wl + (size_t)k * WSTR + c4;
#pragma unroll
      for (int jj = 0; jj < KS; ++jj)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ci = g * KS + jj;
          if (ci >= CIN) continue;
          const f32x4 w = *(const f32x4 *)(wk + ci * COUT);
          acc[0] = fmaf(x[ci], w[0], acc[0]);
          acc[1] = fmaf(x[ci], w[1], acc[1]);
          acc[2] = fmaf(x[ci], w[2], acc[2]);
          acc[3] = fmaf(x[ci], w[3], acc[3]);
        }
    }
    f32x4 v = acc;
    if (a.bias) v += *(const f32x4 *)(a.bias + c4);
    if (a.scale) v = v * *(const f32x4 *)(a.scale + c4) + *(const f32x4 *)(a.shift + c4);
    else if (a.shift) v += *(const f32x4 *)(a.shift + c4);
    const size_t o = (size_t)row * COUT + c4;
    if (a.residual) v += *(const f32x4 *)(a.residual + o);
    if (a.relu) {
      v[0] = fmaxf(v[0], 0.f);
      v[1] = fmaxf(v[1], 0.f);
      v[2] = fmaxf(v[2], 0.f);
      v[3] = fmaxf(v[3], 0.f);
    }
    *(f32x4 *)(a.out + o) = v;
  }
}

static int num_cu_small() {
  static int n = 0;
  if (!n) {
    hipDeviceProp_t p;
    n = (hipGetDeviceProperties(&p, 0) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
  }
  return n;
}

template <int CIN, int COUT>
static int launch_small(const ConvArgs &a, hipStream_t stream) {
  constexpr int RPB0 = 256 / (COUT / 4);
  const size_t lds = (size_t)a.K * (CIN * COUT + 32) * 4 + (size_t)RPB0 * DF3D_MAX_KVOL * 4;
  static bool configured = false;
  if (!configured && lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void *)spconv_small_kernel<CIN, COUT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(DF3D_MAX_KVOL * (CIN * COUT + 32) * 4 + RPB0 * DF3D_MAX_KVOL * 4));
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS) failed: %s", hipGetErrorString(e));
      return DF3D_EHIP;
    }
    configured = true;
  }
  constexpr int RPB = 256 / (COUT / 4);
  const int per_cu = lds > 40 * 1024 ? 2 : 4;      // resident workgroups per CU (LDS-limited)
  const int nblk = cdiv(a.n_out, RPB), grid = nblk < num_cu_small() * per_cu ? nblk : num_cu_small() * per_cu;
  hipLaunchKernelGGL((spconv_small_kernel<CIN, COUT>), dim3(grid), dim3(256), lds, stream, a);
  return 1;
}

// ---------------------------------------------------------------------------------------------------------
// Round 4: the same layers from ROW LISTS.  The table nbr[K][N] of these layers is 89-94 % empty (conv1: ~3 of 27 offsets per
// row, the stride-2 layer behind it: ~1.5), yet every row read all K entries -- 27 loads per lane and a 27-step compaction
// for 2-3 gathers: 4-9 MB of indices per launch against 0.2-0.5 MB of present pairs, 12-13 us (16 -> 16) and 33 us
// (16 -> 32) per launch at 0.03-0.06 of the HBM roof (profiles/r04_pmc_by_kernel.json: waves 64-70 % waiting).  The lists
// are the table's present entries per row, offsets ascending, packed (offset << 26 | input row), rows back to back:
// off[N + 1] u32 + ent[pairs] u32 (df3d_nbr_row_lists: count, scan, fill -- in the geometry phase, i.e. on the frame head's
// stream a frame ahead).  Same products in the same order as spconv_small_kernel: bit-identical results.
typedef unsigned int u32x2s __attribute__((ext_vector_type(2)));
struct RowLists {
  const uint32_t *off;   // [n_out + 1]
  const uint32_t *ent;   // [off[n_out]]
};
static thread_local RowLists g_lists_hint = {nullptr, nullptr};   // set around sparse_conv_impl by df3d_sparse_conv_fused_lists
static thread_local void *g_lists_split = nullptr;                // ... with the split rows the caller wants next to `out`
static thread_local bool g_lists_split_done = false;

template <int CIN, int COUT, int NT>
__global__ __launch_bounds__(NT) void spconv_small_lists_kernel(ConvArgs a, RowLists L) {
  constexpr int LPR = COUT / 4;                   // lanes per row
  constexpr int RPB = NT / LPR;                   // rows per workgroup pass
  constexpr int CINP = CIN <= 8 ? 8 : 16;
  constexpr int KS = CINP / 4;
  constexpr int WSTR = CIN * COUT + 32;
  extern __shared__ float Wl[];                   // [K][WSTR] filters
  const int tid = threadIdx.x;
  const int c4 = (tid % LPR) * 4, rl = tid / LPR;
  // the first row's list bounds travel while the filter bank is staged
  int row = blockIdx.x * RPB + rl;
  uint32_t b = 0u, e = 0u;
  if (row < a.n_out) b = L.off[row], e = L.off[row + 1];
  {
    // filter bank -> LDS, eight 16-byte loads in flight per thread (one load per loop trip made the staging a chain of
    // 7-14 dependent L2 round trips: most of the launch for a workgroup that owns 32-64 rows)
    const int total = a.K * CIN * COUT / 4;
    for (int b0 = 0; b0 < total; b0 += NT * 8) {
      f32x4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = b0 + u * NT + tid;
        t[u] = ((const f32x4 *)a.w)[i < total ? i : 0];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = b0 + u * NT + tid;
        if (i < total) {
          const int k = i / (CIN * COUT / 4), r = i - k * (CIN * COUT / 4);
          *(f32x4 *)(Wl + (size_t)k * WSTR + r * 4) = t[u];
        }
      }
    }
  }
  __syncthreads();
  for (; row < a.n_out; row += gridDim.x * RPB) {
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint32_t pk = b < e ? L.ent[b] : 0u;
    // the next pass's bounds (independent of this row's gathers)
    const int nrow = row + gridDim.x * RPB;
    uint32_t nb = 0u, ne = 0u;
    if (nrow < a.n_out) nb = L.off[nrow], ne = L.off[nrow + 1];
    for (uint32_t j = b; j < e; ++j) {
      const uint32_t cur = pk;
      if (j + 1 < e) pk = L.ent[j + 1];           // the next entry while this one's row is gathered
      const int k = cur >> 26;
      const float *f = a.feat + (size_t)(cur & 0x3ffffffu) * CIN;
      float x[CINP];
      if constexpr (CIN % 4 == 0) {
#pragma unroll
        for (int q = 0; q < CIN / 4; ++q) {
          const f32x4 v = *(const f32x4 *)(f + q * 4);
          x[q * 4] = v[0], x[q * 4 + 1] = v[1], x[q * 4 + 2] = v[2], x[q * 4 + 3] = v[3];
        }
      } else {
#pragma unroll
        for (int ci = 0; ci < CINP; ++ci) x[ci] = ci < CIN ? f[ci] : 0.f;
      }
      const float *wk = Wl + (size_t)k * WSTR + c4;
#pragma unroll
      for (int jj = 0; jj < KS; ++jj)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ci = g * KS + jj;
          if (ci >= CIN) continue;
          const f32x4 w = *(const f32x4 *)(wk + ci * COUT);
          acc[0] = fmaf(x[ci], w[0], acc[0]);
          acc[1] = fmaf(x[ci], w[1], acc[1]);
          acc[2] = fmaf(x[ci], w[2], acc[2]);
          acc[3] = fmaf(x[ci], w[3], acc[3]);
        }
    }
    f32x4 v = acc;
    if (a.bias) v += *(const f32x4 *)(a.bias + c4);
    if (a.scale) v = v * *(const f32x4 *)(a.scale + c4) + *(const f32x4 *)(a.shift + c4);
    else if (a.shift) v += *(const f32x4 *)(a.shift + c4);
    const size_t o = (size_t)row * COUT + c4;
    if (a.residual) v += *(const f32x4 *)(a.residual + o);
    if (a.relu) {
      v[0] = fmaxf(v[0], 0.f);
      v[1] = fmaxf(v[1], 0.f);
      v[2] = fmaxf(v[2], 0.f);
      v[3] = fmaxf(v[3], 0.f);
    }
    *(f32x4 *)(a.out + o) = v;
    if (a.out_split) {
      // the operand split the next (matrix-core) layer reads: 8-channel block = [hi 16 B | lo 16 B], this lane owns 4 of the 8
      unsigned h0, l0, h1, l1;
      split_pair(v[0], v[1], h0, l0);
      split_pair(v[2], v[3], h1, l1);
      char *blk = (char *)a.out_split + (o >> 3) * 32 + ((o >> 2) & 1) * 8;
      *(u32x2s *)blk = (u32x2s){h0, h1};
      *(u32x2s *)(blk + 16) = (u32x2s){l0, l1};
    }
    b = nb, e = ne;
  }
}

template <int CIN, int COUT, int NT>
static int launch_small_lists(const ConvArgs &a, const RowLists &L, hipStream_t stream) {
  constexpr int RPB = NT / (COUT / 4);
  const size_t lds = (size_t)a.K * (CIN * COUT + 32) * 4;
  static bool configured = false;
  if (!configured && lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void *)spconv_small_lists_kernel<CIN, COUT, NT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(DF3D_MAX_KVOL * (CIN * COUT + 32) * 4));
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS) failed: %s", hipGetErrorString(e));
      return DF3D_EHIP;
    }
    configured = true;
  }
  const int per_cu = lds > 40 * 1024 ? 2 : 4;      // resident workgroups per CU (LDS-limited)
  const int nblk = cdiv(a.n_out, RPB), grid = nblk < num_cu_small() * per_cu ? nblk : num_cu_small() * per_cu;
  hipLaunchKernelGGL((spconv_small_lists_kernel<CIN, COUT, NT>), dim3(grid), dim3(NT), lds, stream, a, L);
  return 1;
}

static bool small_shape(int cin, int cout) {
  return (cin == 16 && (cout == 16 || cout == 32)) || ((cin == 5 || cin == 4) && cout == 16);
}

// -> 1 served, 0 not a small shape, < 0 error
static int dispatch_small_lists(const ConvArgs &a, const RowLists &L, hipStream_t stream) {
  static const bool off = getenv("DF3D_SMALL_CONV") && getenv("DF3D_SMALL_CONV")[0] == '0';
  if (off || a.n_out < 2048 || !L.off || !L.ent) return 0;
  static const int nt = getenv("DF3D_SMALL_THREADS") ? atoi(getenv("DF3D_SMALL_THREADS")) : 0;     // tuning aid
  if (a.cin == 16 && a.cout == 16) return nt == 512 ? launch_small_lists<16, 16, 512>(a, L, stream) : launch_small_lists<16, 16, 256>(a, L, stream);
  if (a.cin == 5 && a.cout == 16) return launch_small_lists<5, 16, 256>(a, L, stream);
  if (a.cin == 4 && a.cout == 16) return launch_small_lists<4, 16, 256>(a, L, stream);
  if (a.cin == 16 && a.cout == 32) {
    // 54 KB of filters allow two workgroups per CU: at 256 threads that is two waves per SIMD for a kernel that is a chain
    // of memory round trips per row
    // (measured on MI355X, 83 k output rows: 256 threads 24.3 us, 512 threads 16.0 us, 1024 threads 20.8 us)
    if (nt == 256) return launch_small_lists<16, 32, 256>(a, L, stream);
    if (nt == 1024) return launch_small_lists<16, 32, 1024>(a, L, stream);
    return launch_small_lists<16, 32, 512>(a, L, stream);
  }
  return 0;
}

// -> 1 served, 0 not a small shape, < 0 error
static int dispatch_small(const ConvArgs &a, hipStream_t stream) {
  static const bool off = getenv("DF3D_SMALL_CONV") && getenv("DF3D_SMALL_CONV")[0] == '0';
  if (off || a.n_out < 2048) return 0;
  if (a.cin == 16 && a.cout == 16) return launch_small<16, 16>(a, stream);
  if (a.cin == 5 && a.cout == 16) return launch_small<5, 16>(a, stream);
  if (a.cin == 4 && a.cout == 16) return launch_small<4, 16>(a, stream);
  if (a.cin == 16 && a.cout == 32) return launch_small<16, 32>(a, stream);
  return 0;
}

template <int CINP, int COUT, int KC, bool VEC>
static void launch_mfma(const ConvArgs &a, hipStream_t stream, int groups = 1) {
  // small layers: 64-row tiles so that the grid still covers the 256 CUs several times
  // (grouped launches of the narrow final convs of the detection heads: 64 / 128 / 256 rows per workgroup measured
  // within 5 % of each other on MI355X -- 358 .. 383 us for 36 groups of 64 -> 16 on 32 400 rows)
  const int rt = (long long)a.n_out * groups < 128 * 1024 ? 1 : 2;
  if (rt == 1) {
    hipLaunchKernelGGL((spconv_mfma_kernel<CINP, COUT, KC, 1, VEC>), dim3(cdiv(a.n_out, 64), groups), dim3(256), 0, stream, a);
  } else {
    hipLaunchKernelGGL((spconv_mfma_kernel<CINP, COUT, KC, 2, VEC>), dim3(cdiv(a.n_out, 128), groups), dim3(256), 0, stream, a);
  }
}

template <int COUT>
static bool dispatch_cin(const ConvArgs &a, hipStream_t stream, int groups = 1) {
  if (a.cin <= 8) {
    if (groups > 1) return false;
    if (a.cin % 4 == 0) launch_mfma<8, COUT, 8, false>(a, stream);
    else launch_mfma<8, COUT, 8, false>(a, stream);
    return true;
  }
  // the filter tile of a step is KC x COUT floats, double buffered: 16-channel slices keep 256 columns inside 64 KB
  constexpr int KW = COUT > 128 ? 16 : 32;
  switch (a.cin) {
    case 16: launch_mfma<16, COUT, 16, true>(a, stream, groups); return true;
    case 32: launch_mfma<32, COUT, KW, true>(a, stream, groups); return true;
    case 64: launch_mfma<64, COUT, KW, true>(a, stream, groups); return true;
    case 128: launch_mfma<128, COUT, KW, true>(a, stream, groups); return true;
    // BEV neck / detection heads in the exact-fp32 mode (round 3): 256- and 512-channel inputs
    case 256: launch_mfma<256, COUT, KW, true>(a, stream, groups); return true;
    case 512: launch_mfma<512, COUT, KW, true>(a, stream, groups); return true;
    default: return false;
  }
}


// ---- pair-balanced row ranges for the pair kernel -------------------------------------------
__global__ __launch_bounds__(256) void conv_rowcount_kernel(const int32_t *__restrict__ nbr, int K, int n_out,
                                                            uint32_t *__restrict__ cnt) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_out) return;
  uint32_t c = 0;
  for (int k = 0; k < K; ++k) c += nbr[(size_t)k * n_out + r] >= 0 ? 1u : 0u;
  cnt[r] = c + 1u;   // +1: a row costs something even without neighbours (epilogue), and keeps ranges non-empty
}

__global__ __launch_bounds__(256) void conv_tiles_kernel(const uint32_t *__restrict__ prefix,
                                                         const uint32_t *__restrict__ total, int n_out, int ntiles,
                                                         int32_t *__restrict__ tile_rows) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > ntiles) return;
  if (i == ntiles) {
    tile_rows[i] = n_out;
    return;
  }
  unsigned long long target = (unsigned long long)(*total) * (unsigned long long)i / (unsigned long long)ntiles;
  int lo = 0, hi = n_out;   // first row whose exclusive prefix >= target
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if ((unsigned long long)prefix[mid] < target) lo = mid + 1;
    else hi = mid;
  }
  tile_rows[i] = lo;
}


// ---- optional per-launch timing with HIP events recorded right around the launch (bench.py's roofline
//      leg; hipEventRecord on the launching stream, microseconds apart on the host) ----------------------
struct TimingRec {
  hipEvent_t e0, e1;
  int cin, cout, kvol, n_out;
  long long pairs;       // valid rulebook entries, counted only in pair-count mode (df3d_timing_count_pairs)
  int split;             // 1 = split-precision kernel
};
static bool g_timing_pairs = false;
static unsigned long long *g_pair_counter = nullptr;

__global__ __launch_bounds__(256) void count_valid_kernel(const int32_t *__restrict__ nbr, size_t n,
                                                          unsigned long long *__restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool v = i < n && nbr[i] >= 0;
  unsigned long long b = __ballot(v);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(out, (unsigned long long)__popcll(b));
}
static bool g_timing_on = false;
static int g_timing_filter[3] = {0, 0, 0};   // (cin, cout, kvol) of the only launches to time; 0 = any
static int g_timing_every = 1;               // df3d_timing_sample: events around every N-th matching launch only
static unsigned g_timing_seen = 0;
static std::mutex g_timing_mu;          // several host threads (frames in flight) launch convolutions concurrently
static std::vector<TimingRec> g_timing;
static std::vector<hipEvent_t> g_event_pool;

static hipEvent_t timing_event() {
  if (!g_event_pool.empty()) {
    hipEvent_t e = g_event_pool.back();
    g_event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

int timing_rec_begin(int cin, int cout, int kvol, int n_out, const int32_t *nbr, int split, hipStream_t stream) {
  if (!g_timing_on) return -1;
  if ((g_timing_filter[0] && g_timing_filter[0] != cin) || (g_timing_filter[1] && g_timing_filter[1] != cout) ||
      (g_timing_filter[2] && g_timing_filter[2] != kvol))
    return -1;
  std::lock_guard<std::mutex> lock(g_timing_mu);
  if (g_timing_every > 1 && (g_timing_seen++ % (unsigned)g_timing_every) != 0) return -1;
  TimingRec r = {timing_event(), timing_event(), cin, cout, kvol, n_out, -1, split};
  if (!r.e0 || !r.e1) return -1;
  if (g_timing_pairs && nbr) {
    // metadata pass (never inside a timed region): count the valid (output, offset) pairs of this launch
    if (!g_pair_counter && hipMalloc((void **)&g_pair_counter, 8) != hipSuccess) g_pair_counter = nullptr;
    if (g_pair_counter) {
      size_t tot = (size_t)kvol * n_out;
      unsigned long long h = 0;
      (void)hipMemsetAsync(g_pair_counter, 0, 8, stream);
      hipLaunchKernelGGL(count_valid_kernel, dim3(cdiv((long long)tot, 256)), dim3(256), 0, stream, nbr, tot,
                         g_pair_counter);
      (void)hipMemcpyAsync(&h, g_pair_counter, 8, hipMemcpyDeviceToHost, stream);
      (void)hipStreamSynchronize(stream);
      r.pairs = (long long)h;
    }
  }
  (void)hipEventRecord(r.e0, stream);
  g_timing.push_back(r);
  return (int)g_timing.size() - 1;
}

void timing_rec_end(int rec, hipStream_t stream) {
  if (rec < 0) return;
  std::lock_guard<std::mutex> lock(g_timing_mu);
  if (rec < (int)g_timing.size()) (void)hipEventRecord(g_timing[rec].e1, stream);
}

static int pair_ntiles(int n_out, int cin, int cout) {
  // measured on MI355X (tools/conv_probe.py): the pair kernel wins for COUT=128 (244 vs 306 us at conv4),
  // the output-stationary kernel for COUT=64 (170 vs 218 us at conv3: items are only 32 MFMAs long there)
  bool ok = cout == 128 && (cin == 128 || cin == 64);
  if (getenv("DF3D_PAIR_ALL")) ok = ok || (cout == 64 && (cin == 64 || cin == 32));
  if (!ok || n_out <= 0) return 0;
  static int num_cu = 0;
  if (!num_cu) {
    hipDeviceProp_t p;
    num_cu = (hipGetDeviceProperties(&p, 0) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
  }
  return num_cu;   // one workgroup per CU
}

}  // namespace df3d

using namespace df3d;

extern "C" int df3d_conv_tile_count(int n_out, int cin, int cout, int kvol) {
  (void)kvol;
  // shapes served by the one-workgroup-per-CU pair-compacted kernels (spconv_pair_kernel, spconv_split_kernel);
  // the output-stationary kernels cut equal tiles and need no ranges
  bool ok = (cout == 128 && (cin == 128 || cin == 64)) || (cout == 64 && (cin == 64 || cin == 32));
  if (!ok || n_out <= 0) return 0;
  return pair_ntiles(n_out, 128, 128);
}

extern "C" size_t df3d_conv_tiles_workspace_bytes(int n_out) {
  size_t b = 0;
  b = arena_need(b, (size_t)(n_out > 0 ? n_out : 1) * 4);
  b = arena_need(b, 64);
  b = arena_need(b, scan_scratch_bytes((size_t)(n_out > 0 ? n_out : 1)));
  return b + 256;
}

extern "C" int df3d_conv_tiles(const int32_t *nbr, int kvol, int n_out, int ntiles, int32_t *tile_rows,
                               void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(nbr && tile_rows && ntiles > 0 && n_out > 0 && kvol > 0, "conv_tiles: bad arguments");
  Arena ar(workspace, workspace_bytes);
  uint32_t *cnt = ar.take<uint32_t>(n_out);
  uint32_t *total = ar.take<uint32_t>(16);
  size_t ssz = scan_scratch_bytes((size_t)n_out);
  void *sscr = ar.take<char>(ssz);
  if (!sscr) {
    set_error("conv_tiles: workspace too small");
    return DF3D_ENOMEM;
  }
  hipLaunchKernelGGL(conv_rowcount_kernel, dim3(cdiv(n_out, 256)), dim3(256), 0, stream, nbr, kvol, n_out, cnt);
  int rc = exclusive_scan_u32(cnt, cnt, (size_t)n_out, total, sscr, ssz, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(conv_tiles_kernel, dim3(cdiv(ntiles + 1, 256)), dim3(256), 0, stream, cnt, total, n_out, ntiles,
                     tile_rows);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

struct ConvGroups;
static int sparse_conv_impl(const float *features, int n_in, int cin, const float *filters, int kvol, int cout,
                            const int32_t *nbr, int n_out, const float *bias, const float *scale, const float *shift,
                            const float *residual, int relu, float *out, const int32_t *tile_rows, int ntiles,
                            void *stream_, const ConvGroups *cg = nullptr);

struct ConvGroups {
  int groups, ld_in, group_in, ld_out, group_out;
};

extern "C" int df3d_sparse_conv_grouped(const float *features, int n_in, int cin, int ld_in, int group_in, const float *filters,
                                        int kvol, int cout, int groups, const int32_t *nbr, int n_out, const float *bias,
                                        const float *scale, const float *shift, int relu, float *out, int ld_out, int group_out,
                                        void *stream_) {
  DF3D_CHECK_ARG(groups >= 1 && groups <= 65535 && ld_in >= cin && ld_in % 4 == 0 && group_in % 4 == 0 && group_in >= 0 &&
                     ld_out >= cout && group_out >= 0,
                 "sparse_conv_grouped: bad strides (groups %d, ld_in %d, group_in %d, ld_out %d, group_out %d)", groups, ld_in,
                 group_in, ld_out, group_out);
  DF3D_CHECK_ARG(cin >= 16 && (cin & (cin - 1)) == 0 && cin <= 512 &&
                     (cout == 16 || cout == 32 || cout == 64 || cout == 128 || cout == 256),
                 "sparse_conv_grouped: served by the MFMA kernel for cin 16..512 (power of two), cout 16..256; got %d -> %d", cin,
                 cout);
  DF3D_CHECK_ARG((groups - 1) * (long long)group_in + cin <= ld_in && (groups - 1) * (long long)group_out + cout <= ld_out,
                 "sparse_conv_grouped: the slices of %d groups do not fit the rows", groups);
  const ConvGroups cg = {groups, ld_in, group_in, ld_out, group_out};
  return sparse_conv_impl(features, n_in, cin, filters, kvol, cout, nbr, n_out, bias, scale, shift, nullptr, relu, out,
                          nullptr, 0, stream_, &cg);
}

extern "C" int df3d_sparse_conv_fused(const float *features, int n_in, int cin, const float *filters, int kvol,
                                      int cout, const int32_t *nbr, int n_out, const float *bias, const float *scale,
                                      const float *shift, const float *residual, int relu, float *out,
                                      void *stream_) {
  return sparse_conv_impl(features, n_in, cin, filters, kvol, cout, nbr, n_out, bias, scale, shift, residual, relu, out,
                          nullptr, 0, stream_);
}

extern "C" int df3d_sparse_conv_fused_tiled(const float *features, int n_in, int cin, const float *filters, int kvol,
                                            int cout, const int32_t *nbr, int n_out, const float *bias,
                                            const float *scale, const float *shift, const float *residual, int relu,
                                            float *out, const int32_t *tile_rows, int ntiles, void *stream_) {
  return sparse_conv_impl(features, n_in, cin, filters, kvol, cout, nbr, n_out, bias, scale, shift, residual, relu, out,
                          tile_rows, ntiles, stream_);
}

// ---- row lists of a neighbour table (round 4; see spconv_small_lists_kernel) ----
__global__ __launch_bounds__(256) void nbr_row_count_kernel(const int32_t *__restrict__ nbr, int K, int n_out,
                                                            uint32_t *__restrict__ cnt) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n_out) return;
  uint32_t c = 0;
  if (r < n_out)
    for (int k = 0; k < K; ++k) c += nbr[(size_t)k * n_out + r] >= 0 ? 1u : 0u;
  cnt[r] = c;                                      // cnt[n_out] = 0: the scan then leaves the total in off[n_out]
}

__global__ __launch_bounds__(256) void nbr_row_fill_kernel(const int32_t *__restrict__ nbr, int K, int n_out,
                                                           const uint32_t *__restrict__ off, uint32_t *__restrict__ ent) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_out) return;
  uint32_t j = off[r];
  for (int k = 0; k < K; ++k) {
    const int v = nbr[(size_t)k * n_out + r];
    if (v >= 0) ent[j++] = (uint32_t)v | ((uint32_t)k << 26);
  }
}

static size_t lists_off_bytes(int n_out) { return align_up(((size_t)n_out + 1) * 4, 256); }

extern "C" size_t df3d_nbr_row_lists_bytes(int kvol, int n_out) {
  if (kvol <= 0 || kvol > DF3D_MAX_KVOL || n_out <= 0) return 0;
  // off | ent (capacity: every entry present) | counts | scan scratch
  return lists_off_bytes(n_out) + align_up((size_t)kvol * n_out * 4, 256) + lists_off_bytes(n_out) +
         scan_scratch_bytes((size_t)n_out + 1);
}

extern "C" int df3d_nbr_row_lists(const int32_t *nbr, int kvol, int n_out, int n_in, void *lists, size_t lists_bytes,
                                  void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(nbr && lists, "nbr_row_lists: null argument");
  DF3D_CHECK_ARG(kvol > 0 && kvol <= DF3D_MAX_KVOL && kvol <= 64 && n_out > 0, "nbr_row_lists: bad sizes");
  DF3D_CHECK_ARG(n_in < (1 << 26), "nbr_row_lists: %d input rows do not fit the packed entry (26 bits)", n_in);
  DF3D_CHECK_ARG(lists_bytes >= df3d_nbr_row_lists_bytes(kvol, n_out), "nbr_row_lists: blob of %zu bytes is too small", lists_bytes);
  char *p = (char *)lists;
  uint32_t *off = (uint32_t *)p;
  uint32_t *ent = (uint32_t *)(p + lists_off_bytes(n_out));
  uint32_t *cnt = (uint32_t *)((char *)ent + align_up((size_t)kvol * n_out * 4, 256));
  void *scratch = (char *)cnt + lists_off_bytes(n_out);
  hipLaunchKernelGGL(nbr_row_count_kernel, dim3(cdiv(n_out + 1, 256)), dim3(256), 0, stream, nbr, kvol, n_out, cnt);
  int rc = exclusive_scan_u32(cnt, off, (size_t)n_out + 1, nullptr, scratch, scan_scratch_bytes((size_t)n_out + 1), stream);
  if (rc) return rc;
  hipLaunchKernelGGL(nbr_row_fill_kernel, dim3(cdiv(n_out, 256)), dim3(256), 0, stream, nbr, kvol, n_out, off, ent);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_sparse_conv_fused_lists(const float *features, int n_in, int cin, const float *filters, int kvol,
                                            int cout, const int32_t *nbr, const void *lists, int n_out, const float *bias,
                                            const float *scale, const float *shift, const float *residual, int relu,
                                            float *out, void *out_split, void *stream_) {
  DF3D_CHECK_ARG(!out_split || cout % 8 == 0, "sparse_conv_fused_lists: split rows need a multiple of 8 output channels");
  if (lists && n_out > 0 && small_shape(cin, cout)) {
    const char *p = (const char *)lists;
    g_lists_hint.off = (const uint32_t *)p;
    g_lists_hint.ent = (const uint32_t *)(p + lists_off_bytes(n_out));
    g_lists_split = out_split;
  }
  g_lists_split_done = false;
  int rc = sparse_conv_impl(features, n_in, cin, filters, kvol, cout, nbr, n_out, bias, scale, shift, residual, relu, out,
                            nullptr, 0, stream_);
  g_lists_hint.off = g_lists_hint.ent = nullptr;
  g_lists_split = nullptr;
  if (rc) return rc;
  // the lists kernel did not serve the launch (no lists, another shape, a tiny layer): the split rows as a pass of their own
  if (out_split && !g_lists_split_done && n_out > 0) return df3d_split_rows(out, n_out, cout, out_split, stream_);
  return rc;
}

static int sparse_conv_impl(const float *features, int n_in, int cin, const float *filters, int kvol, int cout,
                            const int32_t *nbr, int n_out, const float *bias, const float *scale, const float *shift,
                            const float *residual, int relu, float *out, const int32_t *tile_rows, int ntiles,
                            void *stream_, const ConvGroups *cg) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(features && filters && nbr && out, "sparse_conv_fused: null argument");
  DF3D_CHECK_ARG(kvol > 0 && kvol <= DF3D_MAX_KVOL, "sparse_conv_fused: kernel volume %d unsupported", kvol);
  DF3D_CHECK_ARG(cin > 0 && cout > 0 && n_in >= 0 && n_out >= 0, "sparse_conv_fused: bad sizes");
  if (n_out == 0) return DF3D_OK;
  ConvArgs a;
  a.feat = features;
  a.w = filters;
  a.nbr = nbr;
  a.bias = bias;
  a.scale = scale;
  a.shift = shift;
  a.residual = residual;
  a.out = out;
  a.n_in = n_in;
  a.n_out = n_out;
  a.K = kvol;
  a.cin = cin;
  a.cout = cout;
  a.relu = relu;
  a.ldi = cg ? cg->ld_in : cin;
  a.ldo = cg ? cg->ld_out : cout;
  a.gi = cg ? cg->group_in : 0;
  a.go = cg ? cg->group_out : 0;
  a.out_split = nullptr;
  const int trec = timing_rec_begin(cin, cout * (cg ? cg->groups : 1), kvol, n_out, nbr, 0, stream);
  bool done = false;
  if (cg) {                                          // column slices of wider rows, groups: the output-stationary MFMA kernel
    switch (cout) {
      case 16: done = dispatch_cin<16>(a, stream, cg->groups); break;
      case 32: done = dispatch_cin<32>(a, stream, cg->groups); break;
      case 64: done = dispatch_cin<64>(a, stream, cg->groups); break;
      case 128: done = dispatch_cin<128>(a, stream, cg->groups); break;
      case 256: done = dispatch_cin<256>(a, stream, cg->groups); break;
      default: break;
    }
    DF3D_CHECK_ARG(done, "sparse_conv_grouped: shape %d -> %d not served", cin, cout);
    timing_rec_end(trec, stream);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
  }
  {
    int r = 0;
    if (g_lists_hint.off) {                        // row lists given (df3d_sparse_conv_fused_lists)
      a.out_split = g_lists_split;
      r = dispatch_small_lists(a, g_lists_hint, stream);
      if (r == 1) g_lists_split_done = true;
      a.out_split = nullptr;
    }
    if (r == 0) r = dispatch_small(a, stream);     // C <= 16 input channels: vector-ALU kernel
    if (r < 0) return r;
    done = r == 1;
  }
  // compute-bound shapes: pair-compacted kernel (DF3D_SPCONV_V1=1 forces the output-stationary kernel)
  static const bool use_v2 = getenv("DF3D_SPCONV_V1") == nullptr;
  if (use_v2 && !done) {
    int r = dispatch_pair(a, tile_rows, ntiles, stream);
    if (r < 0) return r;
    done = r == 1;
  }
  if (!done) switch (cout) {
    case 16: done = dispatch_cin<16>(a, stream); break;
    case 32: done = dispatch_cin<32>(a, stream); break;
    case 64: done = dispatch_cin<64>(a, stream); break;
    case 128: done = dispatch_cin<128>(a, stream); break;
    case 256: done = a.cin >= 16 && dispatch_cin<256>(a, stream); break;
    default: break;
  }
  if (!done) {
    size_t tot = (size_t)n_out * cout;
    hipLaunchKernelGGL(spconv_generic_kernel, dim3(cdiv((long long)tot, 256)), dim3(256), 0, stream, a);
  }
  timing_rec_end(trec, stream);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_timing_begin(void) {
  std::lock_guard<std::mutex> lock(g_timing_mu);
  for (auto &r : g_timing) {
    g_event_pool.push_back(r.e0);
    g_event_pool.push_back(r.e1);
  }
  g_timing.clear();
  g_timing_on = true;
  return DF3D_OK;
}

extern "C" int df3d_timing_filter(int cin, int cout, int kvol) {
  g_timing_filter[0] = cin;
  g_timing_filter[1] = cout;
  g_timing_filter[2] = kvol;
  return DF3D_OK;
}

extern "C" int df3d_timing_count_pairs(int on) {
  g_timing_pairs = on != 0;
  return DF3D_OK;
}

extern "C" int df3d_timing_get2(int i, int *shape4, float *ms, long long *pairs, int *split) {
  DF3D_CHECK_ARG(i >= 0 && i < (int)g_timing.size() && shape4 && ms && pairs && split, "timing_get2: bad index");
  const TimingRec &r = g_timing[i];
  DF3D_HIP(hipEventSynchronize(r.e1));
  DF3D_HIP(hipEventElapsedTime(ms, r.e0, r.e1));
  shape4[0] = r.cin;
  shape4[1] = r.cout;
  shape4[2] = r.kvol;
  shape4[3] = r.n_out;
  *pairs = r.pairs;
  *split = r.split;
  return DF3D_OK;
}

// Events around every `every`-th launch that passes the filter (1 = all): a pair of event records costs the launching stream
// a marker packet each and breaks back-to-back dispatch -- around the four launches per frame of the dominant kernel that was
// ~5 % of the timed step (3.00 against 2.85 ms); a sample of them measures the same average.
extern "C" int df3d_timing_sample(int every) {
  g_timing_every = every > 1 ? every : 1;
  g_timing_seen = 0;
  return DF3D_OK;
}

extern "C" int df3d_timing_end(void) {
  g_timing_on = false;
  return (int)g_timing.size();
}

extern "C" int df3d_timing_get(int i, int *shape4, float *ms) {
  DF3D_CHECK_ARG(i >= 0 && i < (int)g_timing.size() && shape4 && ms, "timing_get: bad index");
  const TimingRec &r = g_timing[i];
  DF3D_HIP(hipEventSynchronize(r.e1));
  DF3D_HIP(hipEventElapsedTime(ms, r.e0, r.e1));
  shape4[0] = r.cin;
  shape4[1] = r.cout;
  shape4[2] = r.kvol;
  shape4[3] = r.n_out;
  return DF3D_OK;
}
