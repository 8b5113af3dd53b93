// Output-stationary split-precision sparse convolution, third structure (round 3): INPUT ROWS STAGED THROUGH LDS.
// Included by spconv_split.hip (shares SplitConvArgs, the packed-weight layout 1, split rows, the MFMA macro).
//
// What rounds 1-2 measured about the two kernels above: a (offset, 32-channel) step costs ~1400 clocks whatever its
// matrix work, because every step GATHERS its A operands from L2 -- a fragment-shaped register gather is 64 line
// requests per wave instruction with every 128-byte line requested twice, an LDS-DMA of whole lines is bound by the
// loader waves' issue rate -- and every input row is fetched once per rulebook PAIR: ~12-18 times per layer.
//
// Here a workgroup owns TM = 64 * RT consecutive output rows (4 waves, 16 * RT rows x all columns each) and exploits
// that the rows of every stage are sorted by flat cell index (strided outputs are emitted that way; the reference sorts
// them too, spconv_ops.h:119-137): the neighbours of consecutive output rows at ONE kernel offset are (nearly)
// consecutive input rows, and the nine offsets of one kz plane (K = 27) -- or all nine taps of a dense 3 x 3 layer
// (K = 9) -- together touch one CONTIGUOUS RANK RANGE of input rows that is only ~1.3 x TM long (measured on the
// nuScenes stages: 540-640 rows loaded per 128-row tile against 1250-2240 rulebook pairs).  Per (32-channel block,
// offset group):
//   * the range [lo, hi] of the group (a min / max over the tile's neighbour table, computed once per tile) is loaded
//     as whole 128-byte lines -- perfectly coalesced, every row once per group instead of once per pair (x2.3-3.5 less
//     L2 traffic, x16 fewer memory requests) -- into registers while the previous group computes, then into LDS;
//   * the nine steps of the group read their A fragments from LDS at slot = neighbour - lo (slots are XOR-swizzled by
//     (slot >> 1) & 3, so that eight consecutive slots -- the common case -- read conflict-free; rows without a neighbour
//     read an all-zero slot);
//   * the packed filter tile of every step goes global -> registers (three sets, fetched five steps ahead) -> a ring of
//     three LDS tiles; one barrier per step;
//   * with 132-160 KB of LDS there is ONE workgroup = one wave per SIMD on a CU, so nothing hides a wave's own latencies:
//     the A and B fragments of step s + 1 are read into a second register set while the matrix instructions of step s
//     run (first version without this: MFMA, LDS and staging times simply added up, 2.3 x slower than round 2).
// A group whose range exceeds HCAP rows (a tile in a sparse plane under a dense one: < 1 % of the groups) is processed in
// several chunks of its range (the offsets of the group run once per chunk); a group without any neighbour is skipped.
// Correctness does not depend on the row order (an unsorted table only makes ranges long); results differ from the
// other kernels in summation order only (channel block outermost).
//
// Per step and workgroup: MFMA 4 waves x RT x CT x 3 instructions of 16 clocks on their own SIMD; LDS B fragments
// 4 x CT x 2 KB + A fragments 4 x RT x 2 KB + the filter tile CT x 2 KB.  RT = 4 makes the 64- and 32-column layers
// matrix-bound (768 / 384 MFMA clocks against 576 / 416 LDS clocks), RT = 2 balances the 128-column layers (768 : 768).

#ifdef DF3D_HALO_EXPERIMENTS            // run-time ablation flags (DF3D_OS_DBG): 1 no A reads, 2 no filter staging, 4 no MFMAs, 8 no staging of rows
#define HALO_DBG(bit) ((a.dbg & (bit)) != 0)
#else
#define HALO_DBG(bit) false
#endif

template <int CIN, int COUT, int RT, int HCAP>
__global__ __launch_bounds__(256) void spconv_halo_kernel(SplitConvArgs a) {
  constexpr int CW = COUT > 128 ? 128 : COUT;
  constexpr int KB = CIN / 32, CT = CW / 16, TM = 64 * RT, WROWS = 16 * RT;
  constexpr int WQ = CT * 2 * 64;                 // u32x4 per (offset, 32-channel block) filter tile
  constexpr int WPT = (WQ + 255) / 256;
  constexpr int HPT = HCAP * 8 / 256;             // u32x4 of one staged range per thread
  constexpr int GK = 9;                           // offsets per group
  static_assert(HCAP % 32 == 0 && WQ % 256 == 0, "tile shape");
  extern __shared__ __align__(16) unsigned char halo_smem[];
  u32x4 *Hl = (u32x4 *)halo_smem;                 // [(HCAP + 1)][8]: staged rows, slot HCAP = zeros
  u32x4 *Wl = Hl + (HCAP + 1) * 8;                // [3][WQ]: ring of filter tiles
  int *nbrL = (int *)(Wl + 3 * WQ);               // [K <= 27][TM]
  int *rowL = nbrL + 27 * TM;                     // [TM]
  int *grp = rowL + TM;                           // [3][2]: lo, hi of a group's rank range

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int row0 = tile * TM;
  const int col0 = blockIdx.y * CW;
  const int NG = a.K / GK;                        // 3 (K = 27: one group per kz) or 1 (K = 9)
  const u32x4 *featb = a.feat + (size_t)blockIdx.y * a.in_goff;

  OS_STAMP(0);
  if (tid < 6) grp[tid] = (tid & 1) ? -1 : 0x7fffffff;
  if (tid < 8) Hl[HCAP * 8 + tid] = (u32x4){0u, 0u, 0u, 0u};
  __syncthreads();
  {
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-1, -1, -1};
    // all loads of the tile's neighbour table first (one round trip), then the LDS stores and the range reduction
    constexpr int KSTEP = 256 / TM, NPASS = (27 + KSTEP - 1) / KSTEP;
    static_assert(256 % TM == 0, "tile height");
    const int r = tid % TM, k0 = tid / TM;
    int row = row0 + r;
    row = row < a.n_out ? (a.order ? a.order[row] : row) : a.n_out;
    if (k0 == 0) rowL[r] = row;
    int v[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int k = k0 + i * KSTEP;
      v[i] = (k < a.K && row < a.n_out) ? a.nbr[(size_t)k * a.n_out + row] : -1;
    }
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int k = k0 + i * KSTEP;
      if (k < a.K) nbrL[k * TM + r] = v[i];
      if (v[i] >= 0) {
        const int gi = k / GK;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          if (gi == q) {
            lo[q] = v[i] < lo[q] ? v[i] : lo[q];
            hi[q] = v[i] > hi[q] ? v[i] : hi[q];
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const int ol = __shfl_xor(lo[q], off, 64), oh = __shfl_xor(hi[q], off, 64);
        lo[q] = ol < lo[q] ? ol : lo[q];
        hi[q] = oh > hi[q] ? oh : hi[q];
      }
      if (lane == 0 && hi[q] >= 0) {
        atomicMin(&grp[2 * q], lo[q]);
        atomicMax(&grp[2 * q + 1], hi[q]);
      }
    }
  }
  __syncthreads();
  OS_STAMP(1);
  int glo[3], gcnt[3];
  unsigned gm = 0u;                               // groups with at least one neighbour
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    glo[q] = __builtin_amdgcn_readfirstlane(grp[2 * q]);
    const int h = __builtin_amdgcn_readfirstlane(grp[2 * q + 1]);
    gcnt[q] = h >= 0 ? h - glo[q] + 1 : 0;
    if (q < NG && h >= 0) gm |= 1u << q;
  }

  f32x4 acc[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // One ITEM = (channel block, group, chunk of the group's range): rows [lo + chunk * HCAP, ...) go to LDS and the nine
  // offsets of the group run against them; a neighbour outside the chunk reads the zero slot.  99 % of the groups are one
  // chunk; a long range (a tile in a sparse plane next to a dense one) simply takes several passes over its offsets, so the
  // step loop knows LDS operands only.  Items are addressed arithmetically: with one wave per SIMD every scalar instruction
  // of a step is exposed (a cursor advanced per step cost more than the step's matrix instructions).
  int gch[3];                                     // chunks per group (0: no neighbour at all: the group is skipped)
#pragma unroll
  for (int q = 0; q < 3; ++q) gch[q] = (gcnt[q] + HCAP - 1) / HCAP;
  const int per_cb = gch[0] + gch[1] + gch[2];
  const int items = per_cb * KB;
  // item -> (channel block, group, first row, rows): a dozen scalar instructions per nine steps
  auto item_of = [&](int it, int &cb, int &q, int &base, int &cnt) {
    cb = it / per_cb;
    int r = it - cb * per_cb;
    q = r < gch[0] ? 0 : (r < gch[0] + gch[1] ? 1 : 2);
    r -= q == 0 ? 0 : (q == 1 ? gch[0] : gch[0] + gch[1]);
    const int lo = q == 0 ? glo[0] : (q == 1 ? glo[1] : glo[2]);
    const int n_ = q == 0 ? gcnt[0] : (q == 1 ? gcnt[1] : gcnt[2]);
    base = lo + r * HCAP;
    cnt = n_ - r * HCAP < HCAP ? n_ - r * HCAP : HCAP;
  };
  const int steps = items * GK;

  // ---- filter stream: global -> three register sets -> a ring of three LDS tiles.  Step s stores W(s + 2) and fetches
  //      W(s + 5); the B fragments of step s + 1 are read DURING step s (the tile was stored in step s - 1, the barrier
  //      that ends a step publishes it), so the matrix instructions of a step never wait for LDS ----
  u32x4 wr[3][WPT];
  const u32x4 *wbase = a.w + (size_t)blockIdx.y * a.K * KB * WQ + tid;
  auto load_w = [&](u32x4 (&wreg)[WPT], const u32x4 *src) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) wreg[i] = src[256 * i];
  };
  auto store_w = [&](int buf, u32x4 (&wreg)[WPT]) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) Wl[buf * WQ + tid + 256 * i] = wreg[i];
  };
  // ---- staged ranges ----
  u32x4 hreg[HPT];
  auto halo_fetch = [&](int cb, int base, int cnt) {      // rows [base, base + cnt) of channel block cb -> registers
    const u32x4 *src = featb + cb * 8 + (tid & 7);
#pragma unroll
    for (int i = 0; i < HPT; ++i) {
      int slot = (tid >> 3) + 32 * i;
      slot = slot < cnt ? slot : cnt - 1;          // slots past the range re-read its last row (never used)
      hreg[i] = src[(size_t)(base + slot) * a.ldi];
    }
  };
  auto halo_store = [&]() {
#pragma unroll
    for (int i = 0; i < HPT; ++i) {
      const int slot = (tid >> 3) + 32 * i, unit = tid & 7;
      Hl[slot * 8 + (unit ^ ((slot >> 1) & 3))] = hreg[i];
    }
  };
  // fragments of one step: A from the staged rows at slot = neighbour - base, B from the LDS ring.  The neighbour indices
  // are read one step before the fragments that depend on them (two LDS round trips would otherwise sit in one step).
  u32x4 af[2][RT][2], bf[2][CT * 2];
  int nv[2][RT];
  auto read_nv = [&](int (&dst)[RT], int k) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) dst[rt] = nbrL[k * TM + wave * WROWS + rt * 16 + n];
  };
  auto read_a = [&](u32x4 (&dst)[RT][2], const int (&idx)[RT], int base, int cnt) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const unsigned sl = (unsigned)(idx[rt] - base);
      int slot = (idx[rt] >= 0 && sl < (unsigned)cnt) ? (int)sl : HCAP;
      if (HALO_DBG(1)) slot = HCAP;                // (every lane reads the zero slot: broadcast, no bank conflicts)
      if (HALO_DBG(16)) slot = (wave * WROWS + rt * 16 + n) % HCAP;      // (consecutive slots: the conflict-free pattern)
      const int sw = (slot >> 1) & 3;
      dst[rt][0] = Hl[slot * 8 + ((2 * g) ^ sw)];
      dst[rt][1] = Hl[slot * 8 + ((2 * g + 1) ^ sw)];
    }
  };
  auto read_b = [&](u32x4 (&dst)[CT * 2], int buf) {
    if (HALO_DBG(32)) return;
    const u32x4 *wb = Wl + buf * WQ + lane;
#pragma unroll
    for (int j = 0; j < CT * 2; ++j) dst[j] = wb[j * 64];
  };
  auto mfmas = [&](u32x4 (&A)[RT][2], u32x4 (&B)[CT * 2]) {
    if (HALO_DBG(4)) return;
#pragma unroll
    for (int i = 0; i < CT / 2; ++i) {
      const int c2 = i * 2;
      const u32x4 bh0 = B[c2 * 2], bl0 = B[c2 * 2 + 1], bh1 = B[c2 * 2 + 2], bl1 = B[c2 * 2 + 3];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        acc[rt][c2] = DF3D_MFMA_BF16(A[rt][1], bh0, acc[rt][c2]);
        acc[rt][c2 + 1] = DF3D_MFMA_BF16(A[rt][1], bh1, acc[rt][c2 + 1]);
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        acc[rt][c2] = DF3D_MFMA_BF16(A[rt][0], bl0, acc[rt][c2]);
        acc[rt][c2 + 1] = DF3D_MFMA_BF16(A[rt][0], bl1, acc[rt][c2 + 1]);
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        acc[rt][c2] = DF3D_MFMA_BF16(A[rt][0], bh0, acc[rt][c2]);
        acc[rt][c2 + 1] = DF3D_MFMA_BF16(A[rt][0], bh1, acc[rt][c2 + 1]);
      }
    }
  };

  if (steps > 0) {
    int cb, q, base, cnt;
    item_of(0, cb, q, base, cnt);
    {
      const u32x4 *w0 = wbase + ((size_t)(q * GK) * KB + cb) * WQ;
      load_w(wr[0], w0);                           // W(0), W(1) -> ring; W(2), W(3), W(4) in the register sets
      load_w(wr[1], w0 + (size_t)KB * WQ);
      if (!HALO_DBG(8)) halo_fetch(cb, base, cnt);
      store_w(0, wr[0]);
      store_w(1, wr[1]);
      load_w(wr[0], w0 + (size_t)2 * KB * WQ);
      load_w(wr[1], w0 + (size_t)3 * KB * WQ);
      load_w(wr[2], w0 + (size_t)4 * KB * WQ);
    }
    OS_STAMP(2);
    for (int it = 0; it < items; ++it) {           // nine steps per item, unrolled
      const bool more = it + 1 < items;
      int ncb = cb, nq = q, nbase = base, ncnt = cnt;
      if (more) item_of(it + 1, ncb, nq, nbase, ncnt);
      // filter tiles of this item's offsets and of the next item's (the fetch runs five steps ahead)
      const u32x4 *wcur = wbase + ((size_t)(q * GK) * KB + cb) * WQ;
      const u32x4 *wnxt = more ? wbase + ((size_t)(nq * GK) * KB + ncb) * WQ : wcur;
      if (it > 0) __syncthreads();                 // everybody has left the previous rows
      if (!HALO_DBG(8)) halo_store();
      if (more && !HALO_DBG(8)) halo_fetch(ncb, nbase, ncnt);      // the next rows, fetched under this item's steps
      read_nv(nv[0], q * GK);
      read_nv(nv[1], q * GK + 1);
      __syncthreads();
      read_a(af[0], nv[0], base, cnt);
      read_b(bf[0], 0);
#pragma unroll
      for (int kk = 0; kk < GK; ++kk) {
        if (kk + 1 < GK) {                         // the operands of the next step
          read_a(af[(kk + 1) & 1], nv[(kk + 1) & 1], base, cnt);
          read_b(bf[(kk + 1) & 1], (kk + 1) % 3);
        }
        if (kk + 2 < GK) read_nv(nv[kk & 1], q * GK + kk + 2);
        if (!HALO_DBG(2)) {
          store_w((kk + 2) % 3, wr[kk % 3]);       // W(s + 2) -> ring (its stage was last read in step s - 2)
          load_w(wr[kk % 3], kk + 5 < GK ? wcur + (size_t)(kk + 5) * KB * WQ : wnxt + (size_t)(kk + 5 - GK) * KB * WQ);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfmas(af[kk & 1], bf[kk & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (kk + 1 < GK) __syncthreads();
      }
      cb = ncb, q = nq, base = nbase, cnt = ncnt;
    }
  }

  OS_STAMP(3);
#ifdef DF3D_OS_TRACE
  if (a.trace && (threadIdx.x & 63) == 0) a.trace[((size_t)(blockIdx.x + gridDim.x * blockIdx.y) * 16 + (threadIdx.x >> 6)) * 8 + 5] = steps;
#endif
  // ---- epilogue (as spconv_os_split_kernel, NP = 2): bias, folded BN, residual, ReLU; optional split rows ----
  static_assert(CT == 2 || CT == 4 || CT == 8, "COUT must be 32, 64, 128 or 256");
  if constexpr (CT == 2) {
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
        float2 v = make_float2((acc[rt][0][r] + bi.x) * sc.x + sh.x, (acc[rt][1][r] + bi.y) * sc.y + sh.y);
        if (a.cols) {
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
          const float2 rr = *(const float2 *)(a.residual + o);
          v.x += rr.x;
          v.y += rr.y;
        }
        if (a.relu) {
          v.x = fmaxf(v.x, 0.f);
          v.y = fmaxf(v.y, 0.f);
        }
        if (a.out) *(float2 *)(a.out + o) = v;
        if (a.out_split) {
          unsigned hp, lp;
          split_pair(v.x, v.y, hp, lp);
          char *blk = (char *)a.out_split + (o >> 3) * 32 + (n & 3) * 4;
          *(unsigned *)blk = hp;
          *(unsigned *)(blk + 16) = lp;
        }
      }
    }
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
        const size_t o = (size_t)row * a.ldo + col0 + n * CT;
        unsigned h[CT / 2], l[CT / 2];
#pragma unroll
        for (int q = 0; q < CT / 4; ++q) {
          f32x4 v = (f32x4){acc[rt][q * 4][r], acc[rt][q * 4 + 1][r], acc[rt][q * 4 + 2][r], acc[rt][q * 4 + 3][r]};
          v = (v + bi[q]) * sc[q] + sh[q];
          if (a.residual) v += *(const f32x4 *)(a.residual + o + q * 4);
          if (a.relu) {
            v[0] = fmaxf(v[0], 0.f);
            v[1] = fmaxf(v[1], 0.f);
            v[2] = fmaxf(v[2], 0.f);
            v[3] = fmaxf(v[3], 0.f);
          }
          if (a.out) *(f32x4 *)(a.out + o + q * 4) = v;
          if (a.out_split) {
            split_pair(v[0], v[1], h[q * 2], l[q * 2]);
            split_pair(v[2], v[3], h[q * 2 + 1], l[q * 2 + 1]);
          }
        }
        if (a.out_split) {
          char *blk = (char *)a.out_split + (o >> 3) * 32;
          if constexpr (CT == 8) {
            *(u32x4 *)blk = (u32x4){h[0], h[1], h[2], h[3]};
            *(u32x4 *)(blk + 16) = (u32x4){l[0], l[1], l[2], l[3]};
          } else {
            blk += (n & 1) * 8;
            *(u32x2 *)blk = (u32x2){h[0], h[1]};
            *(u32x2 *)(blk + 16) = (u32x2){l[0], l[1]};
          }
        }
      }
    }
  }
  OS_STAMP(4);
}

// ---------------------------------------------------------------------------------------------------------
// The same staging with EIGHT waves of 16 rows (TM = 128) and <= 80 KB of LDS, so that two workgroups share a CU: four
// waves per SIMD hide each other's LDS round trips, barriers and scalar code, and one workgroup's prologue / epilogue runs
// under the other's steps -- what the four-wave kernel above lacks (its phases add up: measured 112 us against 67 us of
// spconv_os_split_kernel at 64 -> 64, although it moves a third of the bytes).  A step is spconv_os_split_kernel's step
// with the gathers replaced by LDS reads: barrier; filter tile of step s + 1 -> LDS (two stages), W(s + 4) fetched; A
// fragments from the staged rows; B fragments pair by pair under the MFMAs.
template <int CIN, int COUT, int HCAP, int RT>
__global__ __launch_bounds__(512, RT == 1 ? 2 : 1) void spconv_halo8_kernel(SplitConvArgs a) {
  constexpr int CW = COUT > 128 ? 128 : COUT;
  constexpr int KB = CIN / 32, CT = CW / 16, TM = 128 * RT, NT = 512, WROWS = 16 * RT;
  constexpr int WQ = CT * 2 * 64;
  constexpr int WPT = (WQ + NT - 1) / NT;
  constexpr int HPT = HCAP * 8 / NT;
  constexpr int GK = 9;
  static_assert(HCAP % 64 == 0, "tile shape");
  extern __shared__ __align__(16) unsigned char halo_smem[];
  u32x4 *Hl = (u32x4 *)halo_smem;                 // [(HCAP + 1)][8]
  u32x4 *Wl = Hl + (HCAP + 1) * 8;                // [2][WQ]
  int *nbrL = (int *)(Wl + 2 * WQ);               // [27][TM]
  int *rowL = nbrL + 27 * TM;
  int *grp = rowL + TM;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int row0 = tile * TM;
  const int col0 = blockIdx.y * CW;
  const int NG = a.K / GK;
  const u32x4 *featb = a.feat + (size_t)blockIdx.y * a.in_goff;

  if (tid < 6) grp[tid] = (tid & 1) ? -1 : 0x7fffffff;
  if (tid < 8) Hl[HCAP * 8 + tid] = (u32x4){0u, 0u, 0u, 0u};
  __syncthreads();
  {
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-1, -1, -1};
    constexpr int KSTEP = NT / TM, NPASS = (27 + KSTEP - 1) / KSTEP;
    const int r = tid % TM, k0 = tid / TM;
    int row = row0 + r;
    row = row < a.n_out ? (a.order ? a.order[row] : row) : a.n_out;
    if (k0 == 0) rowL[r] = row;
    int v[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int k = k0 + i * KSTEP;
      v[i] = (k < a.K && row < a.n_out) ? a.nbr[(size_t)k * a.n_out + row] : -1;
    }
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int k = k0 + i * KSTEP;
      if (k < a.K) nbrL[k * TM + r] = v[i];
      if (v[i] >= 0) {
        const int gi = k / GK;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          if (gi == q) {
            lo[q] = v[i] < lo[q] ? v[i] : lo[q];
            hi[q] = v[i] > hi[q] ? v[i] : hi[q];
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const int ol = __shfl_xor(lo[q], off, 64), oh = __shfl_xor(hi[q], off, 64);
        lo[q] = ol < lo[q] ? ol : lo[q];
        hi[q] = oh > hi[q] ? oh : hi[q];
      }
      if (lane == 0 && hi[q] >= 0) {
        atomicMin(&grp[2 * q], lo[q]);
        atomicMax(&grp[2 * q + 1], hi[q]);
      }
    }
  }
  __syncthreads();
  int glo[3], gcnt[3], gch[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    glo[q] = __builtin_amdgcn_readfirstlane(grp[2 * q]);
    const int h = __builtin_amdgcn_readfirstlane(grp[2 * q + 1]);
    gcnt[q] = (q < NG && h >= 0) ? h - glo[q] + 1 : 0;
    gch[q] = (gcnt[q] + HCAP - 1) / HCAP;
  }
  const int per_cb = gch[0] + gch[1] + gch[2];
  const int items = per_cb * KB;
  auto item_of = [&](int it, int &cb, int &q, int &base, int &cnt) {
    cb = it / per_cb;
    int r = it - cb * per_cb;
    q = r < gch[0] ? 0 : (r < gch[0] + gch[1] ? 1 : 2);
    r -= q == 0 ? 0 : (q == 1 ? gch[0] : gch[0] + gch[1]);
    const int lo = q == 0 ? glo[0] : (q == 1 ? glo[1] : glo[2]);
    const int n_ = q == 0 ? gcnt[0] : (q == 1 ? gcnt[1] : gcnt[2]);
    base = lo + r * HCAP;
    cnt = n_ - r * HCAP < HCAP ? n_ - r * HCAP : HCAP;
  };

  f32x4 acc[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 wr[3][WPT];
  const u32x4 *wbase = a.w + (size_t)blockIdx.y * a.K * KB * WQ;
  auto load_w = [&](u32x4 (&wreg)[WPT], const u32x4 *src) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int e = tid + NT * i;
      wreg[i] = src[(WQ % NT == 0 || e < WQ) ? e : 0];
    }
  };
  auto store_w = [&](int buf, u32x4 (&wreg)[WPT]) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int e = tid + NT * i;
      if (WQ % NT == 0 || e < WQ) Wl[buf * WQ + e] = wreg[i];
    }
  };
  u32x4 hreg[HPT];
  auto halo_fetch = [&](int cb, int base, int cnt) {
    const u32x4 *src = featb + cb * 8 + (tid & 7);
#pragma unroll
    for (int i = 0; i < HPT; ++i) {
      int slot = (tid >> 3) + 64 * i;
      slot = slot < cnt ? slot : cnt - 1;
      hreg[i] = src[(size_t)(base + slot) * a.ldi];
    }
  };
  auto halo_store = [&]() {
#pragma unroll
    for (int i = 0; i < HPT; ++i) {
      const int slot = (tid >> 3) + 64 * i, unit = tid & 7;
      Hl[slot * 8 + (unit ^ ((slot >> 1) & 3))] = hreg[i];
    }
  };

  if (items > 0) {
    int cb, q, base, cnt;
    item_of(0, cb, q, base, cnt);
    {
      const u32x4 *w0 = wbase + ((size_t)(q * GK) * KB + cb) * WQ;
      load_w(wr[0], w0);
      halo_fetch(cb, base, cnt);
      store_w(0, wr[0]);                           // W(0) -> stage 0; W(1), W(2), W(3) in the register sets
      load_w(wr[0], w0 + (size_t)KB * WQ);
      load_w(wr[1], w0 + (size_t)2 * KB * WQ);
      load_w(wr[2], w0 + (size_t)3 * KB * WQ);
    }
    for (int it = 0; it < items; ++it) {
      const bool more = it + 1 < items;
      int ncb = cb, nq = q, nbase = base, ncnt = cnt;
      if (more) item_of(it + 1, ncb, nq, nbase, ncnt);
      const u32x4 *wcur = wbase + ((size_t)(q * GK) * KB + cb) * WQ;
      const u32x4 *wnxt = more ? wbase + ((size_t)(nq * GK) * KB + ncb) * WQ : wcur;
      const int par = it & 1;                      // parity of the item's first step (nine steps per item)
      if (it > 0) __syncthreads();                 // everybody has left the previous rows
      halo_store();
      if (more) halo_fetch(ncb, nbase, ncnt);
      int nvn[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) nvn[rt] = nbrL[(q * GK) * TM + wave * WROWS + rt * 16 + n];
#pragma unroll
      for (int kk = 0; kk < GK; ++kk) {
        __syncthreads();
        u32x4 ah[RT], al[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int idx = nvn[rt];
          if (kk + 1 < GK) nvn[rt] = nbrL[(q * GK + kk + 1) * TM + wave * WROWS + rt * 16 + n];
          const unsigned sl = (unsigned)(idx - base);
          const int slot = (idx >= 0 && sl < (unsigned)cnt) ? (int)sl : HCAP;
          const int sw = (slot >> 1) & 3;
          ah[rt] = Hl[slot * 8 + ((2 * g) ^ sw)];
          al[rt] = Hl[slot * 8 + ((2 * g + 1) ^ sw)];
        }
        const u32x4 *wb = Wl + ((kk + par) & 1) * WQ + lane;
        u32x4 bq[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bq[0][j] = wb[j * 64];
#pragma unroll
        for (int i = 0; i < CT / 2; ++i) {
          const int c2 = i * 2;
          if (i == (CT / 2 > 1 ? 1 : 0)) {         // the next step's filter tile, in the shadow of the first MFMA batch
            store_w((kk + par + 1) & 1, wr[kk % 3]);
            load_w(wr[kk % 3], kk + 4 < GK ? wcur + (size_t)(kk + 4) * KB * WQ : wnxt + (size_t)(kk + 4 - GK) * KB * WQ);
          }
          if (i + 1 < CT / 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bq[(i + 1) & 1][j] = wb[((i + 1) * 4 + j) * 64];
          }
          const u32x4 bh0 = bq[i & 1][0], bl0 = bq[i & 1][1], bh1 = bq[i & 1][2], bl1 = bq[i & 1][3];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            acc[rt][c2] = DF3D_MFMA_BF16(al[rt], bh0, acc[rt][c2]);
            acc[rt][c2 + 1] = DF3D_MFMA_BF16(al[rt], bh1, acc[rt][c2 + 1]);
          }
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            acc[rt][c2] = DF3D_MFMA_BF16(ah[rt], bl0, acc[rt][c2]);
            acc[rt][c2 + 1] = DF3D_MFMA_BF16(ah[rt], bl1, acc[rt][c2 + 1]);
          }
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            acc[rt][c2] = DF3D_MFMA_BF16(ah[rt], bh0, acc[rt][c2]);
            acc[rt][c2 + 1] = DF3D_MFMA_BF16(ah[rt], bh1, acc[rt][c2 + 1]);
          }
        }
      }
      cb = ncb, q = nq, base = nbase, cnt = ncnt;
    }
  }

  // ---- epilogue ----
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
  if constexpr (CT == 2) {
    const int col = col0 + n * 2;
    const float2 bi = a.bias ? *(const float2 *)(a.bias + col) : make_float2(0.f, 0.f);
    const float2 sc = a.scale ? *(const float2 *)(a.scale + col) : make_float2(1.f, 1.f);
    const float2 sh = a.shift ? *(const float2 *)(a.shift + col) : make_float2(0.f, 0.f);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rowL[wave * WROWS + rt * 16 + 4 * g + r];
      if (row >= a.n_out) continue;
      float2 v = make_float2((acc[rt][0][r] + bi.x) * sc.x + sh.x, (acc[rt][1][r] + bi.y) * sc.y + sh.y);
      const size_t o = (size_t)row * a.ldo + col;
      if (a.residual) {
        const float2 rr = *(const float2 *)(a.residual + o);
        v.x += rr.x;
        v.y += rr.y;
      }
      if (a.relu) {
        v.x = fmaxf(v.x, 0.f);
        v.y = fmaxf(v.y, 0.f);
      }
      if (a.out) *(float2 *)(a.out + o) = v;
      if (a.out_split) {
        unsigned hp, lp;
        split_pair(v.x, v.y, hp, lp);
        char *blk = (char *)a.out_split + (o >> 3) * 32 + (n & 3) * 4;
        *(unsigned *)blk = hp;
        *(unsigned *)(blk + 16) = lp;
      }
    }
  } else {
    f32x4 bi[CT / 4], sc[CT / 4], sh[CT / 4];
#pragma unroll
    for (int qq = 0; qq < CT / 4; ++qq) {
      const int col = col0 + n * CT + qq * 4;
      bi[qq] = a.bias ? *(const f32x4 *)(a.bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
      sc[qq] = a.scale ? *(const f32x4 *)(a.scale + col) : (f32x4){1.f, 1.f, 1.f, 1.f};
      sh[qq] = a.shift ? *(const f32x4 *)(a.shift + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rowL[wave * WROWS + rt * 16 + 4 * g + r];
      if (row >= a.n_out) continue;
      const size_t o = (size_t)row * a.ldo + col0 + n * CT;
      unsigned h[CT / 2], l[CT / 2];
#pragma unroll
      for (int qq = 0; qq < CT / 4; ++qq) {
        f32x4 v = (f32x4){acc[rt][qq * 4][r], acc[rt][qq * 4 + 1][r], acc[rt][qq * 4 + 2][r], acc[rt][qq * 4 + 3][r]};
        v = (v + bi[qq]) * sc[qq] + sh[qq];
        if (a.residual) v += *(const f32x4 *)(a.residual + o + qq * 4);
        if (a.relu) {
          v[0] = fmaxf(v[0], 0.f);
          v[1] = fmaxf(v[1], 0.f);
          v[2] = fmaxf(v[2], 0.f);
          v[3] = fmaxf(v[3], 0.f);
        }
        if (a.out) *(f32x4 *)(a.out + o + qq * 4) = v;
        if (a.out_split) {
          split_pair(v[0], v[1], h[qq * 2], l[qq * 2]);
          split_pair(v[2], v[3], h[qq * 2 + 1], l[qq * 2 + 1]);
        }
      }
      if (a.out_split) {
        char *blk = (char *)a.out_split + (o >> 3) * 32;
        if constexpr (CT == 8) {
          *(u32x4 *)blk = (u32x4){h[0], h[1], h[2], h[3]};
          *(u32x4 *)(blk + 16) = (u32x4){l[0], l[1], l[2], l[3]};
        } else {
          blk += (n & 1) * 8;
          *(u32x2 *)blk = (u32x2){h[0], h[1]};
          *(u32x2 *)(blk + 16) = (u32x2){l[0], l[1]};
        }
      }
    }
  }
  }
}

template <int CIN, int COUT, int HCAP, int RT>
static int launch_halo8_cfg(const SplitConvArgs &a, hipStream_t stream) {
  constexpr int CW = COUT > 128 ? 128 : COUT, CT = CW / 16, WQ = CT * 2 * 64;
  constexpr size_t lds = (size_t)(HCAP + 1) * 128 + 2 * (size_t)WQ * 16 + (size_t)27 * 128 * RT * 4 + 128 * RT * 4 + 64;
  static_assert(lds <= (RT == 1 ? 80 : 160) * 1024, "LDS budget");
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute((const void *)spconv_halo8_kernel<CIN, COUT, HCAP, RT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS) failed: %s", hipGetErrorString(e));
      return DF3D_EHIP;
    }
    configured = true;
  }
  hipLaunchKernelGGL((spconv_halo8_kernel<CIN, COUT, HCAP, RT>), dim3(cdiv(a.n_out, 128 * RT), a.gy), dim3(512), lds, stream, a);
  return DF3D_OK;
}

template <int CIN, int COUT, int RT, int HCAP>
static int launch_halo_cfg(const SplitConvArgs &a, hipStream_t stream) {
  constexpr int CW = COUT > 128 ? 128 : COUT, CT = CW / 16, TM = 64 * RT, WQ = CT * 2 * 64;
  constexpr size_t lds = (size_t)(HCAP + 1) * 128 + 3 * (size_t)WQ * 16 + (size_t)27 * TM * 4 + TM * 4 + 64;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute((const void *)spconv_halo_kernel<CIN, COUT, RT, HCAP>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS) failed: %s", hipGetErrorString(e));
      return DF3D_EHIP;
    }
    configured = true;
  }
  hipLaunchKernelGGL((spconv_halo_kernel<CIN, COUT, RT, HCAP>), dim3(cdiv(a.n_out, TM), a.gy), dim3(256), lds, stream, a);
  return DF3D_OK;
}

// MEASURED (MI355X, real nuScenes stage geometry, tools/halo_probe.py; us per launch, round-2 kernel first):
//                         os / lc    4 waves x 32 rows   4 waves x 64 rows   8 waves x 16 rows (2 WG/CU)   8 waves x 32 rows
//   conv2  32 ->  32        37              62                  98                     60                        79
//   conv3  64 ->  64        68             111                 109                     71                        89
//   conv4 128 -> 128        64             128                 290                    161                       158
// All variants reproduce the round-2 results to 1e-6 and move a THIRD of the input bytes -- and none is faster.  The
// ablation (tools/halo_ablate.sh, -DDF3D_HALO_EXPERIMENTS) says why: with one wave per SIMD the phases of a step add up
// (skeleton 55 us + staging 10 + A reads 12 + filter staging 12 + MFMAs 33 at 64 -> 64; a per-step cursor alone cost 25 us
// until the items were addressed arithmetically: every scalar instruction is exposed), with four waves per SIMD the step
// costs the same ~1400 clocks as with register gathers.  So the gathers are NOT what a step of the round-2 kernels waits
// for: what all structures share is the barrier-synchronised step in which every wave re-reads the whole filter tile
// from LDS (64 KB per step and workgroup at 64 columns) for 12 MFMAs of its own.  The kernels stay in the library as an
// opt-in (DF3D_CONV_HALO=1, DF3D_HALO_RT=2|4|8|9) with their parity tests; the structure that should profit from the
// staged ranges is the loader / consumer kernel, whose pole IS its row DMA.
static int halo_mode(const SplitConvArgs &a) {
  const char *t = getenv("DF3D_CONV_HALO");         // read per call: the tests switch it
  if ((a.K != 27 && a.K != 9) || a.cols) return 0;
  return (t && t[0] == '1') ? 1 : 0;
}

template <int CIN, int COUT>
static int launch_halo(const SplitConvArgs &a, hipStream_t stream) {
  constexpr int CW = COUT > 128 ? 128 : COUT;
  int rt = CW >= 128 ? 2 : 4;
  // small maps: 128-row tiles fill the chip better
  if (rt == 4 && (long long)cdiv(a.n_out, 256) * a.gy < 160) rt = 2;
  const char *f = getenv("DF3D_HALO_RT");
  if (f && (f[0] == '2' || f[0] == '4')) rt = f[0] - '0';
  if (!f || f[0] == '8') {                         // eight waves of 16 rows, two workgroups per CU
    if constexpr (CW >= 128) return launch_halo8_cfg<CIN, COUT, 256, 1>(a, stream);
    else return launch_halo8_cfg<CIN, COUT, 384, 1>(a, stream);
  }
  if (f[0] == '9') return launch_halo8_cfg<CIN, COUT, 640, 2>(a, stream);      // eight waves of 32 rows, one workgroup per CU
  if (rt == 4) return launch_halo_cfg<CIN, COUT, 4, 640>(a, stream);
  return launch_halo_cfg<CIN, COUT, 2, 512>(a, stream);
}
