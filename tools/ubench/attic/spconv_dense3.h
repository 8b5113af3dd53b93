// 3 x 3 convolution over a DENSE H x W map with 64 input channels per pixel and many 128-column blocks of output channels
// (the 36 middle convolutions of the CenterPoint head: 64 -> 36 x 64 on the 180 x 180 map = 18 blocks), input rows resident
// in LDS.  Included by spconv_split.hip behind the loader / consumer kernel, whose matrix-wave step structure it reuses.
//
// Reference: CP/det3d/models/bbox_heads/center_head.py:66-110 (SepHead: Conv2d 64 -> 64, 3 x 3, padding 1, + BN + ReLU per
// branch).  Why another kernel (round 3, tools/ubench/lc_trace_head.py): in spconv_os_lc_kernel a step moves 16 KB of input
// rows + 16 KB of filters into the CU for 768 clocks of MFMAs, a CU takes ~48 B/clk from L2, and the eight loader waves spend
// as long issuing a step's DMAs (~790 clocks) as the matrix waves multiplying -- the step runs at ~1100-1300 clocks.  On a dense
// map the nine taps of a row tile read the SAME rows over and over: tap (ky, kx) of pixel p is row p + (ky - 1) W + (kx - 1),
// so the three taps of one ky are one run of 130 consecutive rows read at three shifts, and every one of the 18 column blocks
// reads them again.  Here the three runs (3 x 132 rows x 256 B = 99 KB) are loaded ONCE per row tile and stay in LDS; a step
// only streams its 16 KB of filters (ring of three stages).  A lane reads its A fragment at row r + kx of run ky -- or at an
// all-zero row when the tap leaves the map (the padding), decided by arithmetic on the pixel index: there is no neighbour
// table.  Products and their order are those of the generic kernels (offsets ascending, channel blocks inside, lo*hi, hi*lo,
// hi*hi): results are bit-identical to df3d_conv_rows_split on the table of df3d_conv2d_neighbors (tests/test_gpu_head.py).
//
// Synchronisation: one s_barrier per step, placed BEFORE the step's last MFMA batch (its B fragments are in registers by
// then): at barrier s the matrix waves have left filter stage s (the loaders refill it with step s + 3) and stage s + 1 is
// complete, so the first fragments of step s + 1 are read behind the MFMAs of that last batch.  A filter tile therefore has
// two steps to arrive with a ring of three stages (four would not fit beside the resident rows).
//
// NOT BUILT (round 3 negative result): bit-identical to the table path on every tested map and it halves
// what the loader waves issue per step, but the head's middle convolutions take 274 us with it against 267 us without.  In the
// loader / consumer kernel the matrix waves (~916 clocks per step, 768 of them MFMAs) and the loader waves (~760 clocks issuing
// four LDS-DMAs + the landing wait) are balanced and meet at a barrier; taking work off one side changes nothing (ablations:
// no MFMAs 70 %, no DMAs 83 %, neither 54 %, no B-fragment LDS reads 100 % of the launch).  It was wired as
// df3d_conv_rows_split_map3x3(in_split, in_channels, cin, in_group_stride, packed, cout, groups, nbr, batch, H, W, ...) with
// SplitConvArgs.dH / dW and `#include "spconv_dense3.h"` behind spconv_ws.h.
#pragma once

template <int CW>
__global__ __launch_bounds__(768) void conv3x3_c64_resident_kernel(SplitConvArgs a) {
  constexpr int NP = 2, TM = 128, KB = 2, NS = 3, KV = 9, SPB = KV * KB;      // 18 steps per column block
  constexpr int RT = 2, CT = CW / 16;
  static_assert(CW == 128, "column block");
  constexpr int WQ = CT * NP * 64;                // u32x4 per (offset, 32-channel block) filter tile = 16 KB
  constexpr int AR = 132, AQ = AR * 16;           // rows / u32x4 of one resident run (128 + 2 shifts, whole 4-row DMA pieces)
  constexpr int APC = AR / 4;                     // DMA pieces (4 rows x 256 B) per run
  constexpr int NLOAD = 8, WPL = (WQ / 64) / NLOAD;
  __shared__ __attribute__((aligned(256))) u32x4 Ar[3][AQ];
  __shared__ __attribute__((aligned(256))) u32x4 Wl[NS][WQ];
  __shared__ __attribute__((aligned(256))) u32x4 Zr[16];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int row0 = tile * TM;
  const int cbw = a.cbw > 0 ? a.cbw : 1;
  const int cb0 = blockIdx.y * cbw;
  const int ncb = min(cbw, a.gy - cb0);
  const int total = SPB * ncb;
  const int W = a.dW, H = a.dH;
  if (tid < 16) Zr[tid] = (u32x4){0u, 0u, 0u, 0u};
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // in LDS before the first (inline-asm) barrier

  if (wave >= 4) {
    // ------------------------------------------------ loader waves ------------------------------------------------
    __builtin_amdgcn_s_setprio(3);
    const int lw = wave - 4;
    // resident runs: run ky holds rows row0 + (ky - 1) W - 1 .. + 131 (clamped into the map: rows outside are never read
    // through a valid tap).  Row i's sixteen 16-byte units are stored at unit ^ (i & 15): sixteen consecutive rows read at
    // one logical unit then cover the 64 banks once (the XOR is applied to the SOURCE address, an LDS-DMA writes
    // lane-linearly).
    for (int pc = lw; pc < 3 * APC; pc += NLOAD) {
      const int ky = pc / APC, p4 = pc - ky * APC;
      const int i = p4 * 4 + (lane >> 4);
      long long q = (long long)row0 + (long long)(ky - 1) * W - 1 + i;
      q = q < 0 ? 0 : (q > a.n_in - 1 ? a.n_in - 1 : q);
      const u32x4 *src = a.feat + (size_t)q * a.ldi + ((lane & 15) ^ (i & 15));
      __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)&Ar[ky][p4 * 64], 16, 0, 0);
    }
    int st3 = 0;                                  // stage of the step that is fetched next
    auto issue = [&](int t) {
      const int cb = cb0 + t / SPB, kk = t - (t / SPB) * SPB;
      const u32x4 *wsrc = a.w + ((size_t)cb * SPB + kk) * WQ + (lw * WPL) * 64 + lane;
#pragma unroll
      for (int i = 0; i < WPL; ++i)
        __builtin_amdgcn_global_load_lds(wsrc + i * 64, (__attribute__((address_space(3))) void *)&Wl[st3][(lw * WPL + i) * 64],
                                         16, 0, 0);
      st3 = st3 == NS - 1 ? 0 : st3 + 1;
    };
    for (int t = 0; t < NS; ++t) issue(t);        // total >= 18
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WPL) : "memory");       // the runs and the filters of step 0
    asm volatile("s_barrier" ::: "memory");
    for (int s = 0; s < total; ++s) {
      // filters of step s + 1 have landed; those of step s + 2 (the newest issued) may be in flight
      if (s + 2 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPL) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      if (s + NS < total) issue(s + NS);          // into the stage step s has just left
    }
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
  // taps inside the map, per row tile of this lane (bit k = ky * 3 + kx)
  unsigned vm[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int p = row0 + wr * 32 + rt * 16 + n;
    vm[rt] = 0u;
    if (p < a.n_out) {
      const int rem = p % (H * W), y = rem / W, x = rem - y * W;
#pragma unroll
      for (int k = 0; k < KV; ++k) {
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) vm[rt] |= 1u << k;
      }
    }
  }
  const unsigned ar_lds = lds_addr(&Ar[0][0]), z_lds = lds_addr(&Zr[0]) + (unsigned)(g * 2) * 16u;
  const unsigned w_addr = lds_addr(&Wl[0][0]) + (unsigned)lane * 16u;
  // byte addresses of the lane's A fragments of tap k, channel block 0: [rt][hi | lo]; channel block 1 = the same ^ 128
  auto tap_addr = [&](int k, unsigned (&ad)[RT][NP]) {
    const int ky = k / 3, kx = k - ky * 3;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int i = wr * 32 + rt * 16 + n + kx;
      const unsigned base = ar_lds + (unsigned)ky * (AQ * 16u) + (unsigned)i * 256u;
      const bool ok = (vm[rt] >> k) & 1u;
#pragma unroll
      for (int q = 0; q < NP; ++q) ad[rt][q] = ok ? base + (unsigned)(((g * 2 + q) ^ (i & 15)) * 16) : z_lds + q * 16u;
    }
  };
#define LC_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define LC_SB() __builtin_amdgcn_sched_barrier(0)
#define LC_M(rt, c, ap, br, buf, slot) acc[rt][c] = DF3D_MFMA_BF16(af[buf][rt][ap], bq[slot][br], acc[rt][c])
#define LC_BATCH(i, buf, slot, RA, RB, RC, RD, RE, RF, RG, RH)                    \
  do {                                                                            \
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
#define LC_BATCH_X(...) LC_BATCH(__VA_ARGS__)
#define LC_WAITALL(buf)                                                                                                \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                                  \
               : "+v"(af[buf][0][0]), "+v"(af[buf][0][1]), "+v"(af[buf][1][0]), "+v"(af[buf][1][1]), "+v"(bq[0][0]),   \
                 "+v"(bq[0][1]), "+v"(bq[0][2]), "+v"(bq[0][3]), "+v"(bq[1][0]), "+v"(bq[1][1]), "+v"(bq[1][2]), "+v"(bq[1][3]))
#define LC_NONE do { } while (0)
#define LC_READ_B4(slot, wa, i)                                                                                        \
  LC_READ(bq[slot][0], wa, ((i) * 4 + 0) * 1024), LC_READ(bq[slot][1], wa, ((i) * 4 + 1) * 1024),                      \
      LC_READ(bq[slot][2], wa, ((i) * 4 + 2) * 1024), LC_READ(bq[slot][3], wa, ((i) * 4 + 3) * 1024)
#define DS_READ_A4(buf, ad, x)                                                                                         \
  LC_READ(af[buf][0][0], ad##00, 0), LC_READ(af[buf][0][1], ad##01, 0), LC_READ(af[buf][1][0], ad##10, 0),             \
      LC_READ(af[buf][1][1], ad##11, 0)
  // one step on filter stage `cur` (byte offset); its A fragments (buffer `buf`) and B batch 0 are in flight on entry, those
  // of the next step (buffer buf ^ 1, addresses nx00 .. nx11, filter stage `nxt`) on exit
#define DS_STEP(cur, nxt, buf, nx)                                                                                     \
  do {                                                                                                                 \
    const unsigned wt = w_addr + (cur);                                                                                \
    const unsigned wn = w_addr + (nxt);                                                                                \
    LC_WAITALL(buf);                                                                                                   \
    LC_BATCH_X(0, buf, 0, LC_READ_B4(1, wt, 1), LC_NONE, LC_NONE, LC_NONE, LC_NONE);                                   \
    LC_WAITALL(buf);                                                                                                   \
    LC_BATCH_X(1, buf, 1, LC_READ_B4(0, wt, 2), LC_NONE, LC_NONE, LC_NONE, LC_NONE);                                   \
    LC_WAITALL(buf);                                                                                                   \
    LC_BATCH_X(2, buf, 0, LC_READ_B4(1, wt, 3), DS_READ_A4((buf) ^ 1, nx, 0));                                         \
    LC_WAITALL(buf);                                                                                                   \
    asm volatile("s_barrier" ::: "memory");                                                                            \
    LC_BATCH_X(3, buf, 1, LC_READ_B4(0, wn, 0), LC_NONE, LC_NONE, LC_NONE, LC_NONE);                                   \
  } while (0)

  // ---- epilogue of one column block (as in spconv_os_lc_kernel; rows are the tile's pixels) ----
  const float *e_res = a.residual;
  float *e_out = a.out;
  char *e_split = (char *)a.out_split;
  const float *e_bias = a.bias, *e_scale = a.scale, *e_shift = a.shift;
  const int e_relu = a.relu, e_ldo = a.ldo, e_nout = a.n_out;
  static_assert(CT == 8, "two f32x4 per lane and vector below");
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
        const int row = row0 + wr * 32 + rt * 16 + 4 * g + r;
        if (row >= e_nout) continue;
        const size_t o = (size_t)row * e_ldo + col0 + n * CT;
        unsigned hh[CT / 2], ll[CT / 2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          f32x4 v = (f32x4){acc[rt][q * 4][r], acc[rt][q * 4 + 1][r], acc[rt][q * 4 + 2][r], acc[rt][q * 4 + 3][r]};
          v = (v + bi[q]) * sc[q] + sh[q];
          if (e_res) v += *(const f32x4 *)(e_res + o + q * 4);
          if (e_relu) {
            v[0] = fmaxf(v[0], 0.f);
            v[1] = fmaxf(v[1], 0.f);
            v[2] = fmaxf(v[2], 0.f);
            v[3] = fmaxf(v[3], 0.f);
          }
          if (e_out) *(f32x4 *)(e_out + o + q * 4) = v;
          if (e_split) {
            split_pair(v[0], v[1], hh[q * 2], ll[q * 2]);
            split_pair(v[2], v[3], hh[q * 2 + 1], ll[q * 2 + 1]);
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
  unsigned ad[RT][NP], nd[RT][NP];                 // fragment addresses of the current tap / the next one (channel block 0)
  tap_addr(0, ad);
  asm volatile("s_barrier" ::: "memory");          // the resident rows and the filters of step 0 are in LDS
  unsigned ws = 0;                                 // byte offset of the current step's filter stage
  auto next_stage = [](unsigned s) { return s == (NS - 1) * (WQ * 16u) ? 0u : s + WQ * 16u; };
  {
    const unsigned a00 = ad[0][0], a01 = ad[0][1], a10 = ad[1][0], a11 = ad[1][1];
    LC_READ(af[0][0][0], a00, 0);
    LC_READ(af[0][0][1], a01, 0);
    LC_READ(af[0][1][0], a10, 0);
    LC_READ(af[0][1][1], a11, 0);
    LC_READ(bq[0][0], w_addr, 0 * 1024);
    LC_READ(bq[0][1], w_addr, 1 * 1024);
    LC_READ(bq[0][2], w_addr, 2 * 1024);
    LC_READ(bq[0][3], w_addr, 3 * 1024);
  }
  for (int cb = cb0; cb < cb0 + ncb; ++cb) {
    for (int k = 0; k < KV; ++k) {
      tap_addr(k + 1 < KV ? k + 1 : 0, nd);
      {
        // channel block 0 of tap k; the next step reads block 1 of the same rows: unit ^ 8
        const unsigned nx00 = ad[0][0] ^ 128u, nx01 = ad[0][1] ^ 128u, nx10 = ad[1][0] ^ 128u, nx11 = ad[1][1] ^ 128u;
        const unsigned w1 = next_stage(ws);
        DS_STEP(ws, w1, 0, nx);
        ws = w1;
      }
      {
        const unsigned nx00 = nd[0][0], nx01 = nd[0][1], nx10 = nd[1][0], nx11 = nd[1][1];
        const unsigned w1 = next_stage(ws);
        DS_STEP(ws, w1, 1, nx);
        ws = w1;
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int q = 0; q < NP; ++q) ad[rt][q] = nd[rt][q];
    }
    // the first fragments of the next block's step 0 are in flight: they land, the epilogue runs without them, and they are
    // read again (the rows are resident, the filter stage is complete and not refilled before its own barrier)
    LC_WAITALL(0);
    __builtin_amdgcn_s_setprio(0);
    epilogue(cb);
    __builtin_amdgcn_s_setprio(1);
    if (cb + 1 < cb0 + ncb) {
      const unsigned a00 = ad[0][0], a01 = ad[0][1], a10 = ad[1][0], a11 = ad[1][1];
      const unsigned rwn = w_addr + ws;
      LC_READ(af[0][0][0], a00, 0);
      LC_READ(af[0][0][1], a01, 0);
      LC_READ(af[0][1][0], a10, 0);
      LC_READ(af[0][1][1], a11, 0);
      LC_READ(bq[0][0], rwn, 0 * 1024);
      LC_READ(bq[0][1], rwn, 1 * 1024);
      LC_READ(bq[0][2], rwn, 2 * 1024);
      LC_READ(bq[0][3], rwn, 3 * 1024);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#undef LC_READ
#undef LC_SB
#undef LC_M
#undef LC_BATCH
#undef LC_BATCH_X
#undef LC_WAITALL
#undef LC_NONE
#undef LC_READ_B4
#undef DS_READ_A4
#undef DS_STEP
}

// column blocks per workgroup as in launch_os_lc
static int launch_conv3x3_c64_resident(const SplitConvArgs &a_, hipStream_t stream) {
  SplitConvArgs a = a_;
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
  hipLaunchKernelGGL((conv3x3_c64_resident_kernel<128>), dim3(tiles, cdiv(a.gy, a.cbw)), dim3(768), 0, stream, a);
  return DF3D_OK;
}
