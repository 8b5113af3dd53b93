// Weight-stationary split-precision sparse convolution (round 3).  Included by spconv_split.hip (shares SplitConvArgs, the
// packed-weight layout 1, the split rows, the zero row and the MFMA macro).
//
// What rounds 1-3 measured about the output-stationary kernels: a (offset, 32-channel) step costs ~1400 clocks whatever its
// matrix work, because all waves of a workgroup walk the steps in lockstep -- one barrier per step, the step's filter tile
// streamed through LDS behind it -- so the phases of a step (index reads, gathers, fragment reads, MFMAs) ADD instead of
// overlapping, and every (tile, offset) pair costs a step even when none of the tile's rows has a neighbour there.
//
// Here the FILTERS stand still and the waves run free:
//   * the packed filter tiles of GK whole offsets (all their 32-channel blocks: 16 KB per offset at 64 -> 64, 4 KB at
//     32 -> 32) are copied to LDS once per workgroup and offset group -- 3 groups of 9 at 64 -> 64, ONE group at 32 -> 32;
//     the only barriers of the kernel are the two around each copy;
//   * a wave owns up to TMAX 16-row tiles (accumulators in registers across the groups) and walks, per tile, the offsets of
//     the group at which at least one of its 16 rows has a neighbour (a ballot over the tile's table entries) -- empty
//     (tile, offset) pairs cost nothing;
//   * per (tile, offset) item: the rows' split fragments straight from L2 / HBM (absent neighbours read a zero row:
//     branch-free loads, counted waits), issued one item ahead of the matrix instructions that consume them; B fragments by
//     lane-linear 16-byte LDS reads; no synchronisation with the other waves, so one wave's gather latency is another
//     wave's MFMA time.
// Same operands, same accumulation order per output element as the output-stationary kernel (offsets ascending, channel
// blocks ascending, lo*hi + hi*lo + hi*hi) -- the results are bit-identical, which is how tests/test_gpu_ops.py pins it.
// Epilogue = the output-stationary kernel's (bias, folded BN, residual, ReLU, fp32 rows and / or split rows).

#ifndef DF3D_WS_NW
#define DF3D_WS_NW 8
#endif
template <int CIN, int COUT, int GK, int NW, int TMAX>
__global__ __launch_bounds__(NW * 64) void spconv_ws_kernel(SplitConvArgs a, int tiles_per_wg) {
  constexpr int KB = CIN / 32;                 // 32-channel blocks
  constexpr int CT = COUT / 16;                // 16-column tiles
  constexpr int WQ = CT * 2 * 64;              // u32x4 per (offset, channel block) filter tile
  constexpr int NT = NW * 64;
  constexpr int NC = (GK * 16 + 63) / 64;      // 64-entry chunks of a tile's table strip
  extern __shared__ __align__(16) unsigned char ws_smem[];
  u32x4 *Wl = (u32x4 *)ws_smem;                                   // [GK][KB][CT][hi | lo][lane]
  int *nbrAll = (int *)(ws_smem + (size_t)GK * KB * WQ * 16);     // [NW][TMAX][GK][16]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, n = lane & 15;
  int *nbrW = nbrAll + wave * TMAX * GK * 16;
  // a wave owns TMAX CONSECUTIVE 16-row tiles (their rows share neighbours: one gathered line serves several of them)
  const int tile0 = blockIdx.x * tiles_per_wg + wave * (tiles_per_wg / NW);
  const int ntiles_all = (a.n_out + 15) >> 4;
  int my_tiles = 0;
#pragma unroll
  for (int ti = 0; ti < TMAX; ++ti)
    if (ti < tiles_per_wg / NW && tile0 + ti < ntiles_all) my_tiles = ti + 1;

  f32x4 acc[TMAX][CT];
#pragma unroll
  for (int ti = 0; ti < TMAX; ++ti)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ti][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < a.K; k0 += GK) {
    const int nk = a.K - k0 < GK ? a.K - k0 : GK;
    if (k0) __syncthreads();                                      // every wave is done with the previous group's filters
    {
      const u32x4 *src = a.w + (size_t)k0 * KB * WQ;
      const int tot = nk * KB * WQ;
      for (int e0 = 0; e0 < tot; e0 += NT * 4) {                   // four 16-byte loads in flight per thread
        u32x4 t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int e = e0 + tid + NT * i;
          t[i] = src[e < tot ? e : 0];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int e = e0 + tid + NT * i;
          if (e < tot) Wl[e] = t[i];
        }
      }
    }
    // the table entries of the wave's tiles for this group -> its LDS strip; which offsets are alive for which tile
    unsigned tmask[TMAX];
    unsigned any = 0u;
#pragma unroll
    for (int ti = 0; ti < TMAX; ++ti) {
      tmask[ti] = 0u;
      if (ti < my_tiles) {
        const int row0 = (tile0 + ti) * 16;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int e = c * 64 + lane;
          const int kk = e >> 4, row = row0 + (e & 15);
          int v = -1;
          if (kk < nk && row < a.n_out) v = a.nbr[(size_t)(k0 + kk) * a.n_out + row];
          if (e < GK * 16) nbrW[ti * GK * 16 + e] = v;
          const unsigned long long b = __ballot(v >= 0);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if ((b >> (16 * j)) & 0xffffull) tmask[ti] |= 1u << (c * 4 + j);
        }
      }
      any |= tmask[ti];
    }
    __syncthreads();                                              // filters of the group are in LDS (and the strips written)
    if (!any) continue;
    // software pipeline over the alive offsets: the gathers of the next offset (all tiles) fly under this offset's matrix
    // instructions; a B fragment read from LDS serves every alive tile of the wave
    // The loads are issued in a QUAD-COALESCED shape -- lane l reads sub-block l & 3 of row l >> 2, so the four lanes of
    // a quad read 128 contiguous bytes -- and brought into the MFMA operand shape (lane (n, g): row n, sub-block g) by
    // ds_bpermute one offset ahead of their use.  tools/ubench/gather_probe.py: the same bytes loaded directly in the
    // operand shape (a quad = four different rows, 16 bytes each) take 51-64 us at conv3 by themselves -- the whole time
    // of the output-stationary kernel -- against 25 us in this shape.
    u32x4 cur[TMAX][KB][2], nxt[TMAX][KB][2];
    const int lr = lane >> 2, lc = lane & 3;
    const int perm_addr = (4 * n + g) * 4;                        // byte address of the source lane for ds_bpermute
    auto gather = [&](int kk, u32x4 (&dst)[TMAX][KB][2]) {
#pragma unroll
      for (int ti = 0; ti < TMAX; ++ti) {
        if (!((tmask[ti] >> kk) & 1u)) continue;                  // wave-uniform: this tile has no row with a neighbour here
        const int idx = nbrW[(ti * GK + kk) * 16 + lr];
        const u32x4 *p = (idx >= 0 ? a.feat + (size_t)idx * a.ldi : g_zero_row) + lc * 2;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          dst[ti][kb][0] = p[kb * 8];
          dst[ti][kb][1] = p[kb * 8 + 1];
        }
      }
    };
    auto to_operand = [&](int kk, u32x4 (&src)[TMAX][KB][2], u32x4 (&dst)[TMAX][KB][2]) {
#pragma unroll
      for (int ti = 0; ti < TMAX; ++ti) {
        if (!((tmask[ti] >> kk) & 1u)) continue;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int d = 0; d < 4; ++d) dst[ti][kb][q][d] = (unsigned)__builtin_amdgcn_ds_bpermute(perm_addr, (int)src[ti][kb][q][d]);
      }
    };
    unsigned m = any;
    int kk = __builtin_ctz(m);
    m &= m - 1;
    gather(kk, nxt);
    to_operand(kk, nxt, cur);
    while (true) {
      const int kn = m ? __builtin_ctz(m) : kk;                    // past the end: re-read the last offset (never used)
      gather(kn, nxt);
      const u32x4 *wb = Wl + (size_t)kk * KB * WQ + lane;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const u32x4 bh = wb[(kb * CT + ct) * 128], bl = wb[(kb * CT + ct) * 128 + 64];
#pragma unroll
          for (int ti = 0; ti < TMAX; ++ti) {
            if (!((tmask[ti] >> kk) & 1u)) continue;
            acc[ti][ct] = DF3D_MFMA_BF16(cur[ti][kb][1], bh, acc[ti][ct]);
            acc[ti][ct] = DF3D_MFMA_BF16(cur[ti][kb][0], bl, acc[ti][ct]);
            acc[ti][ct] = DF3D_MFMA_BF16(cur[ti][kb][0], bh, acc[ti][ct]);
          }
        }
      }
      if (!m) break;
      m &= m - 1;
      kk = kn;
      to_operand(kk, nxt, cur);
    }
  }

  // ---- epilogue: lane (n, g) holds rows 4g .. 4g+3 of a tile, columns n * CT .. n * CT + CT - 1 (packed-weight layout 1) ----
  if constexpr (CT == 2) {
    const int col = n * 2;
    const float2 bi = a.bias ? *(const float2 *)(a.bias + col) : make_float2(0.f, 0.f);
    const float2 sc = a.scale ? *(const float2 *)(a.scale + col) : make_float2(1.f, 1.f);
    const float2 sh = a.shift ? *(const float2 *)(a.shift + col) : make_float2(0.f, 0.f);
#pragma unroll
    for (int ti = 0; ti < TMAX; ++ti) {
      if (ti >= my_tiles) break;
      const int row0 = (tile0 + ti) * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + 4 * g + r;
        if (row >= a.n_out) continue;
        float2 v = make_float2((acc[ti][0][r] + bi.x) * sc.x + sh.x, (acc[ti][1][r] + bi.y) * sc.y + sh.y);
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
          char *blk = (char *)a.out_split + (o >> 3) * 32 + (n & 3) * 4;     // 8-channel block = [hi 16 B | lo 16 B]
          *(unsigned *)blk = hp;
          *(unsigned *)(blk + 16) = lp;
        }
      }
    }
  } else {
    f32x4 bi[CT / 4], sc[CT / 4], sh[CT / 4];
#pragma unroll
    for (int q = 0; q < CT / 4; ++q) {
      const int col = n * CT + q * 4;
      bi[q] = a.bias ? *(const f32x4 *)(a.bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
      sc[q] = a.scale ? *(const f32x4 *)(a.scale + col) : (f32x4){1.f, 1.f, 1.f, 1.f};
      sh[q] = a.shift ? *(const f32x4 *)(a.shift + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ti = 0; ti < TMAX; ++ti) {
      if (ti >= my_tiles) break;
      const int row0 = (tile0 + ti) * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + 4 * g + r;
        if (row >= a.n_out) continue;
        const size_t o = (size_t)row * a.ldo + n * CT;
        unsigned h[CT / 2], l[CT / 2];
#pragma unroll
        for (int q = 0; q < CT / 4; ++q) {
          f32x4 v = (f32x4){acc[ti][q * 4][r], acc[ti][q * 4 + 1][r], acc[ti][q * 4 + 2][r], acc[ti][q * 4 + 3][r]};
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
  }
}

// DF3D_CONV_WS: 1 = on for the served shapes, 0 = off (default decided by measurement, see launch_ws)
static int ws_mode_env() {
  const char *t = getenv("DF3D_CONV_WS");          // read per call: the tests switch it inside one process
  return t ? atoi(t) : -1;
}

static bool ws_applicable(const SplitConvArgs &a) {
  return !a.cols && !a.order && a.gy == 1 && a.in_goff == 0 && a.K >= 9 && a.n_out >= 4096;
}

// shapes the kernel serves by default (measured faster than the output-stationary kernels on MI355X; the others: DF3D_CONV_WS=1)
template <int CIN, int COUT>
static constexpr bool ws_default() {
  return false;
}

template <int CIN, int COUT>
static int launch_ws(const SplitConvArgs &a, hipStream_t stream) {
  constexpr int KB = CIN / 32, CT = COUT / 16, NW = DF3D_WS_NW;
  constexpr int per_offset = KB * CT * 2 * 64 * 16;                       // filter bytes per offset
  constexpr int TMAX = DF3D_WS_NW == 16 ? (CT <= 4 ? 2 : 1) : (CT <= 4 ? 3 : 2);
  constexpr int unit = per_offset + NW * TMAX * 16 * 4;                   // filters + table strips per offset
  constexpr int GK = (158 * 1024) / unit >= 27 ? 27 : (158 * 1024) / unit;   // offsets per LDS group
  static_assert(GK >= 1, "one offset's filters must fit the LDS");
  static int num_cu = 0;
  if (!num_cu) {
    hipDeviceProp_t p;
    num_cu = (hipGetDeviceProperties(&p, 0) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
  }
  const int ntiles = cdiv(a.n_out, 16);
  // tiles per workgroup: the chip's share, rounded up to whole rounds of the NW waves (every wave the same tile count)
  int per_wave = cdiv(cdiv(ntiles, num_cu), NW);
  if (per_wave > TMAX) per_wave = TMAX;
  if (per_wave < 1) per_wave = 1;
  const int tiles_per_wg = per_wave * NW;
  const size_t lds = (size_t)GK * per_offset + (size_t)NW * TMAX * GK * 16 * 4;
  static_assert((size_t)GK * per_offset + (size_t)NW * TMAX * GK * 16 * 4 <= 160 * 1024, "LDS budget");
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute((const void *)spconv_ws_kernel<CIN, COUT, GK, NW, TMAX>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("hipFuncSetAttribute(max dynamic LDS) failed: %s", hipGetErrorString(e));
      return DF3D_EHIP;
    }
    configured = true;
  }
  hipLaunchKernelGGL((spconv_ws_kernel<CIN, COUT, GK, NW, TMAX>), dim3(cdiv(ntiles, tiles_per_wg)), dim3(NW * 64), lds, stream,
                     a, tiles_per_wg);
  return DF3D_OK;
}
