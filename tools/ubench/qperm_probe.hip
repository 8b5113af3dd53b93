// Round 5: what bounds the K = 27 sub-manifold layers is the SHAPE of their gathers, not their bytes.
//
// An MFMA A operand wants lane (i = lane & 15, g = lane >> 4) to hold 16 bytes of row i: four consecutive lanes read four
// DIFFERENT rows, i.e. four cache lines -- the texture path looks up one line per clock, so a 1 KB wave load costs 64
// clocks instead of 16 (gather_probe.hip: 51-64 us against 25 us for conv3's gathers alone).  Round 3 fixed the shape with
// ds_bpermute and lost to the LDS pipe.  gfx950 has v_permlane16_swap / v_permlane32_swap: with FOUR operand registers in
// flight (hi / lo x two 32-channel blocks of a 64-channel row) the lane-index rotation  new[16 t + row] = old[4 q + t]
// decomposes into four transpositions between a lane bit and a register-select bit -- two by DPP + v_cndmask (lane bits 0, 1),
// two by the new swaps (lane bits 5, 4): 48 vector-ALU instructions per 4 KB, no LDS.
//
//   mode 0  operand-shaped loads (what the library's kernels do)          + 24 MFMAs per (16-row tile, offset)
//   mode 1  quad-contiguous loads + register transposition (this probe)   + the same MFMAs
//   mode 2  quad-contiguous loads, no transposition (wrong operands: the floor of the load shape)
// The accumulators of modes 0 and 1 must agree bit for bit (checked on the host).
//   hipcc --offload-arch=gfx950 -O3 -o qperm_probe qperm_probe.hip && ./qperm_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), C, 0, 0, 0)

// transposition of (register-select bit, lane bit b) for b = 0, 1 through DPP: a' = [a.even, b.even], b' = [a.odd, b.odd]
template <int BIT>
__device__ __forceinline__ void swap_lane_bit_dpp(unsigned &a, unsigned &b, bool hi) {
  // quad_perm encodings: [0,0,2,2] = 0xA0, [1,1,3,3] = 0xF5 (bit 0); [0,1,0,1] = 0x44, [2,3,2,3] = 0xEE (bit 1)
  constexpr int DOWN = BIT == 0 ? 0xA0 : 0x44, UP = BIT == 0 ? 0xF5 : 0xEE;
  const unsigned from_b = (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, DOWN, 0xf, 0xf, false);   // b[lane with the bit cleared]
  const unsigned from_a = (unsigned)__builtin_amdgcn_update_dpp(0, (int)a, UP, 0xf, 0xf, false);     // a[lane with the bit set]
  const unsigned na = hi ? from_b : a, nb = hi ? b : from_a;
  a = na, b = nb;
}
__device__ __forceinline__ void swap32(unsigned &a, unsigned &b) {
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0], b = r[1];
}
__device__ __forceinline__ void swap16(unsigned &a, unsigned &b) {
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = r[0], b = r[1];
}

// L[j] = what load j brought: lane 4 q + t holds (row 4 (q & 3) + j, operand register q >> 2, k-group t)
// -> L[m] = operand register m in MFMA layout: lane 16 t + row
__device__ __forceinline__ void to_operand_shape(u32x4 (&V)[4], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2;
  unsigned L[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int d = 0; d < 4; ++d) L[j][d] = V[j][d];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    swap_lane_bit_dpp<0>(L[0][d], L[1][d], b0);
    swap_lane_bit_dpp<0>(L[2][d], L[3][d], b0);
  }
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    swap_lane_bit_dpp<1>(L[0][d], L[2][d], b1);
    swap_lane_bit_dpp<1>(L[1][d], L[3][d], b1);
  }
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    swap32(L[0][d], L[2][d]);
    swap32(L[1][d], L[3][d]);
  }
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    swap16(L[0][d], L[1][d]);
    swap16(L[2][d], L[3][d]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int d = 0; d < 4; ++d) V[j][d] = L[j][d];
}

// rows: [n][8 blocks of 8 channels][hi 16 B | lo 16 B] (64 channels = 256 B); nbr [K][n_out]; every wave walks 16-row tiles
template <int MODE>
__global__ __launch_bounds__(256) void probe(const u32x4 *__restrict__ feat, const int32_t *__restrict__ nbr, int n_out, int K,
                                             const u32x4 *__restrict__ zero, float *__restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntiles = (n_out + 15) >> 4;
  const int n = lane & 15, g = lane >> 4, q = lane >> 2, t = lane & 3, qa = q & 3, m = q >> 2;
  // a fixed B operand per (register, column tile): any deterministic bits
  u32x4 B[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) B[c] = (u32x4){0x3c003c00u + lane * 17u + c, 0x38003a00u + lane, 0x34003800u + c * 5u, 0x3c003400u};
  // workgroups go to the eight XCDs round-robin: every XCD (= L2) walks one contiguous eighth of the tiles
  const int per_xcd = (ntiles + 7) / 8, xcd = blockIdx.x & 7;
  const int t_end = min(ntiles, (xcd + 1) * per_xcd);
  for (int tile = xcd * per_xcd + (blockIdx.x >> 3) * 4 + wave; tile < t_end; tile += (gridDim.x >> 3) * 4) {
    const int row0 = tile * 16;
    f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int k = 0; k < K; ++k) {
      u32x4 A[4];                                   // operand registers: m = 0 (kb 0, hi), 1 (kb 0, lo), 2 (kb 1, hi), 3 (kb 1, lo)
      if (MODE == 0) {
        const int row = row0 + n;
        const int idx = row < n_out ? nbr[(size_t)k * n_out + row] : -1;
        const u32x4 *p = idx >= 0 ? feat + (size_t)idx * 16 : zero;
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) A[mm] = p[(mm >> 1) * 8 + g * 2 + (mm & 1)];
      } else {
        const int off = (m >> 1) * 8 + t * 2 + (m & 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = row0 + 4 * qa + j;
          const int idx = row < n_out ? nbr[(size_t)k * n_out + row] : -1;
          const u32x4 *p = idx >= 0 ? feat + (size_t)idx * 16 : zero;
          A[j] = p[off];
        }
        if (MODE == 1) to_operand_shape(A, lane);
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          acc[c] = MFMA(A[kb * 2 + 1], B[c], acc[c]);
          acc[c] = MFMA(A[kb * 2], B[(c + 1) & 3], acc[c]);
          acc[c] = MFMA(A[kb * 2], B[c], acc[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + 4 * g + r;
        if (row < n_out) out[(size_t)row * 64 + c * 16 + n] = acc[c][r];
      }
  }
}

// Pipelined variant (what a production kernel would do): the tile's neighbour indices sit in LDS (a wave-private slab), the
// rows of offset k + D are gathered while offset k multiplies.  QUAD: quad-contiguous loads + register transposition.
template <bool QUAD, int D, bool MASKED = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4)))
void probe_pipe(const u32x4 *__restrict__ feat, const int32_t *__restrict__ nbr, int n_out, int K,
                const u32x4 *__restrict__ zero, float *__restrict__ out) {
  __shared__ int nbrL[4][27][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntiles = (n_out + 15) >> 4;
  const int n = lane & 15, g = lane >> 4, q = lane >> 2, t = lane & 3, qa = q & 3, m = q >> 2;
  u32x4 B[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) B[c] = (u32x4){0x3c003c00u + lane * 17u + c, 0x38003a00u + lane, 0x34003800u + c * 5u, 0x3c003400u};
  const int per_xcd = (ntiles + 7) / 8, xcd = blockIdx.x & 7;
  const int t_end = min(ntiles, (xcd + 1) * per_xcd);
  for (int tile = xcd * per_xcd + (blockIdx.x >> 3) * 4 + wave; tile < t_end; tile += (gridDim.x >> 3) * 4) {
    const int row0 = tile * 16;
    f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // the tile's table -> LDS: 27 x 16 entries, 7 per lane
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int e = lane + 64 * i;
      if (e < 27 * 16) {
        const int k = e >> 4, r = e & 15;
        nbrL[wave][k][r] = row0 + r < n_out ? nbr[(size_t)k * n_out + row0 + r] : -1;
      }
    }
    __builtin_amdgcn_wave_barrier();
    u32x4 A[D][4];
    auto gather = [&](int k, u32x4 (&dst)[4]) {
      if (QUAD) {
        const int off = (m >> 1) * 8 + t * 2 + (m & 1);
        const int4 ix = *(const int4 *)&nbrL[wave][k][4 * qa];
        const int id[4] = {ix.x, ix.y, ix.z, ix.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u32x4 *p = id[j] >= 0 ? feat + (size_t)id[j] * 16 : zero;
          dst[j] = p[off];
        }
      } else {
        const int id = nbrL[wave][k][n];
        if (MASKED) {                            // absent neighbours issue no request at all (exec-masked loads)
#pragma unroll
          for (int mm = 0; mm < 4; ++mm) dst[mm] = (u32x4){0u, 0u, 0u, 0u};
          if (id >= 0) {
            const u32x4 *p = feat + (size_t)id * 16;
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) dst[mm] = p[(mm >> 1) * 8 + g * 2 + (mm & 1)];
          }
        } else {
        const u32x4 *p = id >= 0 ? feat + (size_t)id * 16 : zero;
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) dst[mm] = p[(mm >> 1) * 8 + g * 2 + (mm & 1)];
        }
      }
    };
#pragma unroll
    for (int k = 0; k < D; ++k) gather(k, A[k]);
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      u32x4 cur[4];
#pragma unroll
      for (int mm = 0; mm < 4; ++mm) cur[mm] = A[k % D][mm];
      if (k + D < 27) gather(k + D, A[k % D]);
      if (QUAD) to_operand_shape(cur, lane);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          acc[c] = MFMA(cur[kb * 2 + 1], B[c], acc[c]);
          acc[c] = MFMA(cur[kb * 2], B[(c + 1) & 3], acc[c]);
          acc[c] = MFMA(cur[kb * 2], B[c], acc[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + 4 * g + r;
        if (row < n_out) out[(size_t)row * 64 + c * 16 + n] = acc[c][r];
      }
  }
}

int main() {
  // a two-voxel-thick ground disc on a 360 x 360 x 11 grid, rows sorted by (z, y, x): ~66 k voxels, ~14 of 27 neighbours
  const int Z = 11, Y = 360, X = 360;
  std::vector<int> cell((size_t)Z * Y * X, -1);
  std::vector<int> vz, vy, vx;
  srand(7);
  for (int z = 3; z <= 4; ++z)
    for (int y = 0; y < Y; ++y)
      for (int x = 0; x < X; ++x) {
        const double dy = y - 180.0, dx = x - 180.0;
        if (dy * dy + dx * dx < 115.0 * 115.0 && rand() % 100 < 80) {
          cell[((size_t)z * Y + y) * X + x] = (int)vz.size();
          vz.push_back(z), vy.push_back(y), vx.push_back(x);
        }
      }
  const int n = (int)vz.size(), K = 27;
  std::vector<int32_t> nbr((size_t)K * n, -1);
  long long pairs = 0;
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < K; ++k) {
      const int z = vz[i] + k / 9 - 1, y = vy[i] + (k / 3) % 3 - 1, x = vx[i] + k % 3 - 1;
      if (z < 0 || z >= Z || y < 0 || y >= Y || x < 0 || x >= X) continue;
      const int j = cell[((size_t)z * Y + y) * X + x];
      nbr[(size_t)k * n + i] = j;
      pairs += j >= 0;
    }
  printf("%d rows, %.1f of 27 neighbours per row, %.0f MB gathered per launch\n", n, (double)pairs / n, pairs * 256.0 / 1e6);
  std::vector<unsigned short> rows((size_t)n * 128);
  for (auto &v : rows) v = (unsigned short)(0x3000 + rand() % 0x0c00);       // fp16 values in [0.125, 1)
  u32x4 *dfeat, *dzero;
  int32_t *dnbr;
  float *dout[2];
  hipMalloc(&dfeat, rows.size() * 2);
  hipMalloc(&dzero, 4096);
  hipMemset(dzero, 0, 4096);
  hipMalloc(&dnbr, nbr.size() * 4);
  hipMalloc(&dout[0], (size_t)n * 64 * 4);
  hipMalloc(&dout[1], (size_t)n * 64 * 4);
  hipMemcpy(dfeat, rows.data(), rows.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dnbr, nbr.data(), nbr.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int grid : {256, 512, 1024}) {
    for (int mode = 3; mode < 8; ++mode) {
      float *o = dout[mode == 1 || mode == 4 || mode == 6];
      auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), 0, 0, dfeat, dnbr, n, K, dzero, o);
        if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), 0, 0, dfeat, dnbr, n, K, dzero, o);
        if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(256), 0, 0, dfeat, dnbr, n, K, dzero, o);
        if (mode == 3) hipLaunchKernelGGL((probe_pipe<false, 3>), dim3(grid), dim3(256), 0, 0, dfeat, dnbr, n, K, dzero, o);
        if (mode == 4) hipLaunchKernelGGL((probe_pipe<true, 3>), dim3(grid), dim3(256), 0, 0, dfeat, dnbr, n, K, dzero, o);
        if (mode == 5) hipLaunchKernelGGL((probe_pipe<false, 6>), dim3(grid), dim3(256), 0, 0, dfeat, dnbr, n, K, dzero, o);
        if (mode == 6) hipLaunchKernelGGL((probe_pipe<true, 6>), dim3(grid), dim3(256), 0, 0, dfeat, dnbr, n, K, dzero, o);
        if (mode == 7) hipLaunchKernelGGL((probe_pipe<false, 3, true>), dim3(grid), dim3(256), 0, 0, dfeat, dnbr, n, K, dzero, o);
      };
      for (int i = 0; i < 3; ++i) launch();
      hipEventRecord(e0);
      for (int i = 0; i < 20; ++i) launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("grid %4d mode %d: %6.1f us per launch\n", grid, mode, ms * 1e3 / 20);
    }
  }
  // modes 0 and 1 fed the same operands to the same MFMAs
  hipLaunchKernelGGL(probe<0>, dim3(512), dim3(256), 0, 0, dfeat, dnbr, n, K, dzero, dout[0]);
  hipLaunchKernelGGL((probe_pipe<true, 3>), dim3(512), dim3(256), 0, 0, dfeat, dnbr, n, K, dzero, dout[1]);
  std::vector<float> a((size_t)n * 64), b((size_t)n * 64);
  hipMemcpy(a.data(), dout[0], a.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), dout[1], b.size() * 4, hipMemcpyDeviceToHost);
  size_t bad = 0;
  for (size_t i = 0; i < a.size(); ++i) bad += memcmp(&a[i], &b[i], 4) != 0;
  printf("operand-shaped vs transposed quad loads: %zu of %zu accumulators differ%s\n", bad, a.size(), bad ? "  <-- WRONG" : " (bit-identical)");
  return bad != 0;
}
