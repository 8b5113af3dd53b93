// What does the GATHER of a sparse convolution cost by itself on MI355X?  (round 3, tools/ubench/gather_probe.py)
// One kernel, three lane mappings of the same bytes: for every (16-row tile, offset) item a wave reads the 32-channel block
// kb (128 B: [hi 16 B | lo 16 B] x 4 sub-blocks) of each of the 16 neighbour rows, KB blocks per item:
//   mode 0  MFMA operand shape: lane (n = lane & 15, g = lane >> 4) reads row n, sub-block g: hi and lo as two 16-byte loads
//           (what every split-precision conv kernel of this library does);
//   mode 1  quad-coalesced: lane l reads row l >> 2, sub-block l & 3 (a quad of lanes = 128 contiguous bytes over two loads);
//   mode 2  line-coalesced: lane l reads row l >> 3, 16-byte piece l & 7 (8 lanes = one 128-byte line), 8 rows per
//           instruction, two instructions per block.
// The loaded words are xor-reduced into one store per wave so that nothing is optimised away.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int KB, int MODE>
__global__ __launch_bounds__(512) void gather_kernel(const u32x4 *__restrict__ feat, const int32_t *__restrict__ nbr, int n_out, int K,
                                                     int ldi, const u32x4 *__restrict__ zero, unsigned *__restrict__ sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntiles = (n_out + 15) >> 4;
  u32x4 acc = (u32x4){0u, 0u, 0u, 0u};
  for (int tile = blockIdx.x * 8 + wave; tile < ntiles; tile += gridDim.x * 8) {
    const int row0 = tile * 16;
    for (int k = 0; k < K; ++k) {
      int r;
      if (MODE == 0) r = lane & 15;
      else if (MODE == 1) r = lane >> 2;
      else r = lane >> 3;
#pragma unroll
      for (int half = 0; half < (MODE == 2 ? 2 : 1); ++half) {
        const int rr = MODE == 2 ? r + 8 * half : r;
        const int row = row0 + rr;
        const int idx = row < n_out ? nbr[(size_t)k * n_out + row] : -1;
        const u32x4 *p = idx >= 0 ? feat + (size_t)idx * ldi : zero;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          if (MODE == 0) {
            const int g = lane >> 4;
            acc ^= p[kb * 8 + g * 2];
            acc ^= p[kb * 8 + g * 2 + 1];
          } else if (MODE == 1) {
            const int g = lane & 3;
            acc ^= p[kb * 8 + g * 2];
            acc ^= p[kb * 8 + g * 2 + 1];
          } else {
            acc ^= p[kb * 8 + (lane & 7)];
          }
        }
      }
    }
  }
  unsigned v = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  for (int o = 32; o; o >>= 1) v ^= __shfl_xor(v, o, 64);
  if (lane == 0) sink[blockIdx.x * 8 + wave] = v;
}

extern "C" int gather_probe(const void *feat, const int32_t *nbr, int n_out, int K, int cin, int mode, int grid, const void *zero,
                            unsigned *sink, void *stream) {
  const int ldi = cin / 4;     // u32x4 per split row
  hipStream_t s = (hipStream_t)stream;
#define GO(KB, MODE) hipLaunchKernelGGL((gather_kernel<KB, MODE>), dim3(grid), dim3(512), 0, s, (const u32x4 *)feat, nbr, n_out, K, ldi, (const u32x4 *)zero, sink)
  const int kb = cin / 32;
  if (kb == 1) { if (mode == 0) GO(1, 0); else if (mode == 1) GO(1, 1); else GO(1, 2); }
  else if (kb == 2) { if (mode == 0) GO(2, 0); else if (mode == 1) GO(2, 1); else GO(2, 2); }
  else if (kb == 4) { if (mode == 0) GO(4, 0); else if (mode == 1) GO(4, 1); else GO(4, 2); }
  else return 1;
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
