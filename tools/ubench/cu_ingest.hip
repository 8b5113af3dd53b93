// What a compute unit can take in per clock on gfx950, by source and by path.  One workgroup per CU, every wave issues
// whole-line vector loads (8 lanes x 16 B = one 128-byte line, 8 lines per wave instruction) in batches of U with a
// full wait after each batch:
//   source  shared : every workgroup streams the same 1.75 MB buffer (the packed filter bank of a 128 -> 128 K = 27 layer)
//           gather : random 128-byte lines out of S MB (18 MB = the conv4 feature rows; 1 GB = HBM)
//   path    dma    : global_load_lds_dwordx4 into LDS           reg : global_load_dwordx4 into registers
// hipcc --offload-arch=gfx950 -O3 -o cu_ingest cu_ingest.hip && ./cu_ingest
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int U, bool DMA, bool GATHER>
__global__ __launch_bounds__(1024) void k(const u32x4 *buf, unsigned nlines, int iters, unsigned *sink) {
  __shared__ u32x4 L[DMA ? 16 : 1][U > 8 ? 8 : U][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  u32x4 keep = (u32x4){0, 0, 0, 0};
  unsigned seq = (blockIdx.x * 7919u + wave * 8u) % nlines;           // streaming position (in lines)
  for (int it = 0; it < iters; ++it) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      unsigned line;
      if (GATHER) line = hash32((blockIdx.x * 64u + wave) * 65536u + (it * U + u) * 8u + (lane >> 3)) % nlines;
      else { line = seq + (lane >> 3); seq += 8u * nw; if (seq + 8u >= nlines) seq -= (nlines - 8u) / (8u * nw) * (8u * nw); if (line >= nlines) line -= nlines; }
      const u32x4 *src = buf + (size_t)line * 8 + (lane & 7);
      if (DMA) __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)&L[wave][u & 7][0], 16, 0, 0);
      else { v[u] = *src; asm volatile("" : "+v"(v[u])); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!DMA) {
#pragma unroll
      for (int u = 0; u < U; ++u) keep ^= v[u];
    }
  }
  if (DMA) keep = L[wave][0][lane];
  if (keep[0] == 0x12345678u) sink[0] = keep[1];
}

template <int U, bool DMA, bool GATHER>
void run(const char *name, const u32x4 *buf, size_t bytes, int waves, unsigned *sink) {
  const int blocks = 256;
  const unsigned nlines = (unsigned)(bytes / 128);
  const int iters = 2000 / U * 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<U, DMA, GATHER>), dim3(blocks), dim3(waves * 64), 0, 0, buf, nlines, iters / 4, sink);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<U, DMA, GATHER>), dim3(blocks), dim3(waves * 64), 0, 0, buf, nlines, iters, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double total = 1024.0 * U * iters * waves * blocks;
  printf("%-28s %s U=%2d waves=%2d : %7.2f TB/s  = %5.1f B/clk/CU @2.4GHz   (%.3f ms)\n", name, DMA ? "dma" : "reg", U, waves,
         total / ms / 1e9, total / ms / 1e-3 / blocks / 2.4e9, ms);
}

template <bool DMA, bool GATHER>
void sweep(const char *name, const u32x4 *buf, size_t bytes, unsigned *sink) {
  for (int waves : {1, 2, 4, 8, 16}) {
    run<4, DMA, GATHER>(name, buf, bytes, waves, sink);
    run<8, DMA, GATHER>(name, buf, bytes, waves, sink);
    run<16, DMA, GATHER>(name, buf, bytes, waves, sink);
  }
}

int main() {
  u32x4 *buf;
  unsigned *sink;
  const size_t big = 1ull << 30;
  hipMalloc(&buf, big);
  hipMalloc(&sink, 64);
  hipMemset(buf, 1, big);
  hipDeviceSynchronize();
  sweep<true, false>("shared 1.75 MB stream", buf, 1792 * 1024, sink);
  sweep<false, false>("shared 1.75 MB stream", buf, 1792 * 1024, sink);
  sweep<true, true>("gather 2 MB", buf, 2u << 20, sink);
  sweep<true, true>("gather 18 MB", buf, 18u << 20, sink);
  sweep<false, true>("gather 18 MB", buf, 18u << 20, sink);
  sweep<true, true>("gather 1 GB", buf, big, sink);
  return 0;
}
