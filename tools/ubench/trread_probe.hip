// ds_read_b64_tr_b16 on gfx950: which element does lane l get?  LDS holds u16 values = their own element index; every lane of a
// 16-lane group passes the address of "its" 8-byte chunk of a [4 rows][16 columns] block with a free row stride.
// build: hipcc --offload-arch=gfx950 -O2 -o trread_probe trread_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *out, int rowb) {
  __shared__ unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
  // rows g*8 + i/4, chunk i%4 (4 u16 each) of a tile at column offset 0; row stride rowb bytes
  unsigned addr = (unsigned)(size_t)lds + (unsigned)((g * 8 + (i >> 2)) * rowb + (i & 3) * 8);
  u32x2 r0, r1;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r0) : "v"(addr));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r1) : "v"(addr), "n"(4 * 160));
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1));
  out[lane * 4 + 0] = r0[0];
  out[lane * 4 + 1] = r0[1];
  out[lane * 4 + 2] = r1[0];
  out[lane * 4 + 3] = r1[1];
}
int main() {
  unsigned *d, h[256];
  hipMalloc(&d, sizeof(h));
  const int rowb = 160;
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, rowb);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int i = l & 15, g = l >> 4;
    for (int e = 0; e < 8; ++e) {
      const unsigned got = (h[l * 4 + e / 2] >> (16 * (e & 1))) & 0xffff;
      const unsigned want = ((g * 8 + e) * rowb) / 2 + i;       // row g*8+e, column i
      if (got != want) ++bad;
      if (l < 20 || got != want) if (l < 20) printf("lane %2d e %d: got elem %5u (row %u col %u) want row %d col %d\n", l, e, got, got * 2 / rowb, (got * 2 % rowb) / 2, g * 8 + e, i);
    }
  }
  printf("mismatches: %d\n", bad);
  return 0;
}
