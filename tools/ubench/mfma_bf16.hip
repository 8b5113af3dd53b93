// Raw issue rate of v_mfma_f32_16x16x32_bf16 on gfx950 as a function of the number of independent accumulators
// (dependency distance) and waves per SIMD.  hipcc --offload-arch=gfx950 -O3 -o mfma_bf16 mfma_bf16.hip && ./mfma_bf16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(512) void k(float *out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (__bf16)(float)(threadIdx.x + e);
    b[e] = (__bf16)(float)(threadIdx.x * 3 + e);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 24 / NACC; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  f32x4 s = acc[0];
  for (int i = 1; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int NACC>
void run(int threads, float *out) {
  const int iters = 2000, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double flop = 2.0 * 16 * 16 * 32 * 24.0 * iters * (threads / 64) * blocks;
  double per_mfma_cycles = ms * 1e-3 * 2.4e9 / (24.0 * iters * (threads / 64) / 4.0);
  printf("accs %2d waves/WG %2d (1 WG per CU): %8.1f TFLOP/s, %.1f cycles@2.4GHz per MFMA per SIMD\n", NACC, threads / 64,
         flop / ms / 1e9, per_mfma_cycles);
}

int main() {
  float *out;
  hipMalloc(&out, 256 * 512 * 4);
  for (int t : {256, 512}) {
    run<1>(t, out);
    run<2>(t, out);
    run<3>(t, out);
    run<4>(t, out);
    run<8>(t, out);
    run<12>(t, out);
  }
  return 0;
}
