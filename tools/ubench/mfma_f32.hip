// micro-benchmark: fp32 MFMA issue rate with 1 wave per SIMD, 2 accumulator chains (the conv kernel's pattern)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0) {
  f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
  float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c1, 0, 0, 0);
    }
  }
  long long t1 = clock64();
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1];
  if (threadIdx.x == 0 && blockIdx.x == 0) ((long long *)out)[1 << 20] = t1 - t0;
}
int main() {
  float *d;
  hipMalloc(&d, (1 << 23) + 64);
  for (int wg : {256, 512, 1024}) {
    int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(wg), dim3(256), 0, 0, d, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(wg), dim3(256), 0, 0, d, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc; hipMemcpy(&cyc, (char *)d + 8 * (1 << 20), 8, hipMemcpyDeviceToHost);
    double nm = (double)wg * 4 * iters * 64;
    printf("wgs=%d: %.3f ms, %.1f TF, %.1f cycles/MFMA/wave (memtime clk), wall-implied %.2f GHz at 32cyc\n", wg, ms,
           nm * 2 * 16 * 16 * 4 / ms / 1e9, (double)cyc / (iters * 64.0), nm * 32 / (wg < 256 ? wg : 256) / 4 / (ms * 1e-3) / 1e9 / (wg / 256.0 > 1 ? 1 : 1));
  }
  return 0;
}
