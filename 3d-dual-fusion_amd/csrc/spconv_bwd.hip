// Sparse convolution backward.
//
// Reference: indiceConvBackward (TF/mmdet3d/ops/spconv/include/spconv/spconv_ops.h:363-456): per kernel offset k --
// gather the input rows and the output-gradient rows of its rulebook pairs into two buffers, filtersGrad[k] =
// in_buf^T . out_buf (GEMM), in_buf = out_buf . W[k]^T (GEMM), scatter-add in_buf into inputGrad; 27 x (2 gathers +
// 2 GEMMs + 1 scatter-add) launches and a D2H copy of the pair counts.
//
// Here:
//   input gradient   = the FORWARD kernel on the output gradient with transposed filters and the inverse neighbour
//                      table inv[k][i] = o (df3d_invert_neighbors; for submanifold convolutions the inverse of offset
//                      k is the forward table of the mirrored offset K-1-k, so nothing is built at all).  Output-
//                      stationary in the input rows: no atomics, no scatter pass, fused into one launch.
//   filter gradient  = df3d_sparse_conv_grad_filters: grid (row slice, offset, 64-channel group); a wave owns a
//                      16-input-channel x COUT tile of filtersGrad[k] in fp32 MFMA accumulators
//                      (v_mfma_f32_16x16x4_f32: the contraction runs over ROWS, 4 per instruction, operands are
//                      read straight from the feature / gradient rows -- 16 consecutive floats per row and lane
//                      group, no transposition), and adds its partial tile to HBM with fp32 atomics once.
#include "common.h"

namespace df3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void invert_fill_kernel(int32_t *__restrict__ inv, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) inv[i] = -1;
}

__global__ __launch_bounds__(256) void invert_neighbors_kernel(const int32_t *__restrict__ nbr, int kvol, int n_out,
                                                               int n_in, int32_t *__restrict__ inv) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)kvol * n_out) return;
  const int k = (int)(i / n_out), o = (int)(i - (size_t)k * n_out);
  const int src = nbr[i];
  if (src >= 0 && src < n_in) inv[(size_t)k * n_in + src] = o;       // a (k, input) pair has at most one output
}

constexpr int GF_ROWS = 1024;      // output rows per workgroup slice

template <int CT /*16-wide output-channel tiles*/>
__global__ __launch_bounds__(256) void grad_filters_kernel(const float *__restrict__ feat, const float *__restrict__ gout,
                                                           const int32_t *__restrict__ nbr, int n_out, int cin, int cout,
                                                           float *__restrict__ gw) {
  const int k = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ci_tile = blockIdx.z * 4 + wave;
  if (ci_tile * 16 >= cin) return;
  const int m = lane & 15, kg = lane >> 4;
  const int ci = ci_tile * 16 + m;
  const bool ci_ok = ci < cin;
  const int r0 = blockIdx.x * GF_ROWS, r1 = min(r0 + GF_ROWS, n_out);
  const int32_t *nb = nbr + (size_t)k * n_out;
  f32x4 acc[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bool any = false;
  for (int base = r0; base < r1; base += 16) {         // 4 MFMA k-steps per iteration: rows base + 4*u + kg
    int idx[4];
    float a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int o = base + 4 * u + kg;
      idx[u] = o < r1 ? nb[o] : -1;
    }
    if (__ballot(idx[0] >= 0 || idx[1] >= 0 || idx[2] >= 0 || idx[3] >= 0) == 0ull) continue;   // no pair in 16 rows
    any = true;
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = (idx[u] >= 0 && ci_ok) ? feat[(size_t)idx[u] * cin + ci] : 0.f;
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      const int co = t * 16 + m;
      float b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int o = base + 4 * u + kg;
        b[u] = (idx[u] >= 0 && co < cout) ? gout[(size_t)o * cout + co] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc[t], 0, 0, 0);
    }
  }
  if (!any) return;
  // accumulator lane (col j = lane & 15, rows 4*(lane >> 4) + r): D[i][j] = sum_rows in[row][ci_tile*16 + i] * gout[row][t*16 + j]
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const int co = t * 16 + m;
    if (co >= cout) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = ci_tile * 16 + 4 * kg + r;
      if (i < cin && acc[t][r] != 0.f) unsafeAtomicAdd(gw + ((size_t)k * cin + i) * cout + co, acc[t][r]);
    }
  }
}

}  // namespace df3d

using namespace df3d;

extern "C" int df3d_invert_neighbors(const int32_t *nbr, int kvol, int n_out, int n_in, int32_t *inv, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(kvol > 0 && kvol <= DF3D_MAX_KVOL && n_out >= 0 && n_in >= 0, "invert_neighbors: bad sizes");
  if (n_in == 0) return DF3D_OK;
  DF3D_CHECK_ARG(inv, "invert_neighbors: null output");
  const size_t n = (size_t)kvol * n_in;
  hipLaunchKernelGGL(invert_fill_kernel, dim3(cdiv((long long)n, 256)), dim3(256), 0, stream, inv, n);
  if (n_out > 0) {
    DF3D_CHECK_ARG(nbr, "invert_neighbors: null table");
    hipLaunchKernelGGL(invert_neighbors_kernel, dim3(cdiv((long long)kvol * n_out, 256)), dim3(256), 0, stream, nbr, kvol,
                       n_out, n_in, inv);
  }
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_sparse_conv_grad_filters(const float *features, int n_in, int cin, const float *grad_out, int n_out,
                                             int cout, const int32_t *nbr, int kvol, float *grad_filters,
                                             void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(kvol > 0 && kvol <= DF3D_MAX_KVOL && cin > 0 && cout > 0 && n_in >= 0 && n_out >= 0,
                 "sparse_conv_grad_filters: bad sizes");
  DF3D_CHECK_ARG(cout <= 128, "sparse_conv_grad_filters: at most 128 output channels (got %d)", cout);
  DF3D_CHECK_ARG(grad_filters, "sparse_conv_grad_filters: null output");
  DF3D_HIP(hipMemsetAsync(grad_filters, 0, (size_t)kvol * cin * cout * sizeof(float), stream));
  if (n_out == 0 || n_in == 0) return DF3D_OK;
  DF3D_CHECK_ARG(features && grad_out && nbr, "sparse_conv_grad_filters: null argument");
  const dim3 grid(cdiv(n_out, GF_ROWS), kvol, cdiv(cdiv(cin, 16), 4));
  const int ct = cdiv(cout, 16);
  if (ct <= 1) hipLaunchKernelGGL(grad_filters_kernel<1>, grid, dim3(256), 0, stream, features, grad_out, nbr, n_out, cin, cout, grad_filters);
  else if (ct <= 2) hipLaunchKernelGGL(grad_filters_kernel<2>, grid, dim3(256), 0, stream, features, grad_out, nbr, n_out, cin, cout, grad_filters);
  else if (ct <= 4) hipLaunchKernelGGL(grad_filters_kernel<4>, grid, dim3(256), 0, stream, features, grad_out, nbr, n_out, cin, cout, grad_filters);
  else hipLaunchKernelGGL(grad_filters_kernel<8>, grid, dim3(256), 0, stream, features, grad_out, nbr, n_out, cin, cout, grad_filters);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}
