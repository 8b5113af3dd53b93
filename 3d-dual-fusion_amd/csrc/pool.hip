// Sparse max pooling and dynamic voxelisation: the two remaining entry points of the reference's extension modules on
// this path's boundary (sparse_conv_ext.indice_maxpool_* / voxel_layer.dynamic_voxelize).
//
// Reference max pool: TF/mmdet3d/ops/spconv/include/spconv/pool_ops.h:26-94 -- output = zeros, then per kernel offset k
// one launch over its rulebook pairs: out[o] = max(out[o], in[i]) (src/maxpool.cc:22-41, maxpool_cuda.cu); backward
// din[i] += dout[o] for every pair with out[o] == in[i] (:43-66); 27 launches + a D2H copy of the pair counts each.
// Here both directions are one output-stationary launch over the neighbour table: forward rows = outputs (note the
// reference's zero initialisation: the result is max(0, inputs), kept bug for bug), backward rows = INPUTS through
// the inverse table (no atomics; contributions are added in offset order like the reference's loop, so the sums are
// bit-identical).
//
// Reference dynamic voxelisation: TF/mmdet3d/ops/voxel/src/voxelization_cpu.cpp:8-41 / voxelization_cuda.cu:11-45 --
// per point c = floor((p - min) / voxel_size) per axis, (z, y, x) order, all three -1 if any axis is out of range.
#include "common.h"

namespace df3d {

// thread = (output row, 4 channels)
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float *__restrict__ feat, const int32_t *__restrict__ nbr,
                                                          int kvol, int n_out, int c4, float *__restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)n_out * c4) return;
  const int o = (int)(i / c4), q = (int)(i - (long long)o * c4);
  float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < kvol; ++k) {
    const int src = nbr[(size_t)k * n_out + o];
    if (src < 0) continue;
    const float4 v = *(const float4 *)(feat + ((size_t)src * c4 + q) * 4);
    m.x = m.x < v.x ? v.x : m.x;
    m.y = m.y < v.y ? v.y : m.y;
    m.z = m.z < v.z ? v.z : m.z;
    m.w = m.w < v.w ? v.w : m.w;
  }
  *(float4 *)(out + ((size_t)o * c4 + q) * 4) = m;
}

// thread = (input row, 4 channels); inv [K, n_in]: output row fed by input i at offset k
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float *__restrict__ feat, const float *__restrict__ out,
                                                          const float *__restrict__ gout, const int32_t *__restrict__ inv,
                                                          int kvol, int n_in, int c4, float *__restrict__ gin) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)n_in * c4) return;
  const int r = (int)(i / c4), q = (int)(i - (long long)r * c4);
  const float4 x = *(const float4 *)(feat + ((size_t)r * c4 + q) * 4);
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < kvol; ++k) {
    const int o = inv[(size_t)k * n_in + r];
    if (o < 0) continue;
    const float4 y = *(const float4 *)(out + ((size_t)o * c4 + q) * 4);
    const float4 d = *(const float4 *)(gout + ((size_t)o * c4 + q) * 4);
    if (y.x == x.x) g.x += d.x;
    if (y.y == x.y) g.y += d.y;
    if (y.z == x.z) g.z += d.z;
    if (y.w == x.w) g.w += d.w;
  }
  *(float4 *)(gin + ((size_t)r * c4 + q) * 4) = g;
}

struct DynVoxArgs {
  float vs[3], lo[3];
  int grid[3];
};

__global__ __launch_bounds__(256) void dynamic_voxelize_kernel(const float *__restrict__ points, long long n, int nfeat,
                                                               DynVoxArgs a, int32_t *__restrict__ coors) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int c[3] = {0, 0, 0};
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float f = floorf((points[i * nfeat + j] - a.lo[j]) / a.vs[j]);
    if (f >= 0.f && f < (float)a.grid[j]) c[j] = (int)f;         // NaN / out of range: the point is dropped
    else ok = false;
  }
  coors[i * 3] = ok ? c[2] : -1;
  coors[i * 3 + 1] = ok ? c[1] : -1;
  coors[i * 3 + 2] = ok ? c[0] : -1;
}

}  // namespace df3d

using namespace df3d;

extern "C" int df3d_sparse_maxpool(const float *features, int n_in, int channels, const int32_t *nbr, int kvol, int n_out,
                                   float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(kvol > 0 && kvol <= DF3D_MAX_KVOL && channels > 0 && channels % 4 == 0 && n_in >= 0 && n_out >= 0,
                 "sparse_maxpool: bad sizes (channels must be a multiple of 4, got %d)", channels);
  if (n_out == 0) return DF3D_OK;
  DF3D_CHECK_ARG(features && nbr && out, "sparse_maxpool: null argument");
  const long long total = (long long)n_out * (channels / 4);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, features, nbr, kvol, n_out,
                     channels / 4, out);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_sparse_maxpool_backward(const float *features, const float *out_features, const float *grad_out,
                                            int n_in, int channels, const int32_t *inv, int kvol, float *grad_in,
                                            void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(kvol > 0 && kvol <= DF3D_MAX_KVOL && channels > 0 && channels % 4 == 0 && n_in >= 0,
                 "sparse_maxpool_backward: bad sizes (channels must be a multiple of 4, got %d)", channels);
  if (n_in == 0) return DF3D_OK;
  DF3D_CHECK_ARG(features && out_features && grad_out && inv && grad_in, "sparse_maxpool_backward: null argument");
  const long long total = (long long)n_in * (channels / 4);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, features, out_features, grad_out, inv,
                     kvol, n_in, channels / 4, grad_in);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_dynamic_voxelize(const float *points, long long num_points, int num_features, const float *voxel_size,
                                     const float *coors_range, int32_t *coors, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(num_points >= 0 && num_features >= 3 && voxel_size && coors_range, "dynamic_voxelize: bad arguments");
  if (num_points == 0) return DF3D_OK;
  DF3D_CHECK_ARG(points && coors, "dynamic_voxelize: null argument");
  DynVoxArgs a;
  for (int j = 0; j < 3; ++j) {
    DF3D_CHECK_ARG(voxel_size[j] > 0.f, "dynamic_voxelize: voxel_size must be positive");
    a.vs[j] = voxel_size[j];
    a.lo[j] = coors_range[j];
    a.grid[j] = (int)roundf((coors_range[3 + j] - coors_range[j]) / voxel_size[j]);    // voxelization_cpu.cpp:150-153
  }
  hipLaunchKernelGGL(dynamic_voxelize_kernel, dim3(cdiv(num_points, 256)), dim3(256), 0, stream, points, num_points,
                     num_features, a, coors);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}
