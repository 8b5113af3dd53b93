// SparseConvTensor.dense() for gfx950: zero-fill + scatter + (N,D,H,W,C)->(N,C,D,H,W) permute
// in one pass over the features (TF/mmdet3d/ops/spconv/structure.py:5-18,55-64 does
// zeros + index_put + permute().contiguous(), i.e. writes the 33 MB volume twice).
// A block stages 64 voxel rows through LDS (coalesced 16 B reads along C) and writes, for
// each channel, the 64 voxels with consecutive lanes; rows that are sorted by flat index
// (every strided-conv output here) therefore produce mostly contiguous stores.
// Algorithmic bytes: N*C*4 read + B*C*D*H*W*4 written.  Bound: HBM.
#include "common.h"

namespace df3d {

__global__ __launch_bounds__(256) void dense_scatter_kernel(const float *__restrict__ feat,
                                                            const int32_t *__restrict__ ind, int n, int C,
                                                            int D, int H, int W, float *__restrict__ out) {
  extern __shared__ float tile[];  // [64][C+1]
  __shared__ long long base[64];
  const int n0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
  const int ld = C + 1;
  for (int e = tid; e < 64 * C; e += 256) {
    int r = e / C, c = e - r * C;
    tile[r * ld + c] = (n0 + r < n) ? feat[(size_t)(n0 + r) * C + c] : 0.f;
  }
  if (tid < 64) {
    long long b = -1;
    if (n0 + tid < n) {
      const int32_t *p = ind + (size_t)(n0 + tid) * 4;
      // offset of (b, c=0, z, y, x) in [B, C, D, H, W]
      b = ((long long)p[0] * C * D + p[1]) * H * W + (long long)p[2] * W + p[3];
    }
    base[tid] = b;
  }
  __syncthreads();
  const int r = tid & 63;
  const long long b = base[r];
  if (b < 0) return;
  const long long cstride = (long long)D * H * W;
  for (int c = tid >> 6; c < C; c += 4) out[b + c * cstride] = tile[r * ld + c];
}

}  // namespace df3d

using namespace df3d;

extern "C" int df3d_sparse_to_dense(const float *features, const int32_t *indices, int n, int channels, int batch,
                                    const int *shape, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(out && shape && batch > 0 && channels > 0, "sparse_to_dense: bad arguments");
  size_t total = (size_t)batch * channels * shape[0] * shape[1] * shape[2];
  DF3D_HIP(hipMemsetAsync(out, 0, total * sizeof(float), stream));
  if (n == 0) return DF3D_OK;
  DF3D_CHECK_ARG(features && indices, "sparse_to_dense: null input");
  size_t lds = (size_t)64 * (channels + 1) * sizeof(float);
  DF3D_CHECK_ARG(lds <= 150 * 1024, "sparse_to_dense: %d channels exceed the LDS tile", channels);
  hipLaunchKernelGGL(dense_scatter_kernel, dim3(cdiv(n, 64)), dim3(256), lds, stream, features, indices, n, channels,
                     shape[0], shape[1], shape[2], out);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}
