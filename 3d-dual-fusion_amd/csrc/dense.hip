// SparseConvTensor.dense() for gfx950: zero-fill + scatter + (N,D,H,W,C)->(N,C,D,H,W) permute
// in one pass over the features (TF/mmdet3d/ops/spconv/structure.py:5-18,55-64 does
// zeros + index_put + permute().contiguous(), i.e. writes the 33 MB volume twice).
// A block stages 64 voxel rows through LDS (coalesced 16 B reads along C) and writes, for
// each channel, the 64 voxels with consecutive lanes; rows that are sorted by flat index
// (every strided-conv output here) therefore produce mostly contiguous stores.
// Algorithmic bytes: N*C*4 read + B*C*D*H*W*4 written.  Bound: HBM.
#include "common.h"

namespace df3d {

DF3D_SPLIT_OVERFLOW_TU(dense)

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

// dense() directly in the layout the BEV neck consumes: rows [(b, y, x)][c*D + d] (channels-last; the channel
// order is the one `dense().view(N, C*D, H, W)` has, CP/det3d/models/backbones/scn.py:196-199).  One thread per
// (voxel, channel); rows of the output are 4*C*D bytes.
__global__ __launch_bounds__(256) void dense_rows_kernel(const float *__restrict__ feat, const int32_t *__restrict__ ind,
                                                         int n, int C, int D, int H, int W, float *__restrict__ out) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * C) return;
  int i = (int)(t / C), c = (int)(t - (long long)i * C);
  const int32_t *p = ind + (size_t)i * 4;
  size_t row = ((size_t)p[0] * H + p[2]) * W + p[3];
  out[row * ((size_t)C * D) + (size_t)c * D + p[1]] = feat[t];
}

// The same rows in the operand format of the split-precision convolutions ([row][col / 8][hi 8 x bf16 | lo 8 x bf16]): the
// neck's first convolution reads these; an all-zero buffer is the split of zero.
__global__ __launch_bounds__(256) void dense_rows_split_kernel(const float *__restrict__ feat, const int32_t *__restrict__ ind,
                                                               int n, int C, int D, int H, int W,
                                                               unsigned short *__restrict__ out) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * C) return;
  int i = (int)(t / C), c = (int)(t - (long long)i * C);
  const int32_t *p = ind + (size_t)i * 4;
  size_t row = ((size_t)p[0] * H + p[2]) * W + p[3];
  const size_t col = (size_t)c * D + p[1];
  unsigned h, l;
  split_pair(feat[t], 0.f, h, l);
  unsigned short *blk = out + (row * ((size_t)C * D) + (col & ~(size_t)7)) * 2;      // 16 halfwords per 8 columns
  blk[col & 7] = (unsigned short)(h & 0xffffu);
  blk[8 + (col & 7)] = (unsigned short)(l & 0xffffu);
}

// Neighbour table of a dense 2-D convolution over rows (b*H + y)*W + x: nbr[k][o], k = ky*kw + kx,
// input pixel (oy*stride - pad + ky, ox*stride - pad + kx) or -1 outside the map.
__global__ __launch_bounds__(256) void conv2d_neighbors_kernel(int B, int H, int W, int Ho, int Wo, int kh, int kw,
                                                               int stride, int pad, int32_t *__restrict__ nbr) {
  const long long n_out = (long long)B * Ho * Wo;
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_out * kh * kw) return;
  int k = (int)(t / n_out);
  long long o = t - (long long)k * n_out;
  int ox = (int)(o % Wo), oy = (int)((o / Wo) % Ho), b = (int)(o / ((long long)Wo * Ho));
  int iy = oy * stride - pad + k / kw, ix = ox * stride - pad + k % kw;
  nbr[t] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? (int32_t)(((long long)b * H + iy) * W + ix) : -1;
}

// Transposed convolution with kernel == stride == s (the neck's up-sampling): output pixel (oy, ox) reads input
// (oy/s, ox/s) through the single tap k = (oy%s)*s + ox%s.
__global__ __launch_bounds__(256) void deconv2d_neighbors_kernel(int B, int H, int W, int s, int32_t *__restrict__ nbr) {
  const int Ho = H * s, Wo = W * s;
  const long long n_out = (long long)B * Ho * Wo;
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_out * s * s) return;
  int k = (int)(t / n_out);
  long long o = t - (long long)k * n_out;
  int ox = (int)(o % Wo), oy = (int)((o / Wo) % Ho), b = (int)(o / ((long long)Wo * Ho));
  bool hit = ((oy % s) * s + (ox % s)) == k;
  nbr[t] = hit ? (int32_t)(((long long)b * H + oy / s) * W + ox / s) : -1;
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

extern "C" int df3d_sparse_to_dense_rows(const float *features, const int32_t *indices, int n, int channels, int batch,
                                         const int *shape, float *out_rows, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(out_rows && shape && batch > 0 && channels > 0, "sparse_to_dense_rows: bad arguments");
  size_t total = (size_t)batch * channels * shape[0] * shape[1] * shape[2];
  DF3D_HIP(hipMemsetAsync(out_rows, 0, total * sizeof(float), stream));
  if (n == 0) return DF3D_OK;
  DF3D_CHECK_ARG(features && indices, "sparse_to_dense_rows: null input");
  hipLaunchKernelGGL(dense_rows_kernel, dim3(cdiv((long long)n * channels, 256)), dim3(256), 0, stream, features,
                     indices, n, channels, shape[0], shape[1], shape[2], out_rows);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_sparse_to_dense_rows_split(const float *features, const int32_t *indices, int n, int channels, int batch,
                                               const int *shape, void *out_split, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(out_split && shape && batch > 0 && channels > 0, "sparse_to_dense_rows_split: bad arguments");
  DF3D_CHECK_ARG(((long long)channels * shape[0]) % 8 == 0, "sparse_to_dense_rows_split: %d columns per row (need a multiple of 8)",
                 channels * shape[0]);
  size_t total = (size_t)batch * channels * shape[0] * shape[1] * shape[2];
  DF3D_HIP(hipMemsetAsync(out_split, 0, total * sizeof(float), stream));
  if (n == 0) return DF3D_OK;
  DF3D_CHECK_ARG(features && indices, "sparse_to_dense_rows_split: null input");
  hipLaunchKernelGGL(dense_rows_split_kernel, dim3(cdiv((long long)n * channels, 256)), dim3(256), 0, stream, features, indices,
                     n, channels, shape[0], shape[1], shape[2], (unsigned short *)out_split);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_conv2d_neighbors(int batch, int H, int W, int kh, int kw, int stride, int pad, int transposed,
                                     int32_t *nbr, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(nbr && batch > 0 && H > 0 && W > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0,
                 "conv2d_neighbors: bad arguments");
  DF3D_CHECK_ARG(kh * kw <= DF3D_MAX_KVOL, "conv2d_neighbors: %d taps exceed %d", kh * kw, DF3D_MAX_KVOL);
  if (transposed) {
    DF3D_CHECK_ARG(kh == stride && kw == stride && pad == 0,
                   "conv2d_neighbors: transposed convolutions are served for kernel == stride, no padding");
    long long tot = (long long)batch * H * stride * W * stride * stride * stride;
    hipLaunchKernelGGL(deconv2d_neighbors_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, stream, batch, H, W, stride, nbr);
  } else {
    int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    DF3D_CHECK_ARG(Ho > 0 && Wo > 0, "conv2d_neighbors: empty output");
    long long tot = (long long)batch * Ho * Wo * kh * kw;
    hipLaunchKernelGGL(conv2d_neighbors_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, stream, batch, H, W, Ho, Wo, kh, kw,
                       stride, pad, nbr);
  }
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}
