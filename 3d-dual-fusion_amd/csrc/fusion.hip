// Camera-fusion glue of the CenterPoint 3D-DF adapter for gfx950.
//
// The reference (CP/det3d/models/fusion/voxel_with_point_projection.py:131-385,
// point_to_image_projection.py:63-231, model_utils/attention.py:31-61,422-468) walks
// 6 cameras x B samples x 3 scales in Python, with boolean-mask indexing, .unique() /
// .cpu() syncs and a double loop that pads the per-camera query sets.  Here:
//   project_voxels   one thread per (camera, voxel): voxel index -> LiDAR xyz -> camera ->
//                    pixel (the reference's three truncations) -> visibility mask
//   scatter_to_image "pts2img": voxel (features | xyz) rows dropped on the image plane,
//                    last writer wins (= highest voxel row, deterministic via atomicMax)
//   assemble_queries per-camera compaction into the zero-padded [B*ncam, max_ne, .] query
//                    tensors ACTR consumes, incl. the per-query image feature sample
//   writeback        features[row] += enh[b*ncam + cam][pos]  in camera order
// All are HBM/latency-bound gathers over <= ~10^5 rows.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace df3d {

DF3D_SPLIT_OVERFLOW_TU(fusion)

struct ProjArgs {
  const int32_t *ind;       // [n,4] (b,z,y,x)
  int n, batch, ncam;
  float sx, sy, sz, minx, miny, minz;
  const float *l2c;         // [B,ncam,4,4]
  const float *intr;        // [B,ncam,3,3]
  const int32_t *raw_hw;    // [B,ncam,2] (H, W) of the (scaled) input image
  const float *thres;       // [ncam]
  float image_scale;
  const float *feat_scale;  // [B,ncam,2] (feat_w/raw_w, feat_h/raw_h) as float32
  int32_t *grid;            // [ncam,n,2] (x, y) at feature-map resolution
  uint8_t *mask;            // [ncam,n]
  float *pinv;              // [n,3]
  float *depth;             // [ncam,n] or null
  const float *aug;         // [B][30] or null: translate (3), then three row-vector 3x3 factors (rescale, rotate, flip)
};

// k-ordered FMA chain, the accumulation a BLAS sgemm micro-kernel performs on the 4-vector
__device__ __forceinline__ float dot4(float a0, float a1, float a2, float a3, const float *m) {
#pragma clang fp contract(off)
  float acc = a0 * m[0];
  acc = fmaf(a1, m[1], acc);
  acc = fmaf(a2, m[2], acc);
  acc = fmaf(a3, m[3], acc);
  return acc;
}

__global__ __launch_bounds__(256) void project_voxels_kernel(ProjArgs a) {
#pragma clang fp contract(off)  // (HIP's __fmul_rn/__fadd_rn are inline a*b / a+b and still fuse after inlining)
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)a.n * a.ncam) return;
  int cam = (int)(t / a.n);
  int i = (int)(t - (long long)cam * a.n);
  const int32_t *p = a.ind + (size_t)i * 4;
  int b = p[0];
  // grid -> LiDAR (voxel corner): point_to_image_projection.py:82-103.  The reference's 4x4 matmul
  // rounds the product before adding the offset (k-ordered accumulation), so the compiler must
  // not contract these into FMAs: a 1-ulp change here flips pixel truncations further down.
  float px = (float)p[3] * a.sx;
  px = px + a.minx;
  float py = (float)p[2] * a.sy;
  py = py + a.miny;
  float pz = (float)p[1] * a.sz;
  pz = pz + a.minz;
  if (a.aug) {
    // the point the cameras saw: undo the recorded 3-D augmentations in the reference's order
    // (point_to_image_projection.py:121-128): p += translate; p = p @ rescale; p = p @ rotate; p = p @ flip.
    // Absent factors arrive as zeros / identities, which leave the value unchanged bit for bit.
    const float *g = a.aug + (size_t)b * 30;
    px = px + g[0];
    py = py + g[1];
    pz = pz + g[2];
#pragma unroll
    for (int f = 0; f < 3; ++f) {
      const float *m = g + 3 + 9 * f;
      const float qx = fmaf(pz, m[6], fmaf(py, m[3], px * m[0]));
      const float qy = fmaf(pz, m[7], fmaf(py, m[4], px * m[1]));
      const float qz = fmaf(pz, m[8], fmaf(py, m[5], px * m[2]));
      px = qx, py = qy, pz = qz;
    }
  }
  if (cam == 0) {
    a.pinv[(size_t)i * 3 + 0] = px;
    a.pinv[(size_t)i * 3 + 1] = py;
    a.pinv[(size_t)i * 3 + 2] = pz;
  }
  const float *M = a.l2c + ((size_t)b * a.ncam + cam) * 16;
  float cx = dot4(px, py, pz, 1.f, M + 0);
  float cy = dot4(px, py, pz, 1.f, M + 4);
  float cz = dot4(px, py, pz, 1.f, M + 8);
  const float *K = a.intr + ((size_t)b * a.ncam + cam) * 9;
  // camera_to_image (models/utils/transform_utils.py:39-60): [cx,cy,cz,1] @ pad4(K)^T, / z
  float uh = fmaf(cz, K[2], fmaf(cy, K[1], (cx * K[0])));
  float vh = fmaf(cz, K[5], fmaf(cy, K[4], (cx * K[3])));
  float wh = fmaf(cz, K[8], fmaf(cy, K[7], (cx * K[6])));
  float u = uh / wh, v = vh / wh;
  bool ok = isfinite(u) && isfinite(v) && fabsf(u) < 1e9f && fabsf(v) < 1e9f;
  long long gx = 0, gy = 0;
  if (ok) {
    gx = (long long)u;  // .long(): truncation toward zero
    gy = (long long)v;
    gx = (long long)(a.image_scale * (float)gx);
    gy = (long long)(a.image_scale * (float)gy);
    const int32_t *hw = a.raw_hw + ((size_t)b * a.ncam + cam) * 2;
    ok = gx > 0 && gx < hw[1] && gy > 0 && gy < hw[0] && cz > a.thres[cam];
  }
  int fx = 0, fy = 0;
  if (ok) {
    const float *fs = a.feat_scale + ((size_t)b * a.ncam + cam) * 2;
    fx = (int)((float)gx * fs[0]);
    fy = (int)((float)gy * fs[1]);
  }
  a.grid[((size_t)cam * a.n + i) * 2 + 0] = fx;
  a.grid[((size_t)cam * a.n + i) * 2 + 1] = fy;
  a.mask[(size_t)cam * a.n + i] = ok ? 1 : 0;
  if (a.depth) a.depth[(size_t)cam * a.n + i] = ok ? cz : 0.f;
}

struct ScatArgs {
  const float *feat;      // [n,C]
  const float *pinv;      // [n,3]
  const int32_t *ind;     // [n,4]
  const int32_t *grid;    // [ncam,n,2]
  const uint8_t *mask;    // [ncam,n]
  int n, C, ncam, H, W;
  int32_t *winner;        // [B*ncam, H, W] (init -1)
  float *canvas;          // [B*ncam, C+3, H, W] (init 0)
};

__global__ __launch_bounds__(256) void scatter_winner_kernel(ScatArgs a) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)a.n * a.ncam) return;
  int cam = (int)(t / a.n), i = (int)(t - (long long)cam * a.n);
  if (!a.mask[(size_t)cam * a.n + i]) return;
  int gx = a.grid[((size_t)cam * a.n + i) * 2], gy = a.grid[((size_t)cam * a.n + i) * 2 + 1];
  if (gx < 0 || gx >= a.W || gy < 0 || gy >= a.H) return;  // the reference's canvas has one spare row/col that is cropped
  int img = a.ind[(size_t)i * 4] * a.ncam + cam;
  atomicMax(&a.winner[((size_t)img * a.H + gy) * a.W + gx], i);
}

__global__ __launch_bounds__(256) void scatter_write_kernel(ScatArgs a) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)a.n * a.ncam) return;
  int cam = (int)(t / a.n), i = (int)(t - (long long)cam * a.n);
  if (!a.mask[(size_t)cam * a.n + i]) return;
  int gx = a.grid[((size_t)cam * a.n + i) * 2], gy = a.grid[((size_t)cam * a.n + i) * 2 + 1];
  if (gx < 0 || gx >= a.W || gy < 0 || gy >= a.H) return;
  int img = a.ind[(size_t)i * 4] * a.ncam + cam;
  size_t pix = (size_t)gy * a.W + gx;
  if (a.winner[(size_t)img * a.H * a.W + pix] != i) return;
  size_t hw = (size_t)a.H * a.W;
  float *dst = a.canvas + (size_t)img * (a.C + 3) * hw + pix;
  for (int c = 0; c < a.C; ++c) dst[(size_t)c * hw] = a.feat[(size_t)i * a.C + c];
  for (int c = 0; c < 3; ++c) dst[(size_t)(a.C + c) * hw] = a.pinv[(size_t)i * 3 + c];
}

struct AsmArgs {
  const float *feat;      // [n,C]
  const float *pinv;      // [n,3]
  const int32_t *ind;     // [n,4]
  const int32_t *grid;    // [ncam,n,2]
  const uint8_t *mask;    // [ncam,n]
  const int32_t *pos;     // [ncam,n] slot inside the (b,cam) list
  const float *img;       // [B*ncam, Ci, H, W]
  int n, C, Ci, ncam, H, W, max_ne;
  float *v_feat;          // [B*ncam, max_ne, C]
  float *v_i_feat;        // [B*ncam, max_ne, Ci]
  float *qgrid;           // [B*ncam, max_ne, 2]
  float *qpts;            // [B*ncam, max_ne, 3]
};

// one wave per (camera, voxel): lanes stride over the channels
__global__ __launch_bounds__(256) void assemble_queries_kernel(AsmArgs a) {
  long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (w >= (long long)a.n * a.ncam) return;
  int cam = (int)(w / a.n), i = (int)(w - (long long)cam * a.n);
  if (!a.mask[(size_t)cam * a.n + i]) return;
  int slot = a.pos[(size_t)cam * a.n + i];
  if (slot >= a.max_ne) return;
  int img = a.ind[(size_t)i * 4] * a.ncam + cam;
  size_t q = (size_t)img * a.max_ne + slot;
  int gx = a.grid[((size_t)cam * a.n + i) * 2], gy = a.grid[((size_t)cam * a.n + i) * 2 + 1];
  for (int c = lane; c < a.C; c += 64) a.v_feat[q * a.C + c] = a.feat[(size_t)i * a.C + c];
  size_t hw = (size_t)a.H * a.W;
  const float *src = a.img + (size_t)img * a.Ci * hw + (size_t)gy * a.W + gx;
  for (int c = lane; c < a.Ci; c += 64) a.v_i_feat[q * a.Ci + c] = src[(size_t)c * hw];
  if (lane == 0) {
    // voxel_with_point_projection.py:364: img_grid_b /= (W_feat, H_feat)
    a.qgrid[q * 2 + 0] = (float)gx / (float)a.W;
    a.qgrid[q * 2 + 1] = (float)gy / (float)a.H;
  }
  if (lane < 3) a.qpts[q * 3 + lane] = a.pinv[(size_t)i * 3 + lane];
}

// out[row] = feat[row] + sum over cameras (in camera order) of enh[b*ncam+cam][pos]
__global__ __launch_bounds__(256) void writeback_kernel(const float *__restrict__ feat,
                                                        const float *__restrict__ enh,
                                                        const int32_t *__restrict__ ind,
                                                        const uint8_t *__restrict__ mask,
                                                        const int32_t *__restrict__ pos, int n, int C, int ncam,
                                                        int max_ne, float *__restrict__ out) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * C) return;
  int i = (int)(t / C), c = (int)(t - (long long)i * C);
  float v = feat[t];
  int b = ind[(size_t)i * 4];
  for (int cam = 0; cam < ncam; ++cam) {
    if (mask[(size_t)cam * n + i]) {
      int slot = pos[(size_t)cam * n + i];
      if (slot < max_ne) v += enh[((size_t)(b * ncam + cam) * max_ne + slot) * C + c];
    }
  }
  out[t] = v;
}

// The same, eight channels per thread (two 16-byte loads per operand), optionally with the split rows (bf16 hi | lo, the
// operand format of the split-precision convolutions) of the result: the convolution behind the adapter reads those, a
// separate df3d_split_rows pass over the rows is not needed.  Same additions in the same (camera) order: bit-identical.
__global__ __launch_bounds__(256) void writeback8_kernel(const float *__restrict__ feat, const float *__restrict__ enh,
                                                         const int32_t *__restrict__ ind, const uint8_t *__restrict__ mask,
                                                         const int32_t *__restrict__ pos, int n, int C, int ncam, int max_ne,
                                                         float *__restrict__ out, u32x4 *__restrict__ out_split) {
  const int C8 = C >> 3;
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * C8) return;
  const int i = (int)(t / C8), c8 = (int)(t - (long long)i * C8);
  const float *f = feat + (size_t)i * C + c8 * 8;
  f32x4 v0 = *(const f32x4 *)f, v1 = *(const f32x4 *)(f + 4);
  const int b = ind[(size_t)i * 4];
  for (int cam = 0; cam < ncam; ++cam) {
    if (mask[(size_t)cam * n + i]) {
      const int slot = pos[(size_t)cam * n + i];
      if (slot < max_ne) {
        const float *e = enh + ((size_t)(b * ncam + cam) * max_ne + slot) * C + c8 * 8;
        v0 += *(const f32x4 *)e;
        v1 += *(const f32x4 *)(e + 4);
      }
    }
  }
  float *o = out + (size_t)i * C + c8 * 8;
  *(f32x4 *)o = v0;
  *(f32x4 *)(o + 4) = v1;
  if (out_split) {
    u32x4 h, l;
    split_pair(v0[0], v0[1], h[0], l[0]);
    split_pair(v0[2], v0[3], h[1], l[1]);
    split_pair(v1[0], v1[1], h[2], l[2]);
    split_pair(v1[2], v1[3], h[3], l[3]);
    out_split[((size_t)i * C8 + c8) * 2] = h;
    out_split[((size_t)i * C8 + c8) * 2 + 1] = l;
  }
}

}  // namespace df3d

using namespace df3d;

extern "C" int df3d_project_voxels(const int32_t *indices, int n, int batch, int ncam, const float *scale_xyz,
                                   const float *pc_min, const float *lidar2cam, const float *intrinsic,
                                   const int32_t *raw_hw, const float *depth_thres, float image_scale,
                                   const float *feat_scale, int32_t *grid_xy, uint8_t *mask, float *point_inv,
                                   float *depth, const float *aug_inv, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(indices && lidar2cam && intrinsic && raw_hw && depth_thres && feat_scale && grid_xy && mask &&
                     point_inv && scale_xyz && pc_min,
                 "project_voxels: null argument");
  if (n == 0) return DF3D_OK;
  ProjArgs a = {indices, n, batch, ncam, scale_xyz[0], scale_xyz[1], scale_xyz[2], pc_min[0], pc_min[1], pc_min[2],
                lidar2cam, intrinsic, raw_hw, depth_thres, image_scale, feat_scale, grid_xy, mask, point_inv, depth, aug_inv};
  hipLaunchKernelGGL(project_voxels_kernel, dim3(cdiv((long long)n * ncam, 256)), dim3(256), 0, stream, a);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_scatter_to_image(const float *features, const float *point_inv, const int32_t *indices,
                                     const int32_t *grid_xy, const uint8_t *mask, int n, int channels, int batch,
                                     int ncam, int H, int W, int32_t *winner, float *canvas, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(winner && canvas, "scatter_to_image: null output");
  size_t nimg = (size_t)batch * ncam;
  DF3D_HIP(hipMemsetAsync(winner, 0xff, nimg * H * W * sizeof(int32_t), stream));
  DF3D_HIP(hipMemsetAsync(canvas, 0, nimg * (channels + 3) * (size_t)H * W * sizeof(float), stream));
  if (n == 0) return DF3D_OK;
  DF3D_CHECK_ARG(features && point_inv && indices && grid_xy && mask, "scatter_to_image: null input");
  ScatArgs a = {features, point_inv, indices, grid_xy, mask, n, channels, ncam, H, W, winner, canvas};
  dim3 g(cdiv((long long)n * ncam, 256));
  hipLaunchKernelGGL(scatter_winner_kernel, g, dim3(256), 0, stream, a);
  hipLaunchKernelGGL(scatter_write_kernel, g, dim3(256), 0, stream, a);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_scatter_winner(const int32_t *indices, const int32_t *grid_xy, const uint8_t *mask, int n, int batch,
                                   int ncam, int H, int W, int32_t *winner, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(winner && batch > 0 && ncam > 0 && H > 0 && W > 0, "scatter_winner: bad arguments");
  DF3D_HIP(hipMemsetAsync(winner, 0xff, (size_t)batch * ncam * H * W * sizeof(int32_t), stream));
  if (n == 0) return DF3D_OK;
  DF3D_CHECK_ARG(indices && grid_xy && mask, "scatter_winner: null input");
  ScatArgs a = {nullptr, nullptr, indices, grid_xy, mask, n, 0, ncam, H, W, winner, nullptr};
  hipLaunchKernelGGL(scatter_winner_kernel, dim3(cdiv((long long)n * ncam, 256)), dim3(256), 0, stream, a);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_assemble_queries(const float *features, const float *point_inv, const int32_t *indices,
                                     const int32_t *grid_xy, const uint8_t *mask, const int32_t *pos,
                                     const float *img_feats, int n, int channels, int img_channels, int batch,
                                     int ncam, int H, int W, int max_ne, float *v_feat, float *v_i_feat, float *qgrid,
                                     float *qpts, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(v_feat && v_i_feat && qgrid && qpts, "assemble_queries: null output");
  size_t nq = (size_t)batch * ncam * max_ne;
  DF3D_HIP(hipMemsetAsync(v_feat, 0, nq * channels * sizeof(float), stream));
  DF3D_HIP(hipMemsetAsync(v_i_feat, 0, nq * img_channels * sizeof(float), stream));
  DF3D_HIP(hipMemsetAsync(qgrid, 0, nq * 2 * sizeof(float), stream));
  DF3D_HIP(hipMemsetAsync(qpts, 0, nq * 3 * sizeof(float), stream));
  if (n == 0 || max_ne == 0) return DF3D_OK;
  DF3D_CHECK_ARG(features && point_inv && indices && grid_xy && mask && pos && img_feats,
                 "assemble_queries: null input");
  AsmArgs a = {features, point_inv, indices, grid_xy, mask, pos, img_feats, n, channels, img_channels, ncam, H, W,
               max_ne, v_feat, v_i_feat, qgrid, qpts};
  hipLaunchKernelGGL(assemble_queries_kernel, dim3(cdiv((long long)n * ncam * 64, 256)), dim3(256), 0, stream, a);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_fusion_writeback(const float *features, const float *enh, const int32_t *indices,
                                     const uint8_t *mask, const int32_t *pos, int n, int channels, int ncam,
                                     int max_ne, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n == 0) return DF3D_OK;
  DF3D_CHECK_ARG(features && enh && indices && mask && pos && out, "fusion_writeback: null argument");
  if (channels % 8 == 0 && (((uintptr_t)features | (uintptr_t)enh | (uintptr_t)out) & 15) == 0) {
    // eight channels per thread, 16-byte loads (round 6: the one-channel kernel took 122 us for the 118 k x 128 rows of the
    // TransFusion tree); same additions in the same camera order
    hipLaunchKernelGGL(writeback8_kernel, dim3(cdiv((long long)n * (channels / 8), 256)), dim3(256), 0, stream, features, enh,
                       indices, mask, pos, n, channels, ncam, max_ne, out, (u32x4 *)nullptr);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
  }
  hipLaunchKernelGGL(writeback_kernel, dim3(cdiv((long long)n * channels, 256)), dim3(256), 0, stream, features, enh,
                     indices, mask, pos, n, channels, ncam, max_ne, out);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

// =========================================================================================
// Image-side gate without dense canvases.
//
// The reference gate (attention.py:31-61) builds two dense canvases [C+3, H, W] per image, runs 1x1
// convs over them, adds a 1-channel image summary, runs a 3x3 conv to ONE channel and a sigmoid.
// Every step before the sigmoid is linear, and the canvases are non-zero only at the few thousand
// pixels hit by a voxel, so the same map is
//     y[p] = b + sum_t inside(p+t) * (k_t + g_t * gate[p+t]) + sum_t S[t][p+t]
// where t runs over the 3x3 taps, k_t / g_t are scalars derived from the weights, gate is the
// 1-channel image summary and S[t] holds, at each voxel pixel, the tap-t response of that voxel's
// (already projected) feature row.  S is filled by scattering 9 scalars per winning voxel row.
// =========================================================================================
namespace df3d {

// S[img][t][gy][gx] += s[row][t] for the winner rows of one scale (winner map from scatter_winner_kernel)
__global__ __launch_bounds__(256) void gate_scatter_kernel(const float *__restrict__ s9, const int32_t *__restrict__ ind,
                                                           const int32_t *__restrict__ grid,
                                                           const uint8_t *__restrict__ mask,
                                                           const int32_t *__restrict__ winner, int n, int ncam, int H,
                                                           int W, float *__restrict__ S) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * ncam) return;
  int cam = (int)(t / n), i = (int)(t - (long long)cam * n);
  if (!mask[(size_t)cam * n + i]) return;
  int gx = grid[((size_t)cam * n + i) * 2], gy = grid[((size_t)cam * n + i) * 2 + 1];
  if (gx < 0 || gx >= W || gy < 0 || gy >= H) return;
  int img = ind[(size_t)i * 4] * ncam + cam;
  size_t pix = (size_t)gy * W + gx;
  if (winner[(size_t)img * H * W + pix] != i) return;
  size_t hw = (size_t)H * W;
  float *dst = S + (size_t)img * 9 * hw + pix;
#pragma unroll
  for (int k = 0; k < 9; ++k) dst[(size_t)k * hw] += s9[(size_t)i * 9 + k];   // row response is camera independent
}

// The same with the row responses computed on the fly (round 3): s[row][t] = <(features[row], pinv[row]), T[t]> only for the
// WINNER rows -- the caller's torch.cat + [n, C + 3] x [C + 3, 9] GEMM per scale (two launches and ~25 us for a product only
// the ~10 % winning rows of use) are gone.  T stands in LDS; all lanes of a wave read the same element (broadcast).
__global__ __launch_bounds__(256) void gate_scatter_rows_kernel(const float *__restrict__ feat, int C,
                                                                const float *__restrict__ pinv, const float *__restrict__ T,
                                                                const int32_t *__restrict__ winner, long long npix, int hw,
                                                                int clear, float *__restrict__ S) {
  // one thread per (image, pixel): the pixel's winner row (scatter_winner_kernel; -1 = none) is the only row that
  // contributes, so only it is multiplied; with `clear` the thread WRITES its nine values (zeros without a winner) and the
  // caller needs no zero fill of S
  extern __shared__ float Tl[];                     // [C + 3][9]
  const int CE = C + 3;
  for (int e = threadIdx.x; e < 9 * CE; e += 256) {
    const int k = e / CE, c = e - k * CE;
    Tl[c * 9 + k] = T[e];
  }
  __syncthreads();
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= npix) return;
  const long long img = t / hw;
  const int pix = (int)(t - img * hw);
  float *dst = S + (size_t)img * 9 * hw + pix;
  const int i = winner[t];
  if (i < 0) {
    if (clear) {
#pragma unroll
      for (int k = 0; k < 9; ++k) dst[(size_t)k * hw] = 0.f;
    }
    return;
  }
  float acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.f;
  const float *f = feat + (size_t)i * C;
  // (round 6: eight 16-byte loads of the row in flight per pass -- with one load per pass the few winner lanes of a wave walked
  // 32 dependent L2 round trips per 128-channel row: 55 us for 20 k rows; same summation order)
  for (int c0 = 0; c0 < C; c0 += 32) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = c0 + 4 * u < C ? *(const float4 *)(f + c0 + 4 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = c0 + 4 * u;
      if (c < C) {
        const float x[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int k = 0; k < 9; ++k) acc[k] = fmaf(x[j], Tl[(c + j) * 9 + k], acc[k]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float x = pinv[(size_t)i * 3 + j];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = fmaf(x, Tl[(C + j) * 9 + k], acc[k]);
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) dst[(size_t)k * hw] = clear ? acc[k] : dst[(size_t)k * hw] + acc[k];
}

// att[img][p] = sigmoid(bias + sum_t inside * (k[t] + g[t]*gate[p+t] + S[t][p+t]))
// BIAS: the image summary arrives without the bias of its 1x1 convolution (`gate_bias[0]` is added here, as the caller's
// separate element-wise pass did: same operation, same order)
template <bool BIAS>
__global__ __launch_bounds__(256) void gate_finish_kernel(const float *__restrict__ gate, const float *__restrict__ gate_bias,
                                                          const float *__restrict__ S,
                                                          const float *__restrict__ kg /* [9] k_t, [9] g_t, bias */,
                                                          int nimg, int H, int W, float *__restrict__ att) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  size_t hw = (size_t)H * W;
  if (t >= (long long)nimg * hw) return;
  int img = (int)(t / hw);
  int p = (int)(t - (long long)img * hw);
  int y = p / W, x = p - y * W;
  float acc = kg[18];
  const float gb = BIAS ? gate_bias[0] : 0.f;
#pragma unroll
  for (int ty = 0; ty < 3; ++ty)
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
      int yy = y + ty - 1, xx = x + tx - 1;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;   // zero padding of the 3x3 conv
      int k = ty * 3 + tx;
      size_t q = (size_t)yy * W + xx;
      const float gq = BIAS ? gate[(size_t)img * hw + q] + gb : gate[(size_t)img * hw + q];
      acc += kg[k] + kg[9 + k] * gq + S[((size_t)img * 9 + k) * hw + q];
    }
  att[t] = 1.f / (1.f + __expf(-acc));
}

// Pixels that carry a query: flag[img * HW + pixel] = 1 for every (camera, voxel) pair that is visible (round 4: the image
// projection copies exactly these pixels' raw rows into a compact pixel-major buffer)
__global__ __launch_bounds__(256) void mark_query_pixels_kernel(const int32_t *__restrict__ ind, const int32_t *__restrict__ grid,
                                                                const uint8_t *__restrict__ mask, int n, int ncam, int H, int W,
                                                                uint32_t *__restrict__ flag) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * ncam) return;
  const int cam = (int)(t / n), i = (int)(t - (long long)cam * n);
  if (!mask[t]) return;
  const int img = ind[(size_t)i * 4] * ncam + cam;
  const int gx = grid[t * 2], gy = grid[t * 2 + 1];
  flag[(size_t)img * H * W + (size_t)gy * W + gx] = 1u;
}

__global__ __launch_bounds__(256) void pixrow_kernel(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ rank, size_t n,
                                                     int32_t *__restrict__ pixrow) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pixrow[i] = flag[i] ? (int32_t)rank[i] : -1;
}

struct AsmArgs2 {
  const float *feat, *pinv;
  const int32_t *ind, *grid;
  const uint8_t *mask;
  const int32_t *pos;
  const float *img;       // UNGATED image features [B*ncam, Ci, H, W] ...
  const float *const *img_ptrs;   // ... or one [Ci, H, W] map per image (b*ncam + cam), when img is null
  const float *att;       // [B*ncam, H, W] or null
  int n, C, Ci, ncam, H, W, max_ne;
  float *v_feat, *v_i_feat, *qgrid, *qpts, *qpos;   // qpos [B*ncam, max_ne, C] depth sine embedding or null
  const int32_t *pixrow;  // round 4: with `compact`, the image features come from the pixel-major rows compact[pixrow[img][pix]]
  const float *compact;
};

// assemble_queries with (a) the gate applied on the fly to the sampled image feature and (b) the depth
// sine position embedding (position_encoding.py:107-120: x / 60 * 2pi over temperature^(2*(i//2)/C),
// sin on even / cos on odd channels) written directly, so neither a gated image nor a trig pass exists.
__global__ __launch_bounds__(256) void assemble_queries2_kernel(AsmArgs2 a) {
  long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (w >= (long long)a.n * a.ncam) return;
  int cam = (int)(w / a.n), i = (int)(w - (long long)cam * a.n);
  if (!a.mask[(size_t)cam * a.n + i]) return;
  int slot = a.pos[(size_t)cam * a.n + i];
  if (slot >= a.max_ne) return;
  int img = a.ind[(size_t)i * 4] * a.ncam + cam;
  size_t q = (size_t)img * a.max_ne + slot;
  int gx = a.grid[((size_t)cam * a.n + i) * 2], gy = a.grid[((size_t)cam * a.n + i) * 2 + 1];
  for (int c = lane; c < a.C; c += 64) a.v_feat[q * a.C + c] = a.feat[(size_t)i * a.C + c];
  size_t hw = (size_t)a.H * a.W;
  size_t pix = (size_t)gy * a.W + gx;
  float g = a.att ? a.att[(size_t)img * hw + pix] : 1.f;
  if (a.compact) {
    const float *src = a.compact + (size_t)a.pixrow[(size_t)img * hw + pix] * a.Ci;
    for (int c = lane; c < a.Ci; c += 64) a.v_i_feat[q * a.Ci + c] = src[c] * g;
  } else {
    const float *src = (a.img ? a.img + (size_t)img * a.Ci * hw : a.img_ptrs[img]) + pix;
    for (int c = lane; c < a.Ci; c += 64) a.v_i_feat[q * a.Ci + c] = src[(size_t)c * hw] * g;
  }
  if (lane == 0) {
    a.qgrid[q * 2 + 0] = (float)gx / (float)a.W;
    a.qgrid[q * 2 + 1] = (float)gy / (float)a.H;
  }
  if (lane < 3) a.qpts[q * 3 + lane] = a.pinv[(size_t)i * 3 + lane];
  if (a.qpos) {
    float d = a.pinv[(size_t)i * 3] / 60.f * 6.283185307179586f;
    for (int c = lane; c < a.C; c += 64) {
      float dim_t = powf(10000.f, (float)(2 * (c / 2)) / (float)a.C);
      float v = d / dim_t;
      a.qpos[q * a.C + c] = (c & 1) ? cosf(v) : sinf(v);
    }
  }
}

// assemble_queries2 with the LANES OVER QUERIES (round 3).  The kernel above gives a query to a wave and strides its lanes
// over the image channels of a channel-first map: 64 lanes = 64 different 64-byte sectors for 256 useful bytes (77 us for
// 18.6 k queries, 265 us at four samples: 3 % of the HBM rate).  Voxel rows are sorted by cell, so CONSECUTIVE SLOTS of a
// camera's query list are neighbouring voxels and project to neighbouring pixels: here a workgroup owns 64 consecutive slots
// of one image, lane = slot, and walks the channels -- one load instruction reads the 64 queries' pixels of one channel
// plane (a handful of sectors), the values meet in an LDS tile and leave as whole rows.  The voxel of a slot is found by
// binary search in the camera's monotone slot table (no inverse table, no extra launch).  Same values as the kernel above.
constexpr int ASM3_CH = 64;      // channels per LDS tile
__global__ __launch_bounds__(256) void assemble_queries3_kernel(AsmArgs2 a, const int32_t *__restrict__ counts) {
  __shared__ float tile[64][ASM3_CH + 1];
  __shared__ int rowL[64], pixL[64];
  __shared__ float gL[64];
  const int img = blockIdx.y, cam = img % a.ncam, b = img / a.ncam;
  const int cnt = counts[img] < a.max_ne ? counts[img] : a.max_ne;
  const int s0 = blockIdx.x * 64;
  if (s0 >= cnt) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t hw = (size_t)a.H * a.W;
  if (tid < 64) {
    int row = -1, pix = 0;
    float g = 0.f;
    const int s = s0 + tid;
    if (s < cnt) {
      // rows of sample b: [lo, hi); inside, pos[cam][.] counts the visible rows before each row
      int lo = 0, hi = a.n;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a.ind[(size_t)mid * 4] < b) lo = mid + 1;
        else hi = mid;
      }
      int r0 = lo;
      hi = a.n;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a.ind[(size_t)mid * 4] < b + 1) lo = mid + 1;
        else hi = mid;
      }
      int r1 = lo;
      const int32_t *pc = a.pos + (size_t)cam * a.n;
      const uint8_t *mc = a.mask + (size_t)cam * a.n;
      // pos is the EXCLUSIVE count of visible rows: it equals s on the invisible rows in front of the slot's voxel and on the
      // voxel itself, s + 1 behind it -- the voxel is the last row with pos <= s
      lo = r0, hi = r1;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (pc[mid] <= s) lo = mid + 1;
        else hi = mid;
      }
      row = lo - 1;
      if (!mc[row]) row = -1;                // cannot happen for s < cnt; guards a corrupt table
      if (row >= 0) {
        const int gx = a.grid[((size_t)cam * a.n + row) * 2], gy = a.grid[((size_t)cam * a.n + row) * 2 + 1];
        pix = gy * a.W + gx;
        g = a.att ? a.att[(size_t)img * hw + pix] : 1.f;
      }
      if (row >= 0 && blockIdx.z == 0) {
        const int gx = a.grid[((size_t)cam * a.n + row) * 2], gy = a.grid[((size_t)cam * a.n + row) * 2 + 1];
        const size_t q = (size_t)img * a.max_ne + s;
        a.qgrid[q * 2 + 0] = (float)gx / (float)a.W;
        a.qgrid[q * 2 + 1] = (float)gy / (float)a.H;
        a.qpts[q * 3 + 0] = a.pinv[(size_t)row * 3 + 0];
        a.qpts[q * 3 + 1] = a.pinv[(size_t)row * 3 + 1];
        a.qpts[q * 3 + 2] = a.pinv[(size_t)row * 3 + 2];
      }
    }
    rowL[tid] = row;
    pixL[tid] = pix;
    gL[tid] = g;
  }
  __syncthreads();
  const float *base = a.img ? a.img + (size_t)img * a.Ci * hw : a.img_ptrs[img];
  // image features: [channel chunk][slot] loads (lane = slot), [slot][channel] stores (lane = channel)
  for (int c0 = blockIdx.z * ASM3_CH; c0 < a.Ci; c0 += gridDim.z * ASM3_CH) {     // channel chunks over blockIdx.z
    const int myrow = rowL[lane];
    const float g = gL[lane];
    const float *src = base + pixL[lane];
#pragma unroll 4
    for (int c = wave; c < ASM3_CH; c += 4)
      tile[lane][c] = (myrow >= 0 && c0 + c < a.Ci) ? src[(size_t)(c0 + c) * hw] * g : 0.f;
    __syncthreads();
    for (int r = wave; r < 64; r += 4) {
      if (rowL[r] < 0 || c0 + lane >= a.Ci) continue;
      a.v_i_feat[((size_t)img * a.max_ne + s0 + r) * a.Ci + c0 + lane] = tile[r][lane];
    }
    __syncthreads();
  }
  // voxel features and the depth position embedding: row copies, lane = channel
  if (blockIdx.z != 0) return;
  for (int r = wave; r < 64; r += 4) {
    const int row = rowL[r];
    if (row < 0) continue;
    const size_t q = (size_t)img * a.max_ne + s0 + r;
    for (int c = lane; c < a.C; c += 64) a.v_feat[q * a.C + c] = a.feat[(size_t)row * a.C + c];
    if (a.qpos) {
      const float d = a.pinv[(size_t)row * 3] / 60.f * 6.283185307179586f;
      for (int c = lane; c < a.C; c += 64) {
        const float dim_t = powf(10000.f, (float)(2 * (c / 2)) / (float)a.C);
        const float v = d / dim_t;
        a.qpos[q * a.C + c] = (c & 1) ? cosf(v) : sinf(v);
      }
    }
  }
}

// position embedding of a padded (all-zero) query: sin(0) = 0 on even, cos(0) = 1 on odd channels
__global__ __launch_bounds__(256) void qpos_pad_kernel(float *__restrict__ qpos, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) qpos[i] = (i & 1) ? 1.f : 0.f;      // channel count is even, so parity of the flat index = channel parity
}

// Slot of every voxel inside its (sample, camera) query list = number of visible voxels of the same sample and
// camera before it; one block per (camera, sample) walks the sample's rows (indices are batch-sorted, as every
// strided-conv output is) with a ballot/popcount block scan.  counts[b*ncam + cam] = list length.
__global__ __launch_bounds__(1024) void query_slots_kernel(const uint8_t *__restrict__ mask,
                                                           const int32_t *__restrict__ ind, int n, int ncam,
                                                           int32_t *__restrict__ pos, int32_t *__restrict__ counts) {
  const int cam = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ int range[2];
  __shared__ int wsum[16];
  __shared__ int carry;
  if (tid < 2) {          // first row with batch >= b (tid 0) / >= b+1 (tid 1)
    int key = b + tid, lo = 0, hi = n;
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (ind[(size_t)mid * 4] < key) lo = mid + 1;
      else hi = mid;
    }
    range[tid] = lo;
  }
  if (tid == 0) carry = 0;
  __syncthreads();
  const int r0 = range[0], r1 = range[1];
  const uint8_t *m = mask + (size_t)cam * n;
  for (int base = r0; base < r1; base += 1024) {
    int r = base + tid;
    bool v = r < r1 && m[r] != 0;
    unsigned long long bal = __ballot(v);
    int within = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int off = carry;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (r < r1) pos[(size_t)cam * n + r] = off + within;
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int w = 0; w < 16; ++w) t += wsum[w];
      carry += t;
    }
    __syncthreads();
  }
  if (tid == 0) counts[b * ncam + cam] = carry;
}

// Round 4: the query rows BY SLOT.  assemble_queries2_kernel launches a wave per (camera, voxel) CANDIDATE -- 6 x 29 k waves of
// which 31 k own a slot -- and every live wave walks a chain of dependent loads (mask -> slot -> sample index -> pixel -> gate ->
// rows) before its gathers start: 78-84 us for 48 MB, and still 63 us when the image rows are contiguous (round 4, compact
// copy): the latency chain and the dead waves are the cost, not the plane-strided gathers.  Here a one-thread-per-candidate
// pass writes the inverse table (slot -> voxel row, pixel, compact row: one 16-byte entry; ~3 us), and a wave per SLOT of the padded [image][max_ne] tensors does
// the rest: row <- table, then pixel / feature row / gate / planes at once; slots past the list length write the padding
// rows (pad_queries_kernel's values), so no other launch touches the outputs.  Same values as assemble_queries2_kernel.
__global__ __launch_bounds__(256) void query_inverse_kernel(const int32_t *__restrict__ ind, const uint8_t *__restrict__ mask,
                                                            const int32_t *__restrict__ pos, const int32_t *__restrict__ grid,
                                                            const int32_t *__restrict__ pixrow, int n, int ncam, int max_ne,
                                                            int H, int W, int4 *__restrict__ inv) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * ncam) return;
  const int cam = (int)(t / n), i = (int)(t - (long long)cam * n);
  if (!mask[t]) return;
  const int slot = pos[t];
  if (slot >= max_ne) return;
  // everything the slot's wave needs to start ALL its loads at once: voxel row, pixel, rank of the pixel's compact row
  const int img = ind[(size_t)i * 4] * ncam + cam;
  const int gx = grid[(size_t)t * 2], gy = grid[(size_t)t * 2 + 1];
  const int pr = pixrow ? pixrow[(size_t)img * H * W + (size_t)gy * W + gx] : -1;
  inv[(size_t)img * max_ne + slot] = make_int4(i, gx, gy, pr);
}

__global__ __launch_bounds__(256) void assemble_queries4_kernel(AsmArgs2 a, const int32_t *__restrict__ counts,
                                                                const int4 *__restrict__ inv, int nimg) {
  const long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (w >= (long long)nimg * a.max_ne) return;
  const int img = (int)(w / a.max_ne), slot = (int)(w - (long long)img * a.max_ne);
  const size_t q = (size_t)w;
  if (slot >= counts[img]) {                       // padding row (pad_queries_kernel)
    for (int c = lane; c < a.C; c += 64) {
      a.v_feat[q * a.C + c] = 0.f;
      if (a.qpos) a.qpos[q * a.C + c] = (c & 1) ? 1.f : 0.f;
    }
    for (int c = lane; c < a.Ci; c += 64) a.v_i_feat[q * a.Ci + c] = 0.f;
    if (lane < 2) a.qgrid[q * 2 + lane] = 0.f;
    if (lane < 3) a.qpts[q * 3 + lane] = 0.f;
    return;
  }
  const int4 e = inv[q];
  const int i = e.x, gx = e.y, gy = e.z;
  const size_t hw = (size_t)a.H * a.W, pix = (size_t)gy * a.W + gx;
  // every load of the row is issued before the first store
  float vf[4], vi[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) vf[j] = (lane + 64 * j < a.C) ? a.feat[(size_t)i * a.C + lane + 64 * j] : 0.f;
  if (a.compact) {                                 // pixel-major rows written by the image projection: one contiguous row
    const float *src = a.compact + (size_t)e.w * a.Ci;
#pragma unroll
    for (int j = 0; j < 8; ++j) vi[j] = (lane + 64 * j < a.Ci) ? src[lane + 64 * j] : 0.f;
  } else {
    const float *src = (a.img ? a.img + (size_t)img * a.Ci * hw : a.img_ptrs[img]) + pix;
#pragma unroll
    for (int j = 0; j < 8; ++j) vi[j] = (lane + 64 * j < a.Ci) ? src[(size_t)(lane + 64 * j) * hw] : 0.f;
  }
  const float g = a.att ? a.att[(size_t)img * hw + pix] : 1.f;
  const float p0 = a.pinv[(size_t)i * 3], pl = lane < 3 ? a.pinv[(size_t)i * 3 + lane] : 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (lane + 64 * j < a.C) a.v_feat[q * a.C + lane + 64 * j] = vf[j];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (lane + 64 * j < a.Ci) a.v_i_feat[q * a.Ci + lane + 64 * j] = vi[j] * g;
  if (lane == 0) {
    a.qgrid[q * 2 + 0] = (float)gx / (float)a.W;
    a.qgrid[q * 2 + 1] = (float)gy / (float)a.H;
  }
  if (lane < 3) a.qpts[q * 3 + lane] = pl;
  if (a.qpos) {
    const float d = p0 / 60.f * 6.283185307179586f;
    for (int c = lane; c < a.C; c += 64) {
      float dim_t = powf(10000.f, (float)(2 * (c / 2)) / (float)a.C);
      float v = d / dim_t;
      a.qpos[q * a.C + c] = (c & 1) ? cosf(v) : sinf(v);
    }
  }
}

// zero rows of the padded query tensors: slots >= counts[image] (one wave per padding row); the position
// embedding of an all-zero query is sin(0) = 0 on even, cos(0) = 1 on odd channels
__global__ __launch_bounds__(256) void pad_queries_kernel(const int32_t *__restrict__ counts, int nimg, int max_ne, int C,
                                                          int Ci, float *__restrict__ v_feat,
                                                          float *__restrict__ v_i_feat, float *__restrict__ qgrid,
                                                          float *__restrict__ qpts, float *__restrict__ qpos) {
  long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (w >= (long long)nimg * max_ne) return;
  int img = (int)(w / max_ne), slot = (int)(w - (long long)img * max_ne);
  if (slot < counts[img]) return;
  size_t q = (size_t)w;
  for (int c = lane; c < C; c += 64) {
    v_feat[q * C + c] = 0.f;
    if (qpos) qpos[q * C + c] = (c & 1) ? 1.f : 0.f;
  }
  for (int c = lane; c < Ci; c += 64) v_i_feat[q * Ci + c] = 0.f;
  if (lane < 2) qgrid[q * 2 + lane] = 0.f;
  if (lane < 3) qpts[q * 3 + lane] = 0.f;
}

}  // namespace df3d

extern "C" int df3d_gate_scatter(const float *s9, const int32_t *indices, const int32_t *grid_xy, const uint8_t *mask,
                                 int n, int batch, int ncam, int H, int W, int32_t *winner, float *S, int clear,
                                 void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(winner && S, "gate_scatter: null output");
  size_t nimg = (size_t)batch * ncam;
  if (clear) DF3D_HIP(hipMemsetAsync(S, 0, nimg * 9 * (size_t)H * W * sizeof(float), stream));
  DF3D_HIP(hipMemsetAsync(winner, 0xff, nimg * H * W * sizeof(int32_t), stream));
  if (n == 0) return DF3D_OK;
  DF3D_CHECK_ARG(s9 && indices && grid_xy && mask, "gate_scatter: null input");
  ScatArgs a = {nullptr, nullptr, indices, grid_xy, mask, n, 0, ncam, H, W, winner, nullptr};
  dim3 g(cdiv((long long)n * ncam, 256));
  hipLaunchKernelGGL(scatter_winner_kernel, g, dim3(256), 0, stream, a);
  hipLaunchKernelGGL(gate_scatter_kernel, g, dim3(256), 0, stream, s9, indices, grid_xy, mask, winner, n, ncam, H, W, S);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_gate_scatter_rows(const float *features, int channels, const float *point_inv, const float *T,
                                      const int32_t *indices, const int32_t *grid_xy, const uint8_t *mask, int n, int batch,
                                      int ncam, int H, int W, int32_t *winner, float *S, int clear, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(winner && S, "gate_scatter_rows: null output");
  DF3D_CHECK_ARG(channels > 0 && channels % 4 == 0 && channels <= 1024, "gate_scatter_rows: %d channels", channels);
  size_t nimg = (size_t)batch * ncam;
  DF3D_HIP(hipMemsetAsync(winner, 0xff, nimg * H * W * sizeof(int32_t), stream));
  if (n == 0) {
    if (clear) DF3D_HIP(hipMemsetAsync(S, 0, nimg * 9 * (size_t)H * W * sizeof(float), stream));
    return DF3D_OK;
  }
  DF3D_CHECK_ARG(features && point_inv && T && indices && grid_xy && mask, "gate_scatter_rows: null input");
  DF3D_CHECK_ARG((long long)H * W < 0x7fffffffLL, "gate_scatter_rows: map size");
  ScatArgs a = {nullptr, nullptr, indices, grid_xy, mask, n, 0, ncam, H, W, winner, nullptr};
  hipLaunchKernelGGL(scatter_winner_kernel, dim3(cdiv((long long)n * ncam, 256)), dim3(256), 0, stream, a);
  const long long npix = (long long)nimg * H * W;
  hipLaunchKernelGGL(gate_scatter_rows_kernel, dim3(cdiv(npix, 256)), dim3(256), (size_t)(channels + 3) * 9 * sizeof(float),
                     stream, features, channels, point_inv, T, winner, npix, H * W, clear, S);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

// df3d_gate_scatter_rows for a winner map the caller already holds (it depends on the voxel coordinates alone: the frame-head
// worker builds it a frame ahead): the rows kernel only
extern "C" int df3d_gate_rows(const float *features, int channels, const float *point_inv, const float *T, const int32_t *winner,
                              int nimg, int H, int W, float *S, int clear, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(features && point_inv && T && winner && S, "gate_rows: null argument");
  DF3D_CHECK_ARG(channels > 0 && channels % 4 == 0 && channels <= 1024, "gate_rows: %d channels", channels);
  DF3D_CHECK_ARG((long long)H * W < 0x7fffffffLL, "gate_rows: map size");
  const long long npix = (long long)nimg * H * W;
  if (npix == 0) return DF3D_OK;
  hipLaunchKernelGGL(gate_scatter_rows_kernel, dim3(cdiv(npix, 256)), dim3(256), (size_t)(channels + 3) * 9 * sizeof(float),
                     stream, features, channels, point_inv, T, winner, npix, H * W, clear, S);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_gate_finish_bias(const float *gate, const float *gate_bias, const float *S, const float *kg, int nimg, int H,
                                     int W, float *att, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(gate && gate_bias && S && kg && att, "gate_finish_bias: null argument");
  long long tot = (long long)nimg * H * W;
  if (tot == 0) return DF3D_OK;
  hipLaunchKernelGGL(gate_finish_kernel<true>, dim3(cdiv(tot, 256)), dim3(256), 0, stream, gate, gate_bias, S, kg, nimg, H, W, att);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_fusion_writeback_split(const float *features, const float *enh, const int32_t *indices, const uint8_t *mask,
                                           const int32_t *pos, int n, int channels, int ncam, int max_ne, float *out,
                                           void *out_split, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(out && channels > 0 && channels % 8 == 0, "fusion_writeback_split: channels must be a multiple of 8");
  if (n == 0) return DF3D_OK;
  DF3D_CHECK_ARG(features && enh && indices && mask && pos, "fusion_writeback_split: null input");
  hipLaunchKernelGGL(writeback8_kernel, dim3(cdiv((long long)n * (channels / 8), 256)), dim3(256), 0, stream, features, enh,
                     indices, mask, pos, n, channels, ncam, max_ne, out, (u32x4 *)out_split);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_gate_finish(const float *gate, const float *S, const float *kg, int nimg, int H, int W, float *att,
                                void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(gate && S && kg && att, "gate_finish: null argument");
  long long tot = (long long)nimg * H * W;
  if (tot == 0) return DF3D_OK;
  hipLaunchKernelGGL(gate_finish_kernel<false>, dim3(cdiv(tot, 256)), dim3(256), 0, stream, gate, (const float *)nullptr, S, kg,
                     nimg, H, W, att);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

static int assemble_queries2_impl(const float *features, const float *point_inv, const int32_t *indices,
                                  const int32_t *grid_xy, const uint8_t *mask, const int32_t *pos,
                                  const float *img_feats, const float *const *img_ptrs, const float *att, int n,
                                  int channels, int img_channels, int batch, int ncam, int H, int W, int max_ne,
                                  float *v_feat, float *v_i_feat, float *qgrid, float *qpts, float *qpos,
                                  const int32_t *counts, const int32_t *pixrow, const float *compact, int32_t *inv,
                                  void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(v_feat && v_i_feat && qgrid && qpts, "assemble_queries2: null output");
  size_t nq = (size_t)batch * ncam * max_ne;
  if (qpos) DF3D_CHECK_ARG(channels % 2 == 0, "assemble_queries2: odd channel count");
  if (inv && counts && nq && n && channels <= 256 && img_channels <= 512) {
    DF3D_CHECK_ARG(features && point_inv && indices && grid_xy && mask && pos && (img_feats || img_ptrs || (compact && pixrow)),
                   "assemble_queries2: null input");
    AsmArgs2 a = {features, point_inv, indices, grid_xy, mask, pos, img_feats, img_ptrs, att, n, channels, img_channels, ncam, H, W,
                  max_ne, v_feat, v_i_feat, qgrid, qpts, qpos, pixrow, compact};
    hipLaunchKernelGGL(query_inverse_kernel, dim3(cdiv((long long)n * ncam, 256)), dim3(256), 0, stream, indices, mask, pos,
                       grid_xy, compact ? pixrow : nullptr, n, ncam, max_ne, H, W, (int4 *)inv);
    hipLaunchKernelGGL(assemble_queries4_kernel, dim3(cdiv((long long)nq * 64, 256)), dim3(256), 0, stream, a, counts,
                       (const int4 *)inv, batch * ncam);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
  }
  if (counts && nq) {
    // only the padding rows (slot >= list length) are cleared; every other row is written by the gather below
    hipLaunchKernelGGL(pad_queries_kernel, dim3(cdiv((long long)nq * 64, 256)), dim3(256), 0, stream, counts,
                       batch * ncam, max_ne, channels, img_channels, v_feat, v_i_feat, qgrid, qpts, qpos);
  } else {
    DF3D_HIP(hipMemsetAsync(v_feat, 0, nq * channels * sizeof(float), stream));
    DF3D_HIP(hipMemsetAsync(v_i_feat, 0, nq * img_channels * sizeof(float), stream));
    DF3D_HIP(hipMemsetAsync(qgrid, 0, nq * 2 * sizeof(float), stream));
    DF3D_HIP(hipMemsetAsync(qpts, 0, nq * 3 * sizeof(float), stream));
    if (qpos && nq) {
      size_t tot = nq * channels;
      hipLaunchKernelGGL(qpos_pad_kernel, dim3(cdiv((long long)tot, 256)), dim3(256), 0, stream, qpos, tot);
    }
  }
  if (n == 0 || max_ne == 0) return DF3D_OK;
  DF3D_CHECK_ARG(features && point_inv && indices && grid_xy && mask && pos && (img_feats || img_ptrs || compact),
                 "assemble_queries2: null input");
  AsmArgs2 a = {features, point_inv, indices, grid_xy, mask, pos, img_feats, img_ptrs, att, n, channels, img_channels, ncam, H, W,
                max_ne, v_feat, v_i_feat, qgrid, qpts, qpos, pixrow, compact};
  if (compact) {                       // contiguous 1 KB rows: the wave-per-query kernel reads them at full width
    hipLaunchKernelGGL(assemble_queries2_kernel, dim3(cdiv((long long)n * ncam * 64, 256)), dim3(256), 0, stream, a);
    DF3D_LAUNCH_CHECK();
    return DF3D_OK;
  }
  // measured on MI355X: lanes over queries 265 -> ~150 us at four samples x six cameras (TransFusion tree, step 8.02 -> 7.91 ms),
  // but 96-100 us against 82-85 us of the wave-per-query kernel at one sample (too few workgroups): chosen by image count;
  // DF3D_ASSEMBLE=1 / 2 forces one or the other
  static const int forced = getenv("DF3D_ASSEMBLE") ? atoi(getenv("DF3D_ASSEMBLE")) : 0;
  const bool by_slot = forced == 1 || (forced != 2 && batch * ncam >= 12);
  if (counts && by_slot)      // lanes over queries (needs the list lengths)
    hipLaunchKernelGGL(assemble_queries3_kernel, dim3(cdiv(max_ne, 64), batch * ncam, cdiv(img_channels, ASM3_CH)), dim3(256), 0, stream, a, counts);
  else
    hipLaunchKernelGGL(assemble_queries2_kernel, dim3(cdiv((long long)n * ncam * 64, 256)), dim3(256), 0, stream, a);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_assemble_queries2(const float *features, const float *point_inv, const int32_t *indices,
                                      const int32_t *grid_xy, const uint8_t *mask, const int32_t *pos,
                                      const float *img_feats, const float *const *img_ptrs, const float *att, int n,
                                      int channels, int img_channels, int batch, int ncam, int H, int W, int max_ne,
                                      float *v_feat, float *v_i_feat, float *qgrid, float *qpts, float *qpos,
                                      const int32_t *counts, void *stream_) {
  return assemble_queries2_impl(features, point_inv, indices, grid_xy, mask, pos, img_feats, img_ptrs, att, n, channels,
                                img_channels, batch, ncam, H, W, max_ne, v_feat, v_i_feat, qgrid, qpts, qpos, counts, nullptr,
                                nullptr, nullptr, stream_);
}

// the same BY SLOT (round 4): slot_rows [batch * ncam * max_ne][4] i32 is scratch for the slot table
// (voxel row, pixel x, pixel y, rank of the pixel's compact row)
extern "C" int df3d_assemble_queries2_slots(const float *features, const float *point_inv, const int32_t *indices,
                                            const int32_t *grid_xy, const uint8_t *mask, const int32_t *pos,
                                            const float *img_feats, const float *const *img_ptrs, const float *att, int n,
                                            int channels, int img_channels, int batch, int ncam, int H, int W, int max_ne,
                                            float *v_feat, float *v_i_feat, float *qgrid, float *qpts, float *qpos,
                                            const int32_t *counts, int32_t *slot_rows, const int32_t *pixrow,
                                            const float *compact, void *stream_) {
  DF3D_CHECK_ARG(counts && slot_rows, "assemble_queries2_slots: the list lengths and the table scratch are required");
  DF3D_CHECK_ARG((pixrow != nullptr) == (compact != nullptr), "assemble_queries2_slots: pixrow and compact come together");
  return assemble_queries2_impl(features, point_inv, indices, grid_xy, mask, pos, img_feats, img_ptrs, att, n, channels,
                                img_channels, batch, ncam, H, W, max_ne, v_feat, v_i_feat, qgrid, qpts, qpos, counts, pixrow,
                                compact, slot_rows, stream_);
}

// the same with the image features read from pixel-major rows (df3d_query_pixel_rows + df3d_imgproj_split_compact)
extern "C" int df3d_assemble_queries2_compact(const float *features, const float *point_inv, const int32_t *indices,
                                              const int32_t *grid_xy, const uint8_t *mask, const int32_t *pos,
                                              const int32_t *pixrow, const float *compact, const float *att, int n, int channels,
                                              int img_channels, int batch, int ncam, int H, int W, int max_ne, float *v_feat,
                                              float *v_i_feat, float *qgrid, float *qpts, float *qpos, const int32_t *counts,
                                              void *stream_) {
  DF3D_CHECK_ARG(pixrow && compact, "assemble_queries2_compact: null pixel rows");
  return assemble_queries2_impl(features, point_inv, indices, grid_xy, mask, pos, nullptr, nullptr, att, n, channels,
                                img_channels, batch, ncam, H, W, max_ne, v_feat, v_i_feat, qgrid, qpts, qpos, counts, pixrow,
                                compact, nullptr, stream_);
}

static size_t df3d_query_pixel_rows_workspace_bytes_impl(long long npix) {
  return align_up((size_t)npix * 4, 256) * 2 + scan_scratch_bytes((size_t)npix) + 256;
}

extern "C" size_t df3d_query_pixel_rows_workspace_bytes(int batch, int ncam, int H, int W) {
  return df3d_query_pixel_rows_workspace_bytes_impl((long long)batch * ncam * H * W);
}

// pixrow[img * H * W + pixel] = rank of the pixel among the pixels some query of image `img` samples (image-major, row-major
// order), -1 elsewhere; *total (device) = their number.  Depends on the projection alone: the frame head runs it.
extern "C" int df3d_query_pixel_rows(const int32_t *indices, const int32_t *grid_xy, const uint8_t *mask, int n, int batch,
                                     int ncam, int H, int W, int32_t *pixrow, int32_t *total, void *workspace,
                                     size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(pixrow && total && workspace && batch > 0 && ncam > 0 && H > 0 && W > 0, "query_pixel_rows: bad arguments");
  const long long npix = (long long)batch * ncam * H * W;
  DF3D_CHECK_ARG(workspace_bytes >= df3d_query_pixel_rows_workspace_bytes_impl(npix), "query_pixel_rows: workspace too small");
  char *ws = (char *)workspace;
  uint32_t *flag = (uint32_t *)ws;
  uint32_t *rank = (uint32_t *)(ws + align_up((size_t)npix * 4, 256));
  void *scratch = ws + 2 * align_up((size_t)npix * 4, 256);
  DF3D_HIP(hipMemsetAsync(flag, 0, (size_t)npix * 4, stream));
  if (n > 0) {
    DF3D_CHECK_ARG(indices && grid_xy && mask, "query_pixel_rows: null input");
    hipLaunchKernelGGL(mark_query_pixels_kernel, dim3(cdiv((long long)n * ncam, 256)), dim3(256), 0, stream, indices, grid_xy, mask,
                       n, ncam, H, W, flag);
  }
  int rc = exclusive_scan_u32(flag, rank, (size_t)npix, (uint32_t *)total, scratch, scan_scratch_bytes((size_t)npix), stream);
  if (rc) return rc;
  hipLaunchKernelGGL(pixrow_kernel, dim3(cdiv(npix, 256)), dim3(256), 0, stream, flag, rank, (size_t)npix, pixrow);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_query_slots(const uint8_t *mask, const int32_t *indices, int n, int batch, int ncam, int32_t *pos,
                                int32_t *counts, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(counts && batch > 0 && ncam > 0, "query_slots: bad arguments");
  if (n == 0) {
    DF3D_HIP(hipMemsetAsync(counts, 0, (size_t)batch * ncam * sizeof(int32_t), stream));
    return DF3D_OK;
  }
  DF3D_CHECK_ARG(mask && indices && pos, "query_slots: null argument");
  hipLaunchKernelGGL(query_slots_kernel, dim3(ncam, batch), dim3(1024), 0, stream, mask, indices, n, ncam, pos, counts);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

// ---- weight gradient of a one-output 1 x 1 convolution over channel-first maps (the image gate's `reduced_dim3`, training) ----
// out[n][c] = sum_s x[n][c][s] * g[n][s]: one workgroup per (map, channel) row, 16-byte loads, fp32 partials, a tree at the end.
// (The library's batched matrix-vector product reads the 246 MB of camera maps at 0.5 TB/s.)
namespace df3d {
__global__ __launch_bounds__(256) void chanfirst_dot_kernel(const float *__restrict__ x, const float *__restrict__ g, int C,
                                                            long long S, float *__restrict__ out) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int row = blockIdx.x, n = row / C;
  const float *xr = x + (size_t)row * S, *gr = g + (size_t)n * S;
  float acc = 0.f;
  const bool vec = (S % 4 == 0) && (((size_t)xr | (size_t)gr) % 16 == 0);
  if (vec) {
    for (long long i = threadIdx.x; i < S / 4; i += 256) {
      const f4 a = ((const f4 *)xr)[i], b = ((const f4 *)gr)[i];
      acc += (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
    }
  } else {
    for (long long i = threadIdx.x; i < S; i += 256) acc += xr[i] * gr[i];
  }
  __shared__ float red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[row] = red[0];
}
}  // namespace df3d

extern "C" int df3d_chanfirst_dot(const float *x, const float *g, int nmaps, int channels, long long S, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(nmaps >= 0 && channels > 0 && S >= 0, "chanfirst_dot: bad sizes");
  if (nmaps == 0) return DF3D_OK;
  DF3D_CHECK_ARG(x && g && out, "chanfirst_dot: null argument");
  hipLaunchKernelGGL(df3d::chanfirst_dot_kernel, dim3(nmaps * channels), dim3(256), 0, stream, x, g, channels, S, out);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}
