// Point fusion of the Voxel-RCNN tree as ONE kernel (round 3): voxel -> LiDAR point the camera saw -> pixel -> image
// feature at that pixel (-> + voxel feature | -> padded query tensors).
//
// Reference: VR/pcdet/models/backbones_3d/spconv_backbone.py:682-756 (`point_fusion`: voxel centres, the inverse of the
// recorded 3-D augmentations :701-714, `calib.lidar_to_img`, `torch.Tensor(voxels_2d).long()`, the feature map upsampled
// to the image size with `F.interpolate(mode='bilinear')` and indexed at the truncated pixels, MVX sum :746-748) and
// :760-814 (the same sampling at stride 8 for the ACTR queries' image features + the normalised pixel grid).
// Rounds 1-2 composed this from ~55 torch launches per call (host-bound: ~0.6 ms of launch latency at 150 k voxels) and a
// channels-last COPY of the whole feature map per call (240 MB, 0.32 ms) to make the taps row gathers.
//
// Every floating-point operation below is the one the torch composition performs, in its order, with FMA contraction off:
// a 1-ulp change in a pixel coordinate flips its truncation and with it the sampled feature.  The bilinear weights are
// torch's `upsample_bilinear2d` (align_corners = False) for an INTEGER destination pixel:
//   src = max((dst + 0.5) * (in / out) - 0.5, 0), taps floor(src) and its right / lower neighbour (if any).
// A voxel is served by C / 4 adjacent lanes (4 channels each, read from the NCHW map where the camera network left it).
#include "common.h"

namespace df3d {

typedef float mv_f32x4 __attribute__((ext_vector_type(4)));

struct MvxArgs {
  const int32_t *ind;       // [n, 4] (b, z, y, x)
  int n, B;
  float stride;             // voxel stride of this level
  float vz, vy, vx, rz, ry, rx;   // voxel size and range minimum, per axis
  const float *aug;         // [B][5]: global scale, cos(-rot), sin(-rot), sign of flip_x (on y), sign of flip_y (on x)
  const float *l2i;         // [B][12] lidar2img rows
  const float *fmap;        // [B, C, Hin, Win]
  int C, Hin, Win, h, w;
  float sy, sx;             // float32(Hin) / float32(h), float32(Win) / float32(w)
  const float *add;         // [n, C] or null: out = add + feature
  const long long *rows;    // [n] output row of voxel i, or null (= i)
  float *out;               // [rows, C]
  float *uv;                // [n, 2] pixel coordinates (float) or null
  float *grid;              // [rows, 2] = (u / w, v / h) or null
};

struct MvxTap {
  int i0, i1;
  float lam;
};

__device__ __forceinline__ MvxTap mvx_tap(long long px, int in_size, int out_size, float scale) {
#pragma clang fp contract(off)
  long long d = px < 0 ? 0 : (px > out_size - 1 ? out_size - 1 : px);
  float src = ((float)d + 0.5f) * scale;
  src = src - 0.5f;
  src = src < 0.f ? 0.f : src;
  long long i0 = (long long)floorf(src);
  if (i0 > in_size - 1) i0 = in_size - 1;
  MvxTap t;
  t.i0 = (int)i0;
  t.i1 = (int)i0 + (i0 < in_size - 1 ? 1 : 0);
  t.lam = src - (float)i0;
  return t;
}

__global__ __launch_bounds__(256) void mvx_sample_kernel(MvxArgs a) {
#pragma clang fp contract(off)
  const int lpv = a.C / 4;                                   // lanes per voxel
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long i = t / lpv;
  if (i >= a.n) return;
  const int c4 = (int)(t - i * lpv) * 4;
  const int32_t *p = a.ind + (size_t)i * 4;
  const int b = p[0];
  // voxel corner -> LiDAR: ((index * stride) * voxel size) + range minimum
  float z = (float)p[1] * a.stride;
  z = z * a.vz;
  z = z + a.rz;
  float y = (float)p[2] * a.stride;
  y = y * a.vy;
  y = y + a.ry;
  float x = (float)p[3] * a.stride;
  x = x * a.vx;
  x = x + a.rx;
  // the point cloud the camera saw: scale, rotation about z, flips (identity values leave the bits unchanged)
  const float *g = a.aug + (size_t)b * 5;
  x = x / g[0];
  y = y / g[0];
  z = z / g[0];
  {
    const float xc = x * g[1], ys = y * g[2], xs = x * g[2], yc = y * g[1];
    x = xc - ys;
    y = xs + yc;
  }
  y = y * g[3];
  x = x * g[4];
  const float *P = a.l2i + (size_t)b * 12;
  float hh[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float acc = P[r * 4] * x;
    acc = acc + P[r * 4 + 1] * y;
    acc = acc + P[r * 4 + 2] * z;
    hh[r] = acc + P[r * 4 + 3];
  }
  const float u = hh[0] / hh[2], v = hh[1] / hh[2];
  const size_t orow = a.rows ? (size_t)a.rows[i] : (size_t)i;
  if (c4 == 0) {
    if (a.uv) {
      a.uv[(size_t)i * 2] = u;
      a.uv[(size_t)i * 2 + 1] = v;
    }
    if (a.grid) {
      a.grid[orow * 2] = u / (float)a.w;
      a.grid[orow * 2 + 1] = v / (float)a.h;
    }
  }
  // torch's .long(): truncation; NaN / inf / out-of-range convert to the minimum integer there, i.e. "outside"
  const bool fin = isfinite(u) && isfinite(v) && fabsf(u) < 9.0e18f && fabsf(v) < 9.0e18f;
  const long long pu = fin ? (long long)u : -1, pv = fin ? (long long)v : -1;
  const bool ok = pv >= 0 && pv < a.h && pu >= 0 && pu < a.w;
  mv_f32x4 f = (mv_f32x4){0.f, 0.f, 0.f, 0.f};
  if (ok) {
    const MvxTap ty = mvx_tap(pv, a.Hin, a.h, a.sy), tx = mvx_tap(pu, a.Win, a.w, a.sx);
    const size_t plane = (size_t)a.Hin * a.Win;
    const float *m = a.fmap + ((size_t)b * a.C + c4) * plane;
    const float ax = 1.f - tx.lam, ay = 1.f - ty.lam;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float *mc = m + c * plane;
      const float t00 = mc[(size_t)ty.i0 * a.Win + tx.i0], t01 = mc[(size_t)ty.i0 * a.Win + tx.i1];
      const float t10 = mc[(size_t)ty.i1 * a.Win + tx.i0], t11 = mc[(size_t)ty.i1 * a.Win + tx.i1];
      float top = ax * t00;
      top = top + tx.lam * t01;
      float bot = ax * t10;
      bot = bot + tx.lam * t11;
      float r = ay * top;
      f[c] = r + ty.lam * bot;
    }
  }
  if (a.add) f = *(const mv_f32x4 *)(a.add + (size_t)i * a.C + c4) + f;
  *(mv_f32x4 *)(a.out + orow * a.C + c4) = f;
}

}  // namespace df3d

using namespace df3d;

extern "C" int df3d_voxel_image_sample(const int32_t *indices, int n, int batch, float voxel_stride, const float *voxel_size_zyx,
                                       const float *range_min_zyx, const float *aug, const float *lidar2img, const float *fmap,
                                       int C, int Hin, int Win, int img_h, int img_w, float scale_y, float scale_x,
                                       const float *add, const long long *out_rows, float *out, float *uv, float *grid,
                                       void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(indices && voxel_size_zyx && range_min_zyx && aug && lidar2img && fmap && out, "voxel_image_sample: null argument");
  DF3D_CHECK_ARG(n >= 0 && batch > 0 && C > 0 && C % 4 == 0 && C <= 1024 && Hin > 0 && Win > 0 && img_h > 0 && img_w > 0,
                 "voxel_image_sample: bad sizes (n %d, C %d, map %d x %d, image %d x %d)", n, C, Hin, Win, img_h, img_w);
  if (n == 0) return DF3D_OK;
  MvxArgs a;
  a.ind = indices;
  a.n = n;
  a.B = batch;
  a.stride = voxel_stride;
  a.vz = voxel_size_zyx[0], a.vy = voxel_size_zyx[1], a.vx = voxel_size_zyx[2];
  a.rz = range_min_zyx[0], a.ry = range_min_zyx[1], a.rx = range_min_zyx[2];
  a.aug = aug;
  a.l2i = lidar2img;
  a.fmap = fmap;
  a.C = C, a.Hin = Hin, a.Win = Win, a.h = img_h, a.w = img_w;
  a.sy = scale_y, a.sx = scale_x;
  a.add = add;
  a.rows = out_rows;
  a.out = out;
  a.uv = uv;
  a.grid = grid;
  const long long threads = (long long)n * (C / 4);
  hipLaunchKernelGGL(mvx_sample_kernel, dim3((unsigned)cdiv(threads, 256)), dim3(256), 0, stream, a);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}
