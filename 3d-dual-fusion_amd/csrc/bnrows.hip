// BatchNorm with batch statistics over channels-last rows [N, C] for gfx950 (training rows).
//
// Reference: every conv of the sparse backbone, the BEV neck and the head is followed by BatchNorm (+ ReLU)
// (CP/det3d/models/backbones/scn.py:51-118, necks/rpn.py:22-163, bbox_heads/center_head.py:66-110); in training the
// reference runs cuDNN's spatial BatchNorm on NCHW volumes / BatchNorm1d on the [N, C] feature rows.  torch's own kernels for
// the [N, C] layout take ~55 us per call on rows that are 10 MB (a 5 us read): 4 ms of a 26 ms training step.
//
// Layout: thread = (4 consecutive columns, row lane); a workgroup owns (row slice, block of <= 256 columns), so every load is
// 16 bytes and a wave covers whole rows.  Forward = column sums of x and x^2 (fp32 per thread over a few hundred rows, then
// double atomics) -> every workgroup of the second kernel derives scale / shift of its columns from the sums, writes
// y = x * scale + shift (optionally ReLU); workgroup 0 keeps (mean, rstd, scale, shift) for backward and updates the running
// statistics the way nn.BatchNorm does (unbiased variance, momentum).  Backward = column sums of g and g * xhat (g = dy under
// the ReLU mask, which is recomputed from x with the very scale / shift of the forward) -> dx = scale * (g - mean(g) - xhat *
// mean(g xhat)), dweight = sum g xhat, dbias = sum g.
#include "common.h"

namespace df3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Double atomics from hundreds of workgroups onto the SAME 2 x C addresses serialise (~100 ns each: 43 us of a kernel whose
// reads take 5): the sums live in BN_REPL replicas [replica][2][C], workgroup i adds into replica i % BN_REPL, readers add
// the replicas up.
constexpr int BN_REPL = 16;

struct BnArgs {
  const float *x, *dy, *weight, *bias;
  float *y, *dx, *saved;        // saved [4][C]: mean, rstd, scale, shift
  double *sums;                 // [BN_REPL][2][C]
  float *running_mean, *running_var, *dweight, *dbias;
  long long n;
  int c, cb, qb, rows_per_wg, rows_per_wg_red, relu;
  float eps, momentum;
};

__device__ __forceinline__ void bn_block_reduce(f32x4 a, f32x4 b, int qb, double *dst, int col0, int c) {
  __shared__ float red[256 * 8];
  const int tid = threadIdx.x;
  *(f32x4 *)&red[tid * 8] = a;
  *(f32x4 *)&red[tid * 8 + 4] = b;
  __syncthreads();
  if (tid < qb) {                                    // one thread per column quad sums its row lanes
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = tid; r < 256; r += qb)
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += (double)red[r * 8 + e];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsafeAtomicAdd(dst + col0 + tid * 4 + e, s[e]);
      unsafeAtomicAdd(dst + c + col0 + tid * 4 + e, s[4 + e]);
    }
  }
}

__global__ __launch_bounds__(256) void bn_rows_stats_kernel(BnArgs a) {
  const int tid = threadIdx.x, quad = tid % a.qb, rl = tid / a.qb, rp = 256 / a.qb;
  const int col0 = blockIdx.y * a.cb;
  const long long r0 = (long long)blockIdx.x * a.rows_per_wg_red, r1 = min(r0 + a.rows_per_wg_red, a.n);
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, ss = {0.f, 0.f, 0.f, 0.f};
  const float *p = a.x + col0 + quad * 4;
  long long r = r0 + rl;
  for (; r + 3LL * rp < r1; r += 4LL * rp) {          // four independent loads in flight per thread
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *(const f32x4 *)(p + (r + (long long)u * rp) * a.c);
#pragma unroll
    for (int u = 0; u < 4; ++u) s += v[u], ss += v[u] * v[u];
  }
  for (; r < r1; r += rp) {
    const f32x4 v = *(const f32x4 *)(p + r * a.c);
    s += v;
    ss += v * v;
  }
  bn_block_reduce(s, ss, a.qb, a.sums + (size_t)(blockIdx.x % BN_REPL) * 2 * a.c, col0, a.c);
}

// Column sums of this thread's four columns: row lane 0 adds the replicas up, the other row lanes of the workgroup read the
// result from LDS (every thread adding 16 replicas itself doubled the apply kernels' time).
__device__ __forceinline__ void bn_column_sums(const BnArgs &a, int col, int quad, int rl, double (&s1)[4], double (&s2)[4]) {
  __shared__ double tot[64 * 8];
  if (rl == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      double x1 = 0.0, x2 = 0.0;
      for (int r = 0; r < BN_REPL; ++r) {
        x1 += a.sums[(size_t)r * 2 * a.c + col + e];
        x2 += a.sums[(size_t)r * 2 * a.c + a.c + col + e];
      }
      tot[quad * 8 + e] = x1, tot[quad * 8 + 4 + e] = x2;
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; ++e) s1[e] = tot[quad * 8 + e], s2[e] = tot[quad * 8 + 4 + e];
}

// scale / shift of this thread's four columns from the column sums (the same arithmetic in every workgroup and in backward)
__device__ __forceinline__ void bn_columns(const BnArgs &a, int col, int quad, int rl, f32x4 &mean, f32x4 &rstd, f32x4 &scale,
                                           f32x4 &shift, f32x4 &var_unbiased) {
  double s1[4], s2[4];
  bn_column_sums(a, col, quad, rl, s1, s2);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const double m = s1[e] / (double)a.n;
    double v = s2[e] / (double)a.n - m * m;
    v = v > 0.0 ? v : 0.0;
    mean[e] = (float)m;
    rstd[e] = (float)(1.0 / sqrt(v + (double)a.eps));
    var_unbiased[e] = (float)(a.n > 1 ? v * (double)a.n / (double)(a.n - 1) : v);
    const float w = a.weight ? a.weight[col + e] : 1.f, b = a.bias ? a.bias[col + e] : 0.f;
    scale[e] = w * rstd[e];
    shift[e] = b - mean[e] * scale[e];
  }
}

__global__ __launch_bounds__(256) void bn_rows_apply_kernel(BnArgs a) {
  const int tid = threadIdx.x, quad = tid % a.qb, rl = tid / a.qb, rp = 256 / a.qb;
  const int col = blockIdx.y * a.cb + quad * 4;
  f32x4 mean, rstd, scale, shift, varu;
  bn_columns(a, col, quad, rl, mean, rstd, scale, shift, varu);
  if (blockIdx.x == 0 && rl == 0) {
    *(f32x4 *)(a.saved + col) = mean;
    *(f32x4 *)(a.saved + a.c + col) = rstd;
    *(f32x4 *)(a.saved + 2 * a.c + col) = scale;
    *(f32x4 *)(a.saved + 3 * a.c + col) = shift;
    if (a.running_mean) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a.running_mean[col + e] = (1.f - a.momentum) * a.running_mean[col + e] + a.momentum * mean[e];
        a.running_var[col + e] = (1.f - a.momentum) * a.running_var[col + e] + a.momentum * varu[e];
      }
    }
  }
  const long long r0 = (long long)blockIdx.x * a.rows_per_wg, r1 = min(r0 + a.rows_per_wg, a.n);
  for (long long r = r0 + rl; r < r1; r += rp) {
    const f32x4 v = *(const f32x4 *)(a.x + r * a.c + col);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = fmaf(v[e], scale[e], shift[e]);
      if (a.relu) o[e] = fmaxf(o[e], 0.f);
    }
    *(f32x4 *)(a.y + r * a.c + col) = o;
  }
}

__global__ __launch_bounds__(256) void bn_rows_bwd_reduce_kernel(BnArgs a) {
  const int tid = threadIdx.x, quad = tid % a.qb, rl = tid / a.qb, rp = 256 / a.qb;
  const int col0 = blockIdx.y * a.cb, col = col0 + quad * 4;
  const f32x4 mean = *(const f32x4 *)(a.saved + col), rstd = *(const f32x4 *)(a.saved + a.c + col);
  const f32x4 scale = *(const f32x4 *)(a.saved + 2 * a.c + col), shift = *(const f32x4 *)(a.saved + 3 * a.c + col);
  const long long r0 = (long long)blockIdx.x * a.rows_per_wg_red, r1 = min(r0 + a.rows_per_wg_red, a.n);
  f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sgx = {0.f, 0.f, 0.f, 0.f};
  auto take = [&](const f32x4 &v, f32x4 g) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (a.relu && !(fmaf(v[e], scale[e], shift[e]) > 0.f)) g[e] = 0.f;
      sg[e] += g[e];
      sgx[e] += g[e] * ((v[e] - mean[e]) * rstd[e]);
    }
  };
  long long r = r0 + rl;
  for (; r + (long long)rp < r1; r += 2LL * rp) {       // two rows (four loads) in flight per thread
    const f32x4 v0 = *(const f32x4 *)(a.x + r * a.c + col), g0 = *(const f32x4 *)(a.dy + r * a.c + col);
    const f32x4 v1 = *(const f32x4 *)(a.x + (r + rp) * a.c + col), g1 = *(const f32x4 *)(a.dy + (r + rp) * a.c + col);
    take(v0, g0);
    take(v1, g1);
  }
  for (; r < r1; r += rp) take(*(const f32x4 *)(a.x + r * a.c + col), *(const f32x4 *)(a.dy + r * a.c + col));
  bn_block_reduce(sg, sgx, a.qb, a.sums + (size_t)(blockIdx.x % BN_REPL) * 2 * a.c, col0, a.c);
}

__global__ __launch_bounds__(256) void bn_rows_bwd_apply_kernel(BnArgs a) {
  const int tid = threadIdx.x, quad = tid % a.qb, rl = tid / a.qb, rp = 256 / a.qb;
  const int col = blockIdx.y * a.cb + quad * 4;
  const f32x4 mean = *(const f32x4 *)(a.saved + col), rstd = *(const f32x4 *)(a.saved + a.c + col);
  const f32x4 scale = *(const f32x4 *)(a.saved + 2 * a.c + col), shift = *(const f32x4 *)(a.saved + 3 * a.c + col);
  f32x4 mg, mgx;
  double s1[4], s2[4];
  bn_column_sums(a, col, quad, rl, s1, s2);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    mg[e] = (float)(s1[e] / (double)a.n);
    mgx[e] = (float)(s2[e] / (double)a.n);
  }
  if (blockIdx.x == 0 && rl == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (a.dbias) a.dbias[col + e] = (float)s1[e];
      if (a.dweight) a.dweight[col + e] = (float)s2[e];
    }
  }
  const long long r0 = (long long)blockIdx.x * a.rows_per_wg, r1 = min(r0 + a.rows_per_wg, a.n);
  for (long long r = r0 + rl; r < r1; r += rp) {
    const f32x4 v = *(const f32x4 *)(a.x + r * a.c + col);
    f32x4 g = *(const f32x4 *)(a.dy + r * a.c + col), o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (a.relu && !(fmaf(v[e], scale[e], shift[e]) > 0.f)) g[e] = 0.f;
      o[e] = scale[e] * (g[e] - mg[e] - (v[e] - mean[e]) * rstd[e] * mgx[e]);
    }
    *(f32x4 *)(a.dx + r * a.c + col) = o;
  }
}

static bool bn_shape(BnArgs &a, long long n, int c) {
  a.n = n, a.c = c;
  a.cb = c < 256 ? c : 256;
  if (c % 4 != 0 || c % a.cb != 0) return false;
  a.qb = a.cb / 4;
  if (256 % a.qb != 0) return false;
  const int rp = 256 / a.qb, ncb = c / a.cb;
  // ~2048 workgroups, at least 8 passes of the row lanes each
  long long rows = cdiv(n, (long long)std::max(1, 2048 / ncb));
  rows = std::max<long long>(rows, 8LL * rp);
  a.rows_per_wg = (int)(cdiv(rows, (long long)rp) * rp);
  // the two reductions end in double atomics on the SAME 2 x C addresses from every workgroup: ~512 workgroups (each
  // keeps four loads per thread in flight), not 2048, or the atomics serialise into most of the kernel's time
  rows = std::max<long long>(cdiv(n, (long long)std::max(1, 512 / ncb)), 8LL * rp);
  a.rows_per_wg_red = (int)(cdiv(rows, (long long)rp) * rp);
  return true;
}

}  // namespace df3d

using namespace df3d;

extern "C" int df3d_bn_rows_supported(int c) {
  BnArgs a{};
  return bn_shape(a, 1024, c) ? 1 : 0;
}

extern "C" int df3d_bn_rows_forward(const float *x, long long n, int c, const float *weight, const float *bias, float eps,
                                    float momentum, int relu, float *running_mean, float *running_var, double *sums,
                                    float *saved, float *y, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BnArgs a{};
  DF3D_CHECK_ARG(n >= 1 && c >= 4 && bn_shape(a, n, c), "bn_rows_forward: %lld rows x %d channels has no row kernel", n, c);
  DF3D_CHECK_ARG(x && sums && saved && y, "bn_rows_forward: null argument");
  DF3D_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "bn_rows_forward: running statistics come as a pair");
  a.x = x, a.weight = weight, a.bias = bias, a.y = y, a.saved = saved, a.sums = sums;
  a.running_mean = running_mean, a.running_var = running_var, a.relu = relu, a.eps = eps, a.momentum = momentum;
  DF3D_HIP(hipMemsetAsync(sums, 0, (size_t)BN_REPL * 2 * c * sizeof(double), stream));
  const dim3 grid((unsigned)cdiv(n, (long long)a.rows_per_wg), c / a.cb);
  const dim3 grid_red((unsigned)cdiv(n, (long long)a.rows_per_wg_red), c / a.cb);
  hipLaunchKernelGGL(bn_rows_stats_kernel, grid_red, dim3(256), 0, stream, a);
  hipLaunchKernelGGL(bn_rows_apply_kernel, grid, dim3(256), 0, stream, a);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_bn_rows_backward(const float *x, const float *dy, long long n, int c, const float *saved, int relu,
                                     double *sums, float *dx, float *dweight, float *dbias, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BnArgs a{};
  DF3D_CHECK_ARG(n >= 1 && c >= 4 && bn_shape(a, n, c), "bn_rows_backward: %lld rows x %d channels has no row kernel", n, c);
  DF3D_CHECK_ARG(x && dy && saved && sums && dx, "bn_rows_backward: null argument");
  a.x = x, a.dy = dy, a.saved = const_cast<float *>(saved), a.sums = sums, a.dx = dx, a.dweight = dweight, a.dbias = dbias;
  a.relu = relu;
  DF3D_HIP(hipMemsetAsync(sums, 0, (size_t)BN_REPL * 2 * c * sizeof(double), stream));
  const dim3 grid((unsigned)cdiv(n, (long long)a.rows_per_wg), c / a.cb);
  const dim3 grid_red((unsigned)cdiv(n, (long long)a.rows_per_wg_red), c / a.cb);
  hipLaunchKernelGGL(bn_rows_bwd_reduce_kernel, grid_red, dim3(256), 0, stream, a);
  hipLaunchKernelGGL(bn_rows_bwd_apply_kernel, grid, dim3(256), 0, stream, a);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_bn_rows_scratch_doubles(int c) { return BN_REPL * 2 * c; }
