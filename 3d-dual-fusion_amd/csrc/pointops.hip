// Point ops of the 3D local self-attention (LocalTransformer) for gfx950:
// D-FPS, ball query, grouping, gathering.  Semantics (including the FPS tie rule) follow
// the reference CUDA kernels cited in include/df3d_hip.h; the oracle restates them in C.
//
// FPS is inherently sequential over m rounds.  The reference keeps min-distances in global
// memory and does a 10-level LDS tree reduction (10 barriers) per round.  Here the running
// min-distances live in REGISTERS (each thread owns points tid, tid+bs, ... exactly like the
// reference, which makes its tie rule fall out naturally), the arg-max is a 6-step wave64
// butterfly plus one LDS hop across <= 16 waves (2 barriers per round), and xyz (<= 312 KB at
// 26 k points) is re-read from L2.  Bound: latency (m x ~1 us per batch element).
#include <math.h>

#include "common.h"
#include <algorithm>

namespace df3d {

DF3D_SPLIT_OVERFLOW_TU(pointops)

// Squared distance as the reference computes it ON ITS GPU: nvcc contracts `dx * dx + dy * dy + dz * dz` (--fmad=true, the
// default) into fma(dz, dz, fma(dy, dy, dx * dx)).  Voxel centres sit on a lattice, so exact ties between candidates are common
// and the LAST BIT of this sum decides which point furthest point sampling picks (found at the full KITTI size in round 6: the
// oracle, compiled without contraction, left the device's sequence at pick 79 of 2048).  Written out so that neither compiler
// chooses: the kernels here and oracle/df3d_oracle.c evaluate the same three roundings.
__device__ __forceinline__ float dist2_fma(float dx, float dy, float dz) {
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

constexpr int FPS_MAXPT = 32;  // points per thread kept in registers (N <= 32 * 1024)

struct Best {
  float v;
  int tid, k;
};
__device__ __forceinline__ Best better(Best a, Best b) {
  // larger value wins; on ties the lower thread id (== k mod block) wins, as __update() does
  bool takeb = (b.v > a.v) || (b.v == a.v && b.tid < a.tid);
  return takeb ? b : a;
}

// REG: running min-distances in registers; PT points per thread; XYZ: the thread's coordinates in registers too
// (PT <= 24: 4*PT live registers fit the 128-VGPR budget of a 1024-thread block)
template <bool REG, int PT, bool XYZ>
__global__ __launch_bounds__(1024) void fps_kernel(const float *__restrict__ xyz, int N, int m, float *__restrict__ temp_g,
                           int32_t *__restrict__ idx) {
  __shared__ float s_v[16];
  __shared__ int s_t[16], s_k[16];
  __shared__ int s_old;
  const int b = blockIdx.x, tid = threadIdx.x, bs = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nw = (bs + 63) >> 6;
  const float *p = xyz + (size_t)b * N * 3;
  float *tg = temp_g + (size_t)b * N;
  int32_t *o = idx + (size_t)b * m;
  float temp[PT];
  float px[XYZ ? PT : 1], py[XYZ ? PT : 1], pz[XYZ ? PT : 1];   // the thread's points, loaded once
#pragma unroll
  for (int i = 0; i < PT; ++i) temp[i] = 1e10f;
  if (XYZ) {
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      int k = tid + i * bs;
      int kk = k < N ? k : N - 1;
      px[i] = p[kk * 3];
      py[i] = p[kk * 3 + 1];
      pz[i] = p[kk * 3 + 2];
    }
  }
  if (!REG)
    for (int k = tid; k < N; k += bs) tg[k] = 1e10f;
  if (tid == 0 && m > 0) o[0] = 0;
  float x1 = p[0], y1 = p[1], z1 = p[2];
  for (int j = 1; j < m; ++j) {
    Best me = {-1.f, tid, 0};
    if (REG) {
#pragma unroll
      for (int i = 0; i < PT; ++i) {
        int k = tid + i * bs;
        if (k < N) {
          float x2 = XYZ ? px[i] : p[k * 3], y2 = XYZ ? py[i] : p[k * 3 + 1], z2 = XYZ ? pz[i] : p[k * 3 + 2];
          float d = dist2_fma(x2 - x1, y2 - y1, z2 - z1);
          float d2 = fminf(d, temp[i]);
          temp[i] = d2;
          if (d2 > me.v) {
            me.v = d2;
            me.k = k;
          }
        }
      }
    } else {
      for (int k = tid; k < N; k += bs) {
        float x2 = p[k * 3], y2 = p[k * 3 + 1], z2 = p[k * 3 + 2];
        float d = dist2_fma(x2 - x1, y2 - y1, z2 - z1);
        float d2 = fminf(d, tg[k]);
        tg[k] = d2;
        if (d2 > me.v) {
          me.v = d2;
          me.k = k;
        }
      }
    }
    // wave butterfly (inactive upper lanes of a partial wave hold v = -1 and never win)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      Best ot;
      ot.v = __shfl_xor(me.v, off, 64);
      ot.tid = __shfl_xor(me.tid, off, 64);
      ot.k = __shfl_xor(me.k, off, 64);
      me = better(me, ot);
    }
    if (lane == 0) {
      s_v[wave] = me.v;
      s_t[wave] = me.tid;
      s_k[wave] = me.k;
    }
    __syncthreads();
    if (tid == 0) {
      Best r = {s_v[0], s_t[0], s_k[0]};
      for (int w = 1; w < nw; ++w) {
        Best ot = {s_v[w], s_t[w], s_k[w]};
        r = better(r, ot);
      }
      s_old = r.k;
      o[j] = r.k;
    }
    __syncthreads();
    const int old = s_old;
    x1 = p[old * 3];
    y1 = p[old * 3 + 1];
    z1 = p[old * 3 + 2];
  }
}

// Ball query (VR/pcdet/ops/pointnet2/pointnet2_stack|batch ball_query: the FIRST nsample points, in index order, whose squared
// distance to the centre is 0 or in [min_r2, max_r2); unused slots repeat the first hit; no hit: zeros).  A WAVE owns a
// centre: 64 points are tested per step, the ballot's prefix count is the hit's slot, so the index order of the reference's
// serial scan is kept and a centre stops as soon as its slots are full.  (Round 1: one THREAD per centre walking all N
// points -- 64 workgroups on the chip, 2.6 ms at 8 x 2048 centres x 24 k points.)  The 16 waves of a workgroup share
// 1024-point tiles in LDS.
constexpr int BQ_WAVES = 16, BQ_TILE = 1024;

__global__ __launch_bounds__(BQ_WAVES * 64) void ball_query_kernel(const float *__restrict__ new_xyz,
                                                                   const float *__restrict__ xyz, int N, int m, float min_r2,
                                                                   float max_r2, int nsample, int32_t *__restrict__ idx) {
  extern __shared__ float bq_smem[];
  float *tile = bq_smem;                                          // [BQ_TILE][3]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int32_t *hits = (int32_t *)(bq_smem + BQ_TILE * 3) + wave * nsample;
  const int b = blockIdx.y;
  const int c = blockIdx.x * BQ_WAVES + wave;
  const bool live = c < m;
  const float *q = new_xyz + ((size_t)b * m + (live ? c : 0)) * 3;
  const float nx = q[0], ny = q[1], nz = q[2];
  const float *pts = xyz + (size_t)b * N * 3;
  int cnt = live ? 0 : nsample;
  for (int t0 = 0; t0 < N; t0 += BQ_TILE) {
    if (__syncthreads_and(cnt >= nsample)) break;                 // every centre of the workgroup is full
    const int tn = N - t0 < BQ_TILE ? N - t0 : BQ_TILE;
    for (int e = tid; e < tn * 3; e += BQ_WAVES * 64) tile[e] = pts[(size_t)t0 * 3 + e];
    __syncthreads();
    for (int k0 = 0; k0 < tn && cnt < nsample; k0 += 64) {
      const int k = k0 + lane;
      bool hit = false;
      if (k < tn) {
        const float x = tile[k * 3], y = tile[k * 3 + 1], z = tile[k * 3 + 2];
        const float d2 = dist2_fma(nx - x, ny - y, nz - z);
        hit = d2 == 0.f || (d2 >= min_r2 && d2 < max_r2);
      }
      const unsigned long long mask = __ballot(hit);
      const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
      if (hit && pos < nsample) hits[pos] = t0 + k;
      cnt += __popcll(mask);
    }
  }
  if (!live) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  cnt = cnt < nsample ? cnt : nsample;
  int32_t *o = idx + ((size_t)b * m + c) * nsample;
  for (int l = lane; l < nsample; l += 64) o[l] = cnt == 0 ? 0 : hits[l < cnt ? l : 0];
}

// Half-size block for 16k < N <= 24k points: 512 threads x PT points each keep distances AND coordinates in
// registers without spilling (the 1024-thread kernel has 128 VGPRs per thread).  The reference's result depends on
// its block size through the tie rule ("equal distance: lower thread id wins", thread id = k mod 1024); every
// physical thread therefore plays two virtual threads and the reduction compares VIRTUAL ids (k & (VT - 1)).
// One iteration = distances (packed fp32), the thread's arg-max as a TREE over its points, the wave's arg-max by DPP
// (no LDS traffic), ONE barrier, and every thread reducing the eight wave results itself (round 2; before: a serial
// compare chain over 48 points, 18 ds_bpermutes, a serial reduction by thread 0 between two barriers: 4.1 us per
// iteration at 24k points).
template <int PT>
__global__ __launch_bounds__(512) void fps_kernel_half(const float *__restrict__ xyz, int N, int m, int VT,
                                                       int32_t *__restrict__ idx) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  static_assert(PT % 4 == 0, "points are processed in same-parity pairs");
  constexpr int NQ = PT / 4;
  __shared__ float s_v[2][8];
  __shared__ int s_t[2][8], s_k[2][8];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const float *p = xyz + (size_t)b * N * 3;
  int32_t *o = idx + (size_t)b * m;
  // point i of this thread is k = tid + 512 i; its virtual thread (the reference's 1024-thread block) is tid for even i
  // and tid + 512 for odd i.  Pairs (i, i + 2) share the parity: [parity e][pair q] <-> i = 4q + e, 4q + e + 2.
  // Two points per packed instruction (v_pk_add / v_pk_mul / v_pk_fma_f32).
  f2 temp[2][NQ], px[2][NQ], py[2][NQ], pz[2][NQ];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = tid + (4 * q + e + 2 * h) * 512;
        const int kk = k < N ? k : N - 1;
        // a slot past the end never wins: its running minimum starts (and stays) below every real distance
        temp[e][q][h] = k < N ? 1e10f : -1.f;
        px[e][q][h] = p[kk * 3];
        py[e][q][h] = p[kk * 3 + 1];
        pz[e][q][h] = p[kk * 3 + 2];
      }
  if (tid == 0 && m > 0) o[0] = 0;
  float x1 = p[0], y1 = p[1], z1 = p[2];
  const int vmask = VT - 1;
  // wave-level all-reduce steps that stay in the VALU: xor 1, xor 2 inside a quad, then mirrors inside 8 and 16 lanes
  // (any pairing works: `better` is a total order)
#define FPS_DPP_STEP(CTRL)                                                                            \
  do {                                                                                                \
    Best ot;                                                                                          \
    ot.v = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, me.v), CTRL, 0xf, 0xf, false)); \
    ot.tid = __builtin_amdgcn_update_dpp(0, me.tid, CTRL, 0xf, 0xf, false);                           \
    ot.k = __builtin_amdgcn_update_dpp(0, me.k, CTRL, 0xf, 0xf, false);                               \
    me = better(me, ot);                                                                              \
  } while (0)
  for (int j = 1; j < m; ++j) {
    // The reference's thread scans its points in increasing k with a strict '>' (the first maximum stays); the block
    // reduction breaks ties towards the lower thread id.  For a physical thread playing two virtual threads that is:
    // all even-i points first (virtual thread tid), then the odd ones (tid + 512), strict '>' throughout -- i.e. the
    // maximum, ties to the EARLIER slot of the list [e = 0: q, h ascending][e = 1: q, h ascending].  A tree over that
    // list in which the right operand wins only when strictly greater gives the same slot with depth 6 instead of 48.
    float cv[PT];
    int cs[PT];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const f2 dx = px[e][q] - x1, dy = py[e][q] - y1, dz = pz[e][q] - z1;
        const f2 d = (f2){dist2_fma(dx[0], dy[0], dz[0]), dist2_fma(dx[1], dy[1], dz[1])};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float d2 = fminf(d[h], temp[e][q][h]);
          temp[e][q][h] = d2;
          cv[e * (PT / 2) + q * 2 + h] = d2;
          cs[e * (PT / 2) + q * 2 + h] = 4 * q + e + 2 * h;          // slot -> i
        }
      }
#pragma unroll
    for (int w = 1; w < PT; w *= 2)
#pragma unroll
      for (int a0 = 0; a0 + w < PT; a0 += 2 * w) {
        const bool take = cv[a0 + w] > cv[a0];
        cv[a0] = take ? cv[a0 + w] : cv[a0];
        cs[a0] = take ? cs[a0 + w] : cs[a0];
      }
    const float bv = cv[0];
    const int bk = tid + cs[0] * 512;
    Best me = {bv, bv < 0.f ? 0x7fffffff : (bk & vmask), bk};
    FPS_DPP_STEP(0xB1);                              // quad_perm [1, 0, 3, 2]
    FPS_DPP_STEP(0x4E);                              // quad_perm [2, 3, 0, 1]
    FPS_DPP_STEP(0x141);                             // row_half_mirror
    FPS_DPP_STEP(0x140);                             // row_mirror: every lane of a row of 16 holds the row's best
    Best wb = {__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, me.v), 0)),
               __builtin_amdgcn_readlane(me.tid, 0), __builtin_amdgcn_readlane(me.k, 0)};
#pragma unroll
    for (int r = 1; r < 4; ++r) {
      Best ot = {__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, me.v), r * 16)),
                 __builtin_amdgcn_readlane(me.tid, r * 16), __builtin_amdgcn_readlane(me.k, r * 16)};
      wb = better(wb, ot);
    }
    const int par = j & 1;                           // two result buffers: the next iteration writes the other one
    if (lane == 0) {
      s_v[par][wave] = wb.v;
      s_t[par][wave] = wb.tid;
      s_k[par][wave] = wb.k;
    }
    __syncthreads();
    Best r = {s_v[par][0], s_t[par][0], s_k[par][0]};
#pragma unroll
    for (int w = 1; w < 8; ++w) {
      Best ot = {s_v[par][w], s_t[par][w], s_k[par][w]};
      r = better(r, ot);
    }
    const int old = r.k;
    if (tid == 0) o[j] = old;
    x1 = p[old * 3];
    y1 = p[old * 3 + 1];
    z1 = p[old * 3 + 2];
  }
#undef FPS_DPP_STEP
}

// out[b,c,p,s] = feat[b,c,idx[b,p,s]]  (gather_points: nsample == 1)
__global__ __launch_bounds__(256) void group_points_kernel(const float *__restrict__ feat,
                                                           const int32_t *__restrict__ idx, int C, int N, int np,
                                                           int ns, float *__restrict__ out) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= np * ns) return;
  int i = idx[(size_t)b * np * ns + t];
  out[((size_t)b * C + c) * np * ns + t] = feat[((size_t)b * C + c) * N + i];
}

}  // namespace df3d

using namespace df3d;

static int fps_block(int n) {
  // furthest_point_sample_cuda.cu:9-15 (opt_n_threads): largest power of two <= n, <= 1024
  int pow_2 = (int)(log((double)n) / log(2.0));
  int t = 1 << pow_2;
  if (t > 1024) t = 1024;
  if (t < 1) t = 1;
  return t;
}

// F-FPS form of the sampler (furthest_point_sample_cuda.cu:213-330): the distance of point k to the last pick comes from a
// precomputed [N, N] matrix instead of coordinates.  Same per-thread strided scan and the same tie rule as fps_kernel
// (running minima in `temp_g`, as the reference keeps them).
__global__ __launch_bounds__(1024) void fps_with_dist_kernel(const float *__restrict__ dist, int N, int m,
                                                             float *__restrict__ temp_g, int32_t *__restrict__ idx) {
  __shared__ float s_v[16];
  __shared__ int s_t[16], s_k[16];
  __shared__ int s_old;
  const int b = blockIdx.x, tid = threadIdx.x, bs = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nw = (bs + 63) >> 6;
  const float *d = dist + (size_t)b * N * N;
  float *tg = temp_g + (size_t)b * N;
  int32_t *o = idx + (size_t)b * m;
  for (int k = tid; k < N; k += bs) tg[k] = 1e10f;
  if (tid == 0 && m > 0) o[0] = 0;
  int old = 0;
  for (int j = 1; j < m; ++j) {
    Best me = {-1.f, tid, 0};
    for (int k = tid; k < N; k += bs) {
      const float d2 = fminf(d[(size_t)old * N + k], tg[k]);
      tg[k] = d2;
      if (d2 > me.v) {
        me.v = d2;
        me.k = k;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      Best ot;
      ot.v = __shfl_xor(me.v, off, 64);
      ot.tid = __shfl_xor(me.tid, off, 64);
      ot.k = __shfl_xor(me.k, off, 64);
      me = better(me, ot);
    }
    if (lane == 0) {
      s_v[wave] = me.v;
      s_t[wave] = me.tid;
      s_k[wave] = me.k;
    }
    __syncthreads();
    if (tid == 0) {
      Best r = {s_v[0], s_t[0], s_k[0]};
      for (int w = 1; w < nw; ++w) {
        Best ot = {s_v[w], s_t[w], s_k[w]};
        r = better(r, ot);
      }
      s_old = r.k;
      o[j] = r.k;
    }
    __syncthreads();
    old = s_old;
  }
}

// Backward of group_points / gather_points (group_points_cuda.cu:10-31, gather_points_cuda.cu:48-70): the output gradient is
// scatter-added onto the source points; thread = one (channel, output element), fp32 hardware atomics.
__global__ __launch_bounds__(256) void index_add_grad_kernel(const float *__restrict__ grad_out, const int32_t *__restrict__ idx,
                                                             int C, int N, long long per, float *__restrict__ grad_points) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y, b = blockIdx.z;
  if (i >= per) return;
  const int src = idx[(size_t)b * per + i];
  if (src < 0 || src >= N) return;
  atomicAdd(grad_points + ((size_t)b * C + c) * N + src, grad_out[((size_t)b * C + c) * per + i]);
}

extern "C" int df3d_furthest_point_sample(const float *xyz, int B, int N, int m, float *temp, int32_t *idx,
                                          void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(xyz && idx && temp, "furthest_point_sample: null argument");
  DF3D_CHECK_ARG(B >= 0 && N > 0 && m >= 0, "furthest_point_sample: bad sizes");
  if (B == 0 || m == 0) return DF3D_OK;
  int bs = fps_block(N);
  const long long ppt = ((long long)N + bs - 1) / bs;      // points per thread
  if (ppt <= 8)
    hipLaunchKernelGGL((fps_kernel<true, 8, true>), dim3(B), dim3(bs), 0, stream, xyz, N, m, temp, idx);
  else if (ppt <= 16)
    hipLaunchKernelGGL((fps_kernel<true, 16, true>), dim3(B), dim3(bs), 0, stream, xyz, N, m, temp, idx);
  else if (ppt <= 24 && bs == 1024)
    hipLaunchKernelGGL((fps_kernel_half<48>), dim3(B), dim3(512), 0, stream, xyz, N, m, bs, idx);
  else if (ppt <= 24)
    hipLaunchKernelGGL((fps_kernel<true, 24, true>), dim3(B), dim3(bs), 0, stream, xyz, N, m, temp, idx);
  else if (ppt <= FPS_MAXPT)
    hipLaunchKernelGGL((fps_kernel<true, FPS_MAXPT, false>), dim3(B), dim3(bs), 0, stream, xyz, N, m, temp, idx);
  else
    hipLaunchKernelGGL((fps_kernel<false, 1, false>), dim3(B), dim3(bs), 0, stream, xyz, N, m, temp, idx);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

// Self-attention inside small groups (the LocalTransformer of ACTRv2, VR/pcdet/models/.../pointformer.py:10-44 through
// nn.MultiheadAttention: L = nsample tokens per ball-query group, heads of 16 channels).  qkv [L * G][3 * C] fp32 rows in the
// encoder's sequence-first order (row = token * G + group; q | k | v column blocks, the in-projection's output), out
// [L * G][C].  A workgroup owns a group: K and V of its L tokens in LDS, thread = (token, head): scores against the L keys
// with an online softmax, 16 accumulators.  The library's flash kernel spends 300 us per layer on these 16 k sequences of 32.
template <int D>
__global__ void group_attention_kernel(const float *__restrict__ qkv, int L, int G, int H, float scale,
                                       float *__restrict__ out, unsigned *__restrict__ out_split) {
  extern __shared__ __align__(16) float ga_smem[];
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int C = H * D, g = blockIdx.x, tid = threadIdx.x;
  float *sk = ga_smem, *sv = ga_smem + (size_t)L * C;
  const int c4n = C / 4;
  for (int it = tid; it < L * c4n * 2; it += blockDim.x) {
    const int kv = it / (L * c4n), r = it - kv * (L * c4n);
    const int t = r / c4n, c4 = r - t * c4n;
    const f4 v = *(const f4 *)(qkv + ((size_t)t * G + g) * 3 * C + (1 + kv) * C + c4 * 4);
    *(f4 *)((kv ? sv : sk) + t * C + c4 * 4) = v;
  }
  __syncthreads();
  if (tid >= L * H) return;
  const int h = tid / L, t = tid - h * L;
  float q[D], acc[D];
  const float *qp = qkv + ((size_t)t * G + g) * 3 * C + h * D;
#pragma unroll
  for (int e = 0; e < D; e += 4) {
    const f4 v = *(const f4 *)(qp + e);
    q[e] = v[0] * scale, q[e + 1] = v[1] * scale, q[e + 2] = v[2] * scale, q[e + 3] = v[3] * scale;
  }
#pragma unroll
  for (int e = 0; e < D; ++e) acc[e] = 0.f;
  float m = -INFINITY, l = 0.f;
  for (int j = 0; j < L; ++j) {
    const float *kp = sk + j * C + h * D, *vp = sv + j * C + h * D;
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < D; e += 4) {
      const f4 k4 = *(const f4 *)(kp + e);
      s = fmaf(q[e], k4[0], s), s = fmaf(q[e + 1], k4[1], s), s = fmaf(q[e + 2], k4[2], s), s = fmaf(q[e + 3], k4[3], s);
    }
    const float mn = fmaxf(m, s);
    const float a = __expf(m - mn), p = __expf(s - mn);
    l = l * a + p;
#pragma unroll
    for (int e = 0; e < D; e += 4) {
      const f4 v4 = *(const f4 *)(vp + e);
      acc[e] = fmaf(p, v4[0], acc[e] * a), acc[e + 1] = fmaf(p, v4[1], acc[e + 1] * a);
      acc[e + 2] = fmaf(p, v4[2], acc[e + 2] * a), acc[e + 3] = fmaf(p, v4[3], acc[e + 3] * a);
    }
    m = mn;
  }
  const float inv = 1.f / l;
#pragma unroll
  for (int e = 0; e < D; ++e) acc[e] *= inv;
  if (out) {
    float *op = out + ((size_t)t * G + g) * C + h * D;
#pragma unroll
    for (int e = 0; e < D; e += 4) *(f4 *)(op + e) = (f4){acc[e], acc[e + 1], acc[e + 2], acc[e + 3]};
  }
  if (out_split) {           // split rows for the out-projection (per 8 channels: 16 B bf16 hi | 16 B bf16 lo)
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    unsigned *sp = out_split + ((size_t)t * G + g) * C + h * D;          // u32 units: one per channel
#pragma unroll
    for (int b = 0; b < D / 8; ++b) {
      u4 hi, lo;
      split_pair(acc[8 * b], acc[8 * b + 1], hi[0], lo[0]);
      split_pair(acc[8 * b + 2], acc[8 * b + 3], hi[1], lo[1]);
      split_pair(acc[8 * b + 4], acc[8 * b + 5], hi[2], lo[2]);
      split_pair(acc[8 * b + 6], acc[8 * b + 7], hi[3], lo[3]);
      *(u4 *)(sp + 8 * b) = hi;
      *(u4 *)(sp + 8 * b + 4) = lo;
    }
  }
}

static int group_attention_impl(const float *qkv, int tokens, int groups, int heads, int head_dim, float *out,
                                unsigned *out_split, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(tokens >= 1 && groups >= 0 && heads >= 1 && head_dim == 16, "group_attention: heads of 16 channels only (got %d)",
                 head_dim);
  const int C = heads * head_dim;
  const size_t lds = (size_t)2 * tokens * C * sizeof(float);
  DF3D_CHECK_ARG(tokens * heads <= 1024 && lds <= 64 * 1024, "group_attention: %d tokens x %d heads does not fit a workgroup",
                 tokens, heads);
  if (groups == 0) return DF3D_OK;               // (empty tensors carry null pointers)
  DF3D_CHECK_ARG(qkv && (out || out_split), "group_attention: null argument");
  const int threads = cdiv(tokens * heads, 64) * 64;
  hipLaunchKernelGGL(group_attention_kernel<16>, dim3(groups), dim3(threads), lds, stream, qkv, tokens, groups, heads,
                     1.f / sqrtf((float)head_dim), out, out_split);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_group_attention(const float *qkv, int tokens, int groups, int heads, int head_dim, float *out,
                                    void *stream_) {
  return group_attention_impl(qkv, tokens, groups, heads, head_dim, out, nullptr, stream_);
}

extern "C" int df3d_group_attention_split(const float *qkv, int tokens, int groups, int heads, int head_dim, float *out,
                                          void *out_split, void *stream_) {
  return group_attention_impl(qkv, tokens, groups, heads, head_dim, out, (unsigned *)out_split, stream_);
}

// Grouped features + positional MLP of the LocalTransformer in one pass (pointformer.py:232-262: x = group(features) +
// pe(grouped xyz), pe = Conv1x1(3 -> C/2) + BN + ReLU + Conv1x1(C/2 -> C); BN folded by the caller):
//   out[r][:] = feat[sel[r]][:] + W1 relu(W0 xyz[r] + b0) + b1
// 16 lanes x 4 channels per row (C = 64), the 3 -> C/2 layer recomputed per lane (96 FMAs), W1 transposed in LDS.  The library
// runs the two tall-skinny products (K = 3, K = 32 over 524 k rows) at 300 us each and needs a gather and an add on top.
__global__ __launch_bounds__(256) void pe_gather_add_kernel(const float *__restrict__ feat, const int64_t *__restrict__ sel,
                                                            const float *__restrict__ xyz, const float *__restrict__ w0,
                                                            const float *__restrict__ b0, const float *__restrict__ w1,
                                                            const float *__restrict__ b1, long long rows, int C, int H1,
                                                            float *__restrict__ out) {
  extern __shared__ __align__(16) float pe_smem[];
  typedef float f4 __attribute__((ext_vector_type(4)));
  float *w1t = pe_smem;                     // [H1][C]
  float *w0s = pe_smem + (size_t)H1 * C;     // [H1][4] = w0 row | b0
  for (int i = threadIdx.x; i < H1 * C; i += 256) {
    const int k = i / C, o = i - k * C;
    w1t[i] = w1[(size_t)o * H1 + k];
  }
  for (int i = threadIdx.x; i < H1; i += 256) {
    w0s[i * 4] = w0[i * 3], w0s[i * 4 + 1] = w0[i * 3 + 1], w0s[i * 4 + 2] = w0[i * 3 + 2], w0s[i * 4 + 3] = b0[i];
  }
  __syncthreads();
  const int lpr = C / 4;                     // lanes per row
  const int rpb = 256 / lpr;
  const int sub = threadIdx.x % lpr;
  for (long long r = (long long)blockIdx.x * rpb + threadIdx.x / lpr; r < rows; r += (long long)gridDim.x * rpb) {
    const float x = xyz[r * 3], y = xyz[r * 3 + 1], z = xyz[r * 3 + 2];
    f4 acc = *(const f4 *)(b1 + sub * 4);
    for (int k = 0; k < H1; ++k) {
      const f4 w = *(const f4 *)(w0s + k * 4);
      const float h = fmaxf(fmaf(w[0], x, fmaf(w[1], y, fmaf(w[2], z, w[3]))), 0.f);
      const f4 v = *(const f4 *)(w1t + (size_t)k * C + sub * 4);
      acc[0] = fmaf(h, v[0], acc[0]), acc[1] = fmaf(h, v[1], acc[1]), acc[2] = fmaf(h, v[2], acc[2]), acc[3] = fmaf(h, v[3], acc[3]);
    }
    const f4 f = *(const f4 *)(feat + (size_t)sel[r] * C + sub * 4);
    *(f4 *)(out + r * C + sub * 4) = acc + f;
  }
}

extern "C" int df3d_pe_gather_add(const float *feat, const int64_t *sel, const float *xyz, const float *w0, const float *b0,
                                  const float *w1, const float *b1, long long rows, int channels, int hidden, float *out,
                                  void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(rows == 0 || (feat && sel && xyz && w0 && b0 && w1 && b1 && out), "pe_gather_add: null argument");
  DF3D_CHECK_ARG(channels % 4 == 0 && channels >= 4 && 256 % (channels / 4) == 0 && hidden >= 1 &&
                     (size_t)hidden * (channels + 4) * 4 <= 64 * 1024,
                 "pe_gather_add: %d channels / %d hidden not served", channels, hidden);
  if (rows == 0) return DF3D_OK;
  const int rpb = 256 / (channels / 4);
  const long long blocks = std::min<long long>(cdiv(rows, rpb), 4096);
  hipLaunchKernelGGL(pe_gather_add_kernel, dim3((unsigned)blocks), dim3(256), (size_t)hidden * (channels + 4) * sizeof(float),
                     stream, feat, sel, xyz, w0, b0, w1, b1, rows, channels, hidden, out);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_ball_query(const float *new_xyz, const float *xyz, int B, int N, int m, float min_radius,
                               float max_radius, int nsample, int32_t *idx, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(nsample > 0, "ball_query: bad arguments");
  if (B == 0 || m == 0) return DF3D_OK;
  DF3D_CHECK_ARG(new_xyz && xyz && idx, "ball_query: null argument");
  DF3D_CHECK_ARG(nsample <= 1024, "ball_query: at most 1024 samples per centre (got %d)", nsample);
  // (every slot of every centre is written: no hit = zeros, like the reference's zero-initialised idx)
  const size_t lds = (size_t)(BQ_TILE * 3 + BQ_WAVES * nsample) * sizeof(float);
  if (lds > 64 * 1024) {                               // nsample > 832: beyond the default dynamic-LDS limit (160 KB per CU)
    static bool raised = false;
    if (!raised) {
      DF3D_HIP(hipFuncSetAttribute((const void *)ball_query_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)((BQ_TILE * 3 + BQ_WAVES * 1024) * sizeof(float))));
      raised = true;
    }
  }
  hipLaunchKernelGGL(ball_query_kernel, dim3(cdiv(m, BQ_WAVES), B), dim3(BQ_WAVES * 64), lds, stream, new_xyz, xyz, N, m,
                     min_radius * min_radius, max_radius * max_radius, nsample, idx);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_group_points(const float *features, const int32_t *idx, int B, int C, int N, int npoint,
                                 int nsample, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(features && idx && out, "group_points: null argument");
  if (B == 0 || C == 0 || npoint == 0 || nsample == 0) return DF3D_OK;
  DF3D_CHECK_ARG(C <= 65535 && B <= 65535, "group_points: B or C > 65535");
  hipLaunchKernelGGL(group_points_kernel, dim3(cdiv((long long)npoint * nsample, 256), C, B), dim3(256), 0, stream,
                     features, idx, C, N, npoint, nsample, out);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_gather_points(const float *features, const int32_t *idx, int B, int C, int N, int npoint,
                                  float *out, void *stream_) {
  return df3d_group_points(features, idx, B, C, N, npoint, 1, out, stream_);
}

extern "C" int df3d_furthest_point_sample_with_dist(const float *dist, int B, int N, int m, float *temp, int32_t *idx,
                                                    void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(dist && idx && temp, "furthest_point_sample_with_dist: null argument");
  DF3D_CHECK_ARG(B >= 0 && N > 0 && m >= 0, "furthest_point_sample_with_dist: bad sizes");
  if (B == 0 || m == 0) return DF3D_OK;
  hipLaunchKernelGGL(fps_with_dist_kernel, dim3(B), dim3(fps_block(N)), 0, stream, dist, N, m, temp, idx);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_group_points_grad(const float *grad_out, const int32_t *idx, int B, int C, int N, int npoint, int nsample,
                                      float *grad_points, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(B >= 0 && C >= 0 && N > 0 && npoint >= 0 && nsample >= 0, "group_points_grad: bad sizes");
  const long long per = (long long)npoint * nsample;
  if (B == 0 || C == 0 || per == 0) return DF3D_OK;
  DF3D_CHECK_ARG(grad_out && idx && grad_points, "group_points_grad: null argument");
  DF3D_CHECK_ARG(C <= 65535 && B <= 65535, "group_points_grad: more than 65535 channels / samples");
  hipLaunchKernelGGL(index_add_grad_kernel, dim3(cdiv(per, 256), C, B), dim3(256), 0, stream, grad_out, idx, C, N, per,
                     grad_points);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}

extern "C" int df3d_gather_points_grad(const float *grad_out, const int32_t *idx, int B, int C, int N, int npoint,
                                       float *grad_points, void *stream) {
  return df3d_group_points_grad(grad_out, idx, B, C, N, npoint, 1, grad_points, stream);
}
