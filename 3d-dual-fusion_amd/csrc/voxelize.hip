// Hard voxelisation + fused mean VFE for gfx950.
//
// Replaces the reference's GPU path (TF/mmdet3d/ops/voxel/src/voxelization_cuda.cu:25-326:
// an O(P^2) "scan all previous points" kernel, a <<<1,1>>> sequential numbering kernel and
// four device syncs) and its CPU path (voxelization_cpu.cpp:43-141: 340 MB dense grid per
// frame) with a deterministic parallel formulation that yields BIT-IDENTICAL results:
//
//   1. hash the voxel key of every in-range point (open addressing, atomicCAS) and keep the
//      smallest point index per voxel (atomicMin)      -> "first appearance" of each voxel
//   2. flag first-appearance points, exclusive-scan the flags -> voxel id = rank in point order
//      (exactly the numbering a sequential pass produces); the point that would open voxel
//      #max_voxels is the reference's `break` position
//   3. per voxel keep the max_points smallest point indices with a min-cascade
//      (x = atomicMin(slot_t, x) carried forward): arrival order without sorting
//   4. one thread per voxel gathers its points, zero-pads, writes coors / num and the mean.
//
// HBM traffic: 20*P bytes of points read twice + M*(T*C*4 + 36) written; everything else is
// L2-resident scratch.  Bound: HBM / latency (P ~ 60 k-300 k points).
#include "common.h"

namespace df3d {

struct VoxParams {
  float min[3], vs[3];
  int grid[3];  // x, y, z
  int P, C, T, maxV, cap_mask;
  int batch_index;   // >= 0: coors rows are (batch, z, y, x); < 0: (z, y, x)
};

__device__ __forceinline__ uint32_t hash_u32(uint32_t k) {
  k ^= k >> 16;
  k *= 0x85ebca6bu;
  k ^= k >> 13;
  k *= 0xc2b2ae35u;
  k ^= k >> 16;
  return k;
}

__global__ __launch_bounds__(256) void vox_insert_kernel(const float *__restrict__ pts, VoxParams p,
                                                         int *__restrict__ keys, int *__restrict__ first,
                                                         int *__restrict__ slot_of_point) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.P) return;
  const float *q = pts + (size_t)i * p.C;
  int c[3];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    // same float expression as voxelization_cpu.cpp:22 (IEEE divide, then floor)
    float f = floorf((q[j] - p.min[j]) / p.vs[j]);
    if (!(f >= 0.0f && f < (float)p.grid[j])) ok = false;
    c[j] = (int)f;
  }
  if (!ok) {
    slot_of_point[i] = -1;
    return;
  }
  int key = (c[2] * p.grid[1] + c[1]) * p.grid[0] + c[0];  // (z*GY + y)*GX + x
  uint32_t s = hash_u32((uint32_t)key) & (uint32_t)p.cap_mask;
  while (true) {
    int prev = atomicCAS(&keys[s], -1, key);
    if (prev == -1 || prev == key) break;
    s = (s + 1) & (uint32_t)p.cap_mask;
  }
  atomicMin(&first[s], i);
  slot_of_point[i] = (int)s;
}

__global__ __launch_bounds__(256) void vox_flag_kernel(int P, const int *__restrict__ first,
                                                       const int *__restrict__ slot_of_point,
                                                       uint32_t *__restrict__ flag) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  int s = slot_of_point[i];
  flag[i] = (s >= 0 && first[s] == i) ? 1u : 0u;
}

// misc[0] = total voxels (uncapped, from the scan), misc[1] = i_break, misc[2] = voxel_num
__global__ __launch_bounds__(256) void vox_number_kernel(VoxParams p, const int *__restrict__ keys,
                                                         const int *__restrict__ first,
                                                         const int *__restrict__ slot_of_point,
                                                         const uint32_t *__restrict__ rank, int *__restrict__ vid,
                                                         int32_t *__restrict__ coors, int *__restrict__ misc,
                                                         int32_t *__restrict__ voxel_num) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    int tot = misc[0];
    int n = tot < p.maxV ? tot : p.maxV;
    misc[2] = n;
    *voxel_num = n;
  }
  if (i >= p.P) return;
  int s = slot_of_point[i];
  if (s < 0 || first[s] != i) return;
  int r = (int)rank[i];
  vid[s] = r;
  if (r == p.maxV) misc[1] = i;  // the point at which the reference's loop breaks
  if (r < p.maxV) {
    int key = keys[s];
    int x = key % p.grid[0];
    int t = key / p.grid[0];
    int y = t % p.grid[1];
    int z = t / p.grid[1];
    if (p.batch_index >= 0) {
      *(int4 *)(coors + (size_t)r * 4) = make_int4(p.batch_index, z, y, x);
    } else {
      coors[r * 3 + 0] = z;
      coors[r * 3 + 1] = y;
      coors[r * 3 + 2] = x;
    }
  }
}

__global__ __launch_bounds__(256) void vox_assign_kernel(VoxParams p, int break_at_cap,
                                                         const int *__restrict__ slot_of_point,
                                                         const int *__restrict__ vid, const int *__restrict__ misc,
                                                         int *__restrict__ count, int *__restrict__ ptlist) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.P) return;
  int s = slot_of_point[i];
  if (s < 0) return;
  int v = vid[s];
  if (v >= p.maxV) return;
  if (break_at_cap && i >= misc[1]) return;
  atomicAdd(&count[v], 1);
  int *lst = ptlist + (size_t)v * p.T;
  int x = i;
  for (int t = 0; t < p.T; ++t) {
    int old = atomicMin(&lst[t], x);
    if (old == 0x7f7f7f7f) break;  // slot was empty: x (or the smaller value) is placed, nothing to carry
    x = old > x ? old : x;
  }
}

__global__ __launch_bounds__(256) void vox_gather_kernel(const float *__restrict__ pts, VoxParams p,
                                                         const int *__restrict__ misc, const int *__restrict__ count,
                                                         const int *__restrict__ ptlist, float *__restrict__ voxels,
                                                         int32_t *__restrict__ num, float *__restrict__ mean) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= misc[2]) return;
  int n = count[v];
  if (n > p.T) n = p.T;
  num[v] = n;
  const int *lst = ptlist + (size_t)v * p.T;
  for (int c = 0; c < p.C; ++c) {
    float acc = 0.f;
    for (int t = 0; t < p.T; ++t) {
      float x = 0.f;
      if (t < n) x = pts[(size_t)lst[t] * p.C + c];
      if (voxels) voxels[((size_t)v * p.T + t) * p.C + c] = x;
      acc += x;
    }
    if (mean) mean[(size_t)v * p.C + c] = acc / (float)n;
  }
}

static int table_capacity(int P) {
  int cap = 1024;
  while (cap < 2 * P) cap <<= 1;
  return cap;
}

// workspace initialisation in one launch: keys = -1, first = 0x7f7f7f7f, count = 0, ptlist = 0x7f7f7f7f,
// misc[16] = 0x7f7f7f7f (i_break = "never")
__global__ __launch_bounds__(256) void vox_init_kernel(int *__restrict__ keys, size_t n_keys, int *__restrict__ first,
                                                       size_t n_first, int *__restrict__ count, size_t n_count,
                                                       int *__restrict__ ptlist, size_t n_pt, int *__restrict__ misc) {
  const size_t total = n_keys + n_first + n_count + n_pt + 16;
  const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  for (int e = 0; e < 4; ++e) {
    size_t j = i0 + e;
    if (j >= total) return;
    if (j < n_keys) {
      keys[j] = -1;
      continue;
    }
    j -= n_keys;
    if (j < n_first) {
      first[j] = 0x7f7f7f7f;
      continue;
    }
    j -= n_first;
    if (j < n_count) {
      count[j] = 0;
      continue;
    }
    j -= n_count;
    if (j < n_pt) {
      ptlist[j] = 0x7f7f7f7f;
      continue;
    }
    misc[j - n_pt] = 0x7f7f7f7f;
  }
}

}  // namespace df3d

using namespace df3d;

extern "C" size_t df3d_hard_voxelize_workspace_bytes(int num_points, int max_points, int max_voxels) {
  if (num_points < 0) return 0;
  if (max_voxels < 0 || max_voxels > num_points) max_voxels = num_points;
  size_t cap = (size_t)table_capacity(num_points);
  size_t b = 0;
  b = arena_need(b, cap * 4);                              // keys
  b = arena_need(b, cap * 4);                              // first
  b = arena_need(b, cap * 4);                              // vid
  b = arena_need(b, (size_t)num_points * 4);               // slot_of_point
  b = arena_need(b, (size_t)num_points * 4);               // flag / rank
  b = arena_need(b, (size_t)max_voxels * 4);               // count
  b = arena_need(b, (size_t)max_voxels * (size_t)(max_points > 0 ? max_points : 1) * 4);  // ptlist
  b = arena_need(b, 64);                                   // misc
  b = arena_need(b, scan_scratch_bytes((size_t)num_points));
  return b + 256;
}

static int hard_voxelize_impl(const float *points, int num_points, int num_features, const float *voxel_size,
                              const float *coors_range, int max_points, int max_voxels, int break_at_cap,
                              float *voxels, int32_t *coors, int batch_index, int32_t *num_points_per_voxel,
                              float *mean, int32_t *voxel_num, void *workspace, size_t workspace_bytes,
                              void *stream_);

extern "C" int df3d_hard_voxelize(const float *points, int num_points, int num_features,
                                  const float *voxel_size, const float *coors_range, int max_points,
                                  int max_voxels, int break_at_cap, float *voxels, int32_t *coors,
                                  int32_t *num_points_per_voxel, float *mean, int32_t *voxel_num,
                                  void *workspace, size_t workspace_bytes, void *stream_) {
  return hard_voxelize_impl(points, num_points, num_features, voxel_size, coors_range, max_points, max_voxels,
                            break_at_cap, voxels, coors, -1, num_points_per_voxel, mean, voxel_num, workspace,
                            workspace_bytes, stream_);
}

extern "C" int df3d_hard_voxelize_batched(const float *points, int num_points, int num_features,
                                          const float *voxel_size, const float *coors_range, int max_points,
                                          int max_voxels, int break_at_cap, int batch_index, float *voxels,
                                          int32_t *coors4, int32_t *num_points_per_voxel, float *mean,
                                          int32_t *voxel_num, void *workspace, size_t workspace_bytes, void *stream_) {
  DF3D_CHECK_ARG(batch_index >= 0, "hard_voxelize_batched: negative batch index");
  return hard_voxelize_impl(points, num_points, num_features, voxel_size, coors_range, max_points, max_voxels,
                            break_at_cap, voxels, coors4, batch_index, num_points_per_voxel, mean, voxel_num, workspace,
                            workspace_bytes, stream_);
}

static int hard_voxelize_impl(const float *points, int num_points, int num_features, const float *voxel_size,
                              const float *coors_range, int max_points, int max_voxels, int break_at_cap,
                              float *voxels, int32_t *coors, int batch_index, int32_t *num_points_per_voxel,
                              float *mean, int32_t *voxel_num, void *workspace, size_t workspace_bytes,
                              void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_CHECK_ARG(num_points >= 0 && num_features >= 3, "hard_voxelize: need points [P,C>=3]");
  DF3D_CHECK_ARG(max_points > 0, "hard_voxelize: max_points must be > 0 (got %d)", max_points);
  DF3D_CHECK_ARG(coors && num_points_per_voxel && voxel_num, "hard_voxelize: null output");
  if (max_voxels < 0 || max_voxels > num_points) max_voxels = num_points;  // -1 = unlimited
  VoxParams p;
  long long vol = 1;
  for (int i = 0; i < 3; ++i) {
    p.min[i] = coors_range[i];
    p.vs[i] = voxel_size[i];
    // voxelization_cpu.cpp:119-122: round((max - min) / vs) in float
    p.grid[i] = (int)roundf((coors_range[3 + i] - coors_range[i]) / voxel_size[i]);
    DF3D_CHECK_ARG(p.grid[i] > 0, "hard_voxelize: empty grid on axis %d", i);
    vol *= p.grid[i];
  }
  DF3D_CHECK_ARG(vol < 2147483647LL, "hard_voxelize: grid volume %lld overflows int32 keys", vol);
  p.P = num_points;
  p.C = num_features;
  p.T = max_points;
  p.maxV = max_voxels;
  p.batch_index = batch_index;
  if (num_points == 0) {
    DF3D_HIP(hipMemsetAsync(voxel_num, 0, sizeof(int32_t), stream));
    return DF3D_OK;
  }
  int cap = table_capacity(num_points);
  p.cap_mask = cap - 1;
  if (workspace_bytes < df3d_hard_voxelize_workspace_bytes(num_points, max_points, max_voxels)) {
    set_error("hard_voxelize: workspace too small");
    return DF3D_ENOMEM;
  }
  Arena ar(workspace, workspace_bytes);
  int *keys = ar.take<int>(cap);
  int *first = ar.take<int>(cap);
  int *vid = ar.take<int>(cap);
  int *slot = ar.take<int>(num_points);
  uint32_t *rank = ar.take<uint32_t>(num_points);
  int *count = ar.take<int>(max_voxels);
  int *ptlist = ar.take<int>((size_t)max_voxels * max_points);
  int *misc = ar.take<int>(16);
  size_t ssz = scan_scratch_bytes((size_t)num_points);
  void *sscr = ar.take<char>(ssz);
  if (!sscr) {
    set_error("hard_voxelize: workspace too small");
    return DF3D_ENOMEM;
  }
  // one launch instead of five memsets (each costs ~5 us of launch latency at the head of every frame)
  {
    const size_t n_keys = (size_t)cap, n_first = (size_t)cap, n_count = (size_t)max_voxels,
                 n_pt = (size_t)max_voxels * max_points, total = n_keys + n_first + n_count + n_pt + 16;
    hipLaunchKernelGGL(vox_init_kernel, dim3(cdiv((long long)((total + 3) / 4), 256)), dim3(256), 0, stream, keys, n_keys,
                       first, n_first, count, n_count, ptlist, n_pt, misc);
  }
  int nb = cdiv(num_points, 256);
  hipLaunchKernelGGL(vox_insert_kernel, dim3(nb), dim3(256), 0, stream, points, p, keys, first, slot);
  hipLaunchKernelGGL(vox_flag_kernel, dim3(nb), dim3(256), 0, stream, num_points, first, slot, rank);
  int rc = exclusive_scan_u32(rank, rank, (size_t)num_points, (uint32_t *)misc, sscr, ssz, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(vox_number_kernel, dim3(nb), dim3(256), 0, stream, p, keys, first, slot, rank, vid, coors, misc,
                     voxel_num);
  hipLaunchKernelGGL(vox_assign_kernel, dim3(nb), dim3(256), 0, stream, p, break_at_cap, slot, vid, misc, count,
                     ptlist);
  hipLaunchKernelGGL(vox_gather_kernel, dim3(cdiv(max_voxels, 256)), dim3(256), 0, stream, points, p, misc, count,
                     ptlist, voxels, num_points_per_voxel, mean);
  DF3D_LAUNCH_CHECK();
  return DF3D_OK;
}
