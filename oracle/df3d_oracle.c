/*
 * oracle/df3d_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the integer/index algorithms on the 3D-Dual-Fusion
 * hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the product path (3d-dual-fusion_amd/) never does.
 *
 * Every function cites the reference file:line whose behaviour it restates
 * (TF/ = /root/reference/TransFusion, CP/ = /root/reference/CenterPoint).
 * The restatement is pinned against the reference's own compiled CPU code
 * (oracle/_ref, see oracle/build_ref.py) and against the golden vectors in
 * tests/golden/ (tests/test_oracle_*.py).
 *
 * Build: gcc -O2 -shared -fPIC -o oracle/libdf3d_oracle.so oracle/df3d_oracle.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------
 * hard voxelisation.
 * Restates TF/mmdet3d/ops/voxel/src/voxelization_cpu.cpp:7-41 (dynamic_voxelize_kernel:
 * c = floor((p - min) / vs), reject c<0 || c>=grid, coords stored reversed -> (z,y,x))
 * and :43-102 (hard_voxelize_kernel: dense coor_to_voxelidx grid, first-appearance
 * voxel numbering, `break` when a NEW voxel would exceed max_voxels, at most
 * max_points points per voxel kept in arrival order) and :105-141 (grid_size =
 * round((max-min)/vs)).
 * points [P,C] f32; voxels [max_voxels,max_points,C] (caller zero-filled);
 * coors [max_voxels,3] i32 (z,y,x); num [max_voxels] i32 (caller zero-filled).
 * Returns voxel_num.
 * ---------------------------------------------------------------------- */
int orc_hard_voxelize(const float *points, int P, int C, const float *voxel_size,
                      const float *range, int max_points, int max_voxels,
                      float *voxels, int32_t *coors, int32_t *num) {
  int grid[3];
  for (int i = 0; i < 3; ++i)
    grid[i] = (int)roundf((range[3 + i] - range[i]) / voxel_size[i]);
  size_t vol = (size_t)grid[0] * grid[1] * grid[2];
  int32_t *c2v = (int32_t *)malloc(vol * sizeof(int32_t));
  if (!c2v) return -1;
  memset(c2v, 0xff, vol * sizeof(int32_t)); /* -1 */
  int voxel_num = 0;
  for (int i = 0; i < P; ++i) {
    int coor[3];
    int failed = 0;
    for (int j = 0; j < 3; ++j) {
      /* float arithmetic exactly as the reference: (float - float) / float, then floor */
      int c = (int)floor((points[(size_t)i * C + j] - range[j]) / voxel_size[j]);
      if (c < 0 || c >= grid[j]) { failed = 1; break; }
      coor[2 - j] = c;
    }
    if (failed) continue;
    size_t lin = ((size_t)coor[0] * grid[1] + coor[1]) * grid[0] + coor[2];
    int vid = c2v[lin];
    if (vid == -1) {
      vid = voxel_num;
      if (max_voxels != -1 && voxel_num >= max_voxels) break;
      voxel_num += 1;
      c2v[lin] = vid;
      for (int k = 0; k < 3; ++k) coors[vid * 3 + k] = coor[k];
    }
    int n = num[vid];
    if (max_points == -1 || n < max_points) {
      memcpy(voxels + ((size_t)vid * max_points + n) * C, points + (size_t)i * C, sizeof(float) * C);
      num[vid] += 1;
    }
  }
  free(c2v);
  return voxel_num;
}

/* numba variant: CP/det3d/ops/point_cloud/point_cloud_ops.py:7-55
 * (_points_to_voxel_reverse_kernel).  Identical except that at the max_voxels cap
 * it `continue`s (later points may still join EXISTING voxels) instead of
 * breaking the loop, and grid_size is computed as round((max-min)/vs) in numpy. */
int orc_points_to_voxel_numba(const float *points, int P, int C, const float *voxel_size,
                              const float *range, int max_points, int max_voxels,
                              float *voxels, int32_t *coors, int32_t *num) {
  int grid[3];
  for (int i = 0; i < 3; ++i)
    grid[i] = (int)roundf((range[3 + i] - range[i]) / voxel_size[i]);
  size_t vol = (size_t)grid[0] * grid[1] * grid[2];
  int32_t *c2v = (int32_t *)malloc(vol * sizeof(int32_t));
  if (!c2v) return -1;
  memset(c2v, 0xff, vol * sizeof(int32_t));
  int voxel_num = 0;
  for (int i = 0; i < P; ++i) {
    int coor[3];
    int failed = 0;
    for (int j = 0; j < 3; ++j) {
      int c = (int)floor((points[(size_t)i * C + j] - range[j]) / voxel_size[j]);
      if (c < 0 || c >= grid[j]) { failed = 1; break; }
      coor[2 - j] = c;
    }
    if (failed) continue;
    size_t lin = ((size_t)coor[0] * grid[1] + coor[1]) * grid[0] + coor[2];
    int vid = c2v[lin];
    if (vid == -1) {
      vid = voxel_num;
      if (voxel_num >= max_voxels) continue;
      voxel_num += 1;
      c2v[lin] = vid;
      for (int k = 0; k < 3; ++k) coors[vid * 3 + k] = coor[k];
    }
    int n = num[vid];
    if (n < max_points) {
      memcpy(voxels + ((size_t)vid * max_points + n) * C, points + (size_t)i * C, sizeof(float) * C);
      num[vid] += 1;
    }
  }
  free(c2v);
  return voxel_num;
}

/* ------------------------------------------------------------------------
 * rulebook (indice pairs), NDim = 3.
 * getValidOutPos restates TF/mmdet3d/ops/spconv/include/spconv/geometry.h:24-85:
 * per axis lowers=(in-(k-1)d-1+s+p)/s, uppers=(in+p)/s (C integer division),
 * enumerate val=uppers-counter*d with the LAST axis fastest, kernel offset
 * = sum_j m_j*(in_j - val_j*s_j + p_j)/d_j with m row-major over (z,y,x).
 * out: [<=K][4] = (z,y,x,offset) of the valid positions; returns their count.
 * ---------------------------------------------------------------------- */
static int get_valid_out_pos(const int *in, const int *ks, const int *st, const int *pad,
                             const int *dil, const int *oshape, int *out) {
  int lowers[3], uppers[3], counter[3], csize[3];
  int npts = 1, cnt = 0;
  for (int i = 0; i < 3; ++i) {
    lowers[i] = (in[i] - (ks[i] - 1) * dil[i] - 1 + st[i] + pad[i]) / st[i];
    uppers[i] = (in[i] + pad[i]) / st[i];
  }
  for (int i = 0; i < 3; ++i) {
    csize[i] = (uppers[i] - lowers[i]) / dil[i] + 1;
    npts *= csize[i];
    counter[i] = 0;
  }
  for (int i = 0; i < npts; ++i) {
    int valid = 1, m = 1, offset = 0;
    for (int j = 2; j >= 0; --j) {
      int val = uppers[j] - counter[j] * dil[j];
      out[cnt * 4 + j] = val;
      if (val < 0 || val > oshape[j] - 1) valid = 0;
      offset += m * (in[j] - val * st[j] + pad[j]) / dil[j];
      m *= ks[j];
    }
    out[cnt * 4 + 3] = offset;
    if (valid) ++cnt;
    counter[2] += 1;
    for (int c = 2; c >= 0; --c) {
      if (counter[c] == csize[c] && c > 0) {
        counter[c - 1] += 1;
        counter[c] = 0;
      }
    }
  }
  return cnt;
}

/* getValidOutPosTranspose restates geometry.h:88-142 (transposed convolution): per axis lowers = in*s - p,
 * uppers = lowers + (k-1)*d; enumerate val = uppers - counter*d with the LAST axis fastest; kernel offset =
 * sum_j m_j*(val_j - lowers_j)/d_j, i.e. kernel index c writes the output cell in*s - p + c*d. */
static int get_valid_out_pos_transpose(const int *in, const int *ks, const int *st, const int *pad,
                                       const int *dil, const int *oshape, int *out) {
  int lowers[3], uppers[3], counter[3], csize[3];
  int npts = 1, cnt = 0;
  for (int i = 0; i < 3; ++i) {
    lowers[i] = in[i] * st[i] - pad[i];
    uppers[i] = lowers[i] + (ks[i] - 1) * dil[i];
  }
  for (int i = 0; i < 3; ++i) {
    csize[i] = (uppers[i] - lowers[i]) / dil[i] + 1;
    npts *= csize[i];
    counter[i] = 0;
  }
  for (int i = 0; i < npts; ++i) {
    int valid = 1, m = 1, offset = 0;
    for (int j = 2; j >= 0; --j) {
      int val = uppers[j] - counter[j] * dil[j];
      out[cnt * 4 + j] = val;
      if (val < 0 || val > oshape[j] - 1) valid = 0;
      offset += m * (val - lowers[j]) / dil[j];
      m *= ks[j];
    }
    out[cnt * 4 + 3] = offset;
    if (valid) ++cnt;
    counter[2] += 1;
    for (int c = 2; c >= 0; --c) {
      if (counter[c] == csize[c] && c > 0) {
        counter[c - 1] += 1;
        counter[c] = 0;
      }
    }
  }
  return cnt;
}

/* getIndicePairsDeConv (geometry.h:194-245): the loop of getIndicePairsConv over getValidOutPosTranspose.
 * out_shape = (in-1)*s - 2p + k + output_padding (ops.py:33-44).  Same conventions and order as below. */
int orc_get_indice_pairs_transpose(const int32_t *indices, int N, int batch, const int *out_shape,
                                   const int *ksize, const int *stride, const int *padding,
                                   const int *dilation, int32_t *outids, int32_t *pairs, int32_t *num) {
  int K = ksize[0] * ksize[1] * ksize[2];
  size_t vol = (size_t)out_shape[0] * out_shape[1] * out_shape[2];
  int32_t *grid = (int32_t *)malloc(vol * batch * sizeof(int32_t));
  if (!grid) return -1;
  memset(grid, 0xff, vol * batch * sizeof(int32_t));
  for (size_t i = 0; i < (size_t)K * 2 * N; ++i) pairs[i] = -1;
  for (int k = 0; k < K; ++k) num[k] = 0;
  int *vp = (int *)malloc(sizeof(int) * K * 4);
  int numAct = 0;
  for (int j = 0; j < N; ++j) {
    const int32_t *p = indices + (size_t)j * 4;
    int nv = get_valid_out_pos_transpose(p + 1, ksize, stride, padding, dilation, out_shape, vp);
    for (int i = 0; i < nv; ++i) {
      const int *q = vp + i * 4;
      int off = q[3];
      size_t idx = ((size_t)q[0] * out_shape[1] + q[1]) * out_shape[2] + q[2] + vol * p[0];
      if (grid[idx] == -1) {
        outids[(size_t)numAct * 4 + 0] = p[0];
        outids[(size_t)numAct * 4 + 1] = q[0];
        outids[(size_t)numAct * 4 + 2] = q[1];
        outids[(size_t)numAct * 4 + 3] = q[2];
        grid[idx] = numAct++;
      }
      pairs[((size_t)off * 2 + 0) * N + num[off]] = j;
      pairs[((size_t)off * 2 + 1) * N + num[off]] = grid[idx];
      num[off] += 1;
    }
  }
  free(vp);
  free(grid);
  return numAct;
}

/* getIndicePairsSubM (geometry.h:247-297) / getIndicePairsConv (geometry.h:144-192),
 * with the host-side conventions of getIndicePair<3> (spconv_ops.h:27-141):
 * subM forces stride 1, pad k/2 (:76-79); indicePairs [K,2,N] prefilled -1,
 * indiceNum [K] zero, dense grid [B*vol] prefilled -1.
 * indices [N,4] = (b,z,y,x).  outids [N*K,4] (only written for !subm).
 * Order = the reference CPU order: out voxels numbered by first touch, pairs per
 * offset in input order.  Returns numActOut (N for subm). */
int orc_get_indice_pairs(const int32_t *indices, int N, int batch, const int *out_shape,
                         const int *ksize, const int *stride, const int *padding,
                         const int *dilation, int subm, int32_t *outids, int32_t *pairs,
                         int32_t *num) {
  int K = ksize[0] * ksize[1] * ksize[2];
  int st[3], pad[3], dil[3];
  for (int i = 0; i < 3; ++i) {
    st[i] = subm ? 1 : stride[i];
    pad[i] = subm ? ksize[i] / 2 : padding[i];
    dil[i] = dilation[i];
  }
  size_t vol = (size_t)out_shape[0] * out_shape[1] * out_shape[2];
  int32_t *grid = (int32_t *)malloc(vol * batch * sizeof(int32_t));
  if (!grid) return -1;
  memset(grid, 0xff, vol * batch * sizeof(int32_t));
  for (size_t i = 0; i < (size_t)K * 2 * N; ++i) pairs[i] = -1;
  for (int k = 0; k < K; ++k) num[k] = 0;
  int *vp = (int *)malloc(sizeof(int) * K * 4);
  int numAct = 0;
  if (subm) {
    for (int j = 0; j < N; ++j) {
      const int32_t *p = indices + (size_t)j * 4;
      size_t idx = ((size_t)p[1] * out_shape[1] + p[2]) * out_shape[2] + p[3] + vol * p[0];
      grid[idx] = j;
    }
    for (int j = 0; j < N; ++j) {
      const int32_t *p = indices + (size_t)j * 4;
      int nv = get_valid_out_pos(p + 1, ksize, st, pad, dil, out_shape, vp);
      for (int i = 0; i < nv; ++i) {
        const int *q = vp + i * 4;
        int off = q[3];
        size_t idx = ((size_t)q[0] * out_shape[1] + q[1]) * out_shape[2] + q[2] + vol * p[0];
        if (grid[idx] > -1) {
          pairs[((size_t)off * 2 + 0) * N + num[off]] = j;
          pairs[((size_t)off * 2 + 1) * N + num[off]] = grid[idx];
          num[off] += 1;
        }
      }
    }
    numAct = N;
  } else {
    for (int j = 0; j < N; ++j) {
      const int32_t *p = indices + (size_t)j * 4;
      int nv = get_valid_out_pos(p + 1, ksize, st, pad, dil, out_shape, vp);
      for (int i = 0; i < nv; ++i) {
        const int *q = vp + i * 4;
        int off = q[3];
        size_t idx = ((size_t)q[0] * out_shape[1] + q[1]) * out_shape[2] + q[2] + vol * p[0];
        if (grid[idx] == -1) {
          outids[(size_t)numAct * 4 + 0] = p[0];
          outids[(size_t)numAct * 4 + 1] = q[0];
          outids[(size_t)numAct * 4 + 2] = q[1];
          outids[(size_t)numAct * 4 + 3] = q[2];
          grid[idx] = numAct++;
        }
        pairs[((size_t)off * 2 + 0) * N + num[off]] = j;
        pairs[((size_t)off * 2 + 1) * N + num[off]] = grid[idx];
        num[off] += 1;
      }
    }
  }
  free(vp);
  free(grid);
  return numAct;
}

/* ------------------------------------------------------------------------
 * point ops used by LocalTransformer (reference has CUDA only; restated from the
 * kernels, see SURVEY.md §8c "LocalTransformer").
 * ---------------------------------------------------------------------- */

/* D-FPS.  CP/det3d/ops/furthest_point_sample/src/furthest_point_sample_cuda.cu:26-141
 * (kernel) and :9-15,143-205 (block size = largest power of two <= N, capped at 1024,
 * computed as int(log(N)/log(2)) in double).  Start at index 0; temp[k] (caller's 1e10
 * fill, furthest_point_sample.py:31) = min(temp[k], d2(k,last)); next = argmax temp.
 * Tie rule = what the block reduction yields deterministically: each thread keeps
 * the LOWEST k of its strided sequence (strict '>', :66-67) and the tree reduction
 * keeps the LOWER tid on ties (__update :17-24), i.e. among maxima the winner has
 * the smallest (k mod block, k).  xyz [B,N,3]; idx out [B,m]. */
/* The squared distance of both kernels as the reference's GPU build evaluates it: nvcc contracts dx*dx + dy*dy + dz*dz
 * (--fmad=true by default) into fma(dz, dz, fma(dy, dy, dx*dx)).  On lattice points (voxel centres) exact ties between
 * candidates are common, and the last bit of this sum decides the pick; this file is compiled with -ffp-contract=off, so
 * the contraction is written out (csrc/pointops.hip does the same). */
static float orc_dist2(float dx, float dy, float dz) { return fmaf(dz, dz, fmaf(dy, dy, dx * dx)); }

int orc_fps_block(int n) {
  int pow_2 = (int)(log((double)n) / log(2.0));
  int t = 1 << pow_2;
  if (t > 1024) t = 1024;
  if (t < 1) t = 1;
  return t;
}

void orc_fps(const float *xyz, int B, int N, int m, int32_t *idx) {
  float *temp = (float *)malloc(sizeof(float) * N);
  int bs = orc_fps_block(N);
  for (int b = 0; b < B; ++b) {
    const float *p = xyz + (size_t)b * N * 3;
    int32_t *o = idx + (size_t)b * m;
    for (int k = 0; k < N; ++k) temp[k] = 1e10f;
    int old = 0;
    if (m > 0) o[0] = 0;
    for (int j = 1; j < m; ++j) {
      float x1 = p[old * 3], y1 = p[old * 3 + 1], z1 = p[old * 3 + 2];
      float best = -1.f;
      int besti = 0;
      for (int k = 0; k < N; ++k) {
        float x2 = p[k * 3], y2 = p[k * 3 + 1], z2 = p[k * 3 + 2];
        float d = orc_dist2(x2 - x1, y2 - y1, z2 - z1);
        float d2 = d < temp[k] ? d : temp[k];
        temp[k] = d2;
        if (d2 > best || (d2 == best && (k % bs) < (besti % bs))) { best = d2; besti = k; }
      }
      old = besti;
      o[j] = old;
    }
  }
  free(temp);
}

/* ball query.  CP/det3d/ops/ball_query/src/ball_query_cuda.cu:11-54: for each centre
 * scan points in index order; accept if d2 == 0 || (d2 >= min_r2 && d2 < max_r2); on the first
 * hit fill all nsample slots with it; stop after nsample hits.  idx [B,m,nsample]
 * (zero-initialised by the caller, as the reference does). */
void orc_ball_query(const float *new_xyz, const float *xyz, int B, int N, int m,
                    float min_radius, float max_radius, int nsample, int32_t *idx) {
  float max_r2 = max_radius * max_radius, min_r2 = min_radius * min_radius;
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < m; ++c) {
      const float *q = new_xyz + ((size_t)b * m + c) * 3;
      int32_t *o = idx + ((size_t)b * m + c) * nsample;
      int cnt = 0;
      for (int k = 0; k < N && cnt < nsample; ++k) {
        const float *p = xyz + ((size_t)b * N + k) * 3;
        float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
        float d2 = orc_dist2(dx, dy, dz);
        if (d2 == 0 || (d2 >= min_r2 && d2 < max_r2)) {
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) o[l] = k;
          o[cnt] = k;
          ++cnt;
        }
      }
    }
}

/* ------------------------------------------------------------------------
 * rotated BEV overlap / IoU / NMS (detection tail).
 * Restates CP/det3d/ops/iou3d_nms/src/iou3d_cpu.cpp:58-221 (the reference's CPU path; its GPU kernel
 * iou3d_nms_kernel.cu:36-245 is the same arithmetic): boxes are [x, y, z, dx, dy, dz, heading]; the overlap polygon
 * is assembled from (i) the proper crossings of the 4x4 edge pairs (bounding-rectangle rejection, strict
 * cross-product sign test, line intersection with the EPS = 1e-8 fallback formula), (ii) the corners of either
 * box inside the other with a 1e-2 margin (corner of b tested before corner of a for each k), ordered by a bubble
 * sort on atan2 around the vertex mean, and its area is the fan of cross products around vertex 0.
 * All arithmetic in float like the reference (cosf/sinf/atan2f are what the C++ float overloads resolve to). */
typedef struct { float x, y; } orc_pt;

static float orc_cross3(orc_pt p1, orc_pt p2, orc_pt p0) {
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

static int orc_rect_cross(orc_pt p1, orc_pt p2, orc_pt q1, orc_pt q2) {
  return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
         fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

static int orc_in_box2d(const float *box, orc_pt p) { /* iou3d_cpu.cpp:73-83 */
  const float margin = 1e-2f;
  float ca = cosf(-box[6]), sa = sinf(-box[6]);
  float rx = (p.x - box[0]) * ca + (p.y - box[1]) * (-sa);
  float ry = (p.x - box[0]) * sa + (p.y - box[1]) * ca;
  return fabsf(rx) < box[3] / 2 + margin && fabsf(ry) < box[4] / 2 + margin;
}

static int orc_intersection(orc_pt p1, orc_pt p0, orc_pt q1, orc_pt q0, orc_pt *ans) { /* :85-115 */
  if (!orc_rect_cross(p0, p1, q0, q1)) return 0;
  float s1 = orc_cross3(q0, p1, p0), s2 = orc_cross3(p1, q1, p0);
  float s3 = orc_cross3(p0, q1, q0), s4 = orc_cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
  float s5 = orc_cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > 1e-8f) {
    ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float D = a0 * b1 - a1 * b0;
    ans->x = (b0 * c1 - b1 * c0) / D;
    ans->y = (a1 * c0 - a0 * c1) / D;
  }
  return 1;
}

static void orc_corners(const float *box, orc_pt *c) { /* :131-163 */
  float hx = box[3] / 2, hy = box[4] / 2, ca = cosf(box[6]), sa = sinf(box[6]);
  float x1 = box[0] - hx, y1 = box[1] - hy, x2 = box[0] + hx, y2 = box[1] + hy;
  float px[4] = {x1, x2, x2, x1}, py[4] = {y1, y1, y2, y2};
  for (int k = 0; k < 4; ++k) {
    c[k].x = (px[k] - box[0]) * ca + (py[k] - box[1]) * (-sa) + box[0];
    c[k].y = (px[k] - box[0]) * sa + (py[k] - box[1]) * ca + box[1];
  }
  c[4] = c[0];
}

float orc_box_overlap(const float *a, const float *b) { /* :125-212 */
  orc_pt ca[5], cb[5], pts[16], ctr = {0.f, 0.f};
  int cnt = 0;
  orc_corners(a, ca);
  orc_corners(b, cb);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (orc_intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], &pts[cnt])) {
        ctr.x += pts[cnt].x;
        ctr.y += pts[cnt].y;
        cnt++;
      }
  for (int k = 0; k < 4; ++k) {
    if (orc_in_box2d(a, cb[k])) {
      ctr.x += cb[k].x;
      ctr.y += cb[k].y;
      pts[cnt++] = cb[k];
    }
    if (orc_in_box2d(b, ca[k])) {
      ctr.x += ca[k].x;
      ctr.y += ca[k].y;
      pts[cnt++] = ca[k];
    }
  }
  ctr.x /= cnt;
  ctr.y /= cnt;
  for (int j = 0; j < cnt - 1; ++j)
    for (int i = 0; i < cnt - j - 1; ++i)
      if (atan2f(pts[i].y - ctr.y, pts[i].x - ctr.x) > atan2f(pts[i + 1].y - ctr.y, pts[i + 1].x - ctr.x)) {
        orc_pt t = pts[i];
        pts[i] = pts[i + 1];
        pts[i + 1] = t;
      }
  float area = 0;
  for (int k = 0; k < cnt - 1; ++k) {
    float ax = pts[k].x - pts[0].x, ay = pts[k].y - pts[0].y;
    float bx = pts[k + 1].x - pts[0].x, by = pts[k + 1].y - pts[0].y;
    area += ax * by - ay * bx;
  }
  return fabsf(area) / 2.0f;
}

float orc_iou_bev(const float *a, const float *b) { /* :214-221 */
  float sa = a[3] * a[4], sb = b[3] * b[4], so = orc_box_overlap(a, b);
  return so / fmaxf(sa + sb - so, 1e-8f);
}

static float orc_iou_normal(const float *a, const float *b) { /* iou3d_nms_kernel.cu:309-320 */
  float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f), inter = w * h;
  return inter / fmaxf(a[3] * a[4] + b[3] * b[4] - inter, 1e-8f);
}

/* mode 0: overlap area, 1: rotated IoU (boxes_overlap_bev_gpu / boxes_iou_bev_gpu, iou3d_nms.cpp:38-85) */
void orc_boxes_pairwise(const float *a, int na, const float *b, int nb, int mode, float *out) {
  for (int i = 0; i < na; ++i)
    for (int j = 0; j < nb; ++j)
      out[(size_t)i * nb + j] = mode ? orc_iou_bev(a + i * 7, b + j * 7) : orc_box_overlap(a + i * 7, b + j * 7);
}

/* nms_gpu / nms_normal_gpu (iou3d_nms.cpp:88-139,142-188 + nms_kernel iou3d_nms_kernel.cu:262-306): boxes are
 * already sorted by descending score; box i is kept unless an earlier kept box j < i has IoU(j, i) > thresh
 * (the bit matrix is evaluated as iou(row, col) with col > row).  Returns the number kept; keep[] = indices. */
int orc_nms_bev(const float *boxes, int n, float thresh, int rotated, int64_t *keep, float margin,
                int *n_close) {
  unsigned char *removed = (unsigned char *)calloc((size_t)n > 0 ? n : 1, 1);
  int nk = 0, close = 0;
  for (int i = 0; i < n; ++i) {
    if (removed[i]) continue;
    keep[nk++] = i;
    for (int j = i + 1; j < n; ++j) {
      if (rotated == 2) { /* circle NMS, CP/det3d/core/utils/circle_nms_jit.py:4-27: squared centre distance <= thresh */
        float dx = boxes[i * 7] - boxes[j * 7], dy = boxes[i * 7 + 1] - boxes[j * 7 + 1];
        if (dx * dx + dy * dy <= thresh) removed[j] = 1;
        continue;
      }
      float v = rotated ? orc_iou_bev(boxes + i * 7, boxes + j * 7) : orc_iou_normal(boxes + i * 7, boxes + j * 7);
      if (fabsf(v - thresh) < margin) close++;     /* decisions a last-ulp difference of cos/sin/atan2 could flip */
      if (v > thresh) removed[j] = 1;
    }
  }
  free(removed);
  if (n_close) *n_close = close;
  return nk;
}

/* ------------------------------------------------------------------------------------------------------------
 * TransFusion tree's BEV overlap (TF/mmdet3d/ops/iou3d/src/iou3d_kernel.cu:56-240): boxes are
 * [x1, y1, x2, y2, angle]; corners are turned by -angle about the box centre (rotate_around_center :106-114 with
 * cos(angle), sin(angle)), the containment test turns the point back (check_in_box2d :56-79, margin 1e-5).
 * The reference has no CPU build of this kernel: this restatement is pinned on the GPU box against the reference's
 * own kernel (oracle/_ref/iou3d_cuda_tf.so, tests/test_gpu_tfloss.py). */
static int orc_tf_in_box2d(const float *box, orc_pt p) {
  const float margin = 1e-5f;
  float cx = (box[0] + box[2]) / 2, cy = (box[1] + box[3]) / 2;
  float ca = cosf(-box[4]), sa = sinf(-box[4]);
  float rx = (p.x - cx) * ca + (p.y - cy) * sa + cx;
  float ry = -(p.x - cx) * sa + (p.y - cy) * ca + cy;
  return rx > box[0] - margin && rx < box[2] + margin && ry > box[1] - margin && ry < box[3] + margin;
}

static void orc_tf_corners(const float *box, orc_pt *c) {
  float cx = (box[0] + box[2]) / 2, cy = (box[1] + box[3]) / 2, ca = cosf(box[4]), sa = sinf(box[4]);
  float px[4] = {box[0], box[2], box[2], box[0]}, py[4] = {box[1], box[1], box[3], box[3]};
  for (int k = 0; k < 4; ++k) {
    c[k].x = (px[k] - cx) * ca + (py[k] - cy) * sa + cx;
    c[k].y = -(px[k] - cx) * sa + (py[k] - cy) * ca + cy;
  }
  c[4] = c[0];
}

float orc_tf_box_overlap(const float *a, const float *b) { /* iou3d_kernel.cu:122-230 */
  orc_pt ca[5], cb[5], pts[16], ctr = {0.f, 0.f};
  int cnt = 0;
  orc_tf_corners(a, ca);
  orc_tf_corners(b, cb);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (orc_intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], &pts[cnt])) {
        ctr.x += pts[cnt].x;
        ctr.y += pts[cnt].y;
        cnt++;
      }
  for (int k = 0; k < 4; ++k) {
    if (orc_tf_in_box2d(a, cb[k])) {
      ctr.x += cb[k].x;
      ctr.y += cb[k].y;
      pts[cnt++] = cb[k];
    }
    if (orc_tf_in_box2d(b, ca[k])) {
      ctr.x += ca[k].x;
      ctr.y += ca[k].y;
      pts[cnt++] = ca[k];
    }
  }
  ctr.x /= cnt;
  ctr.y /= cnt;
  for (int j = 0; j < cnt - 1; ++j)
    for (int i = 0; i < cnt - j - 1; ++i)
      if (atan2f(pts[i].y - ctr.y, pts[i].x - ctr.x) > atan2f(pts[i + 1].y - ctr.y, pts[i + 1].x - ctr.x)) {
        orc_pt t = pts[i];
        pts[i] = pts[i + 1];
        pts[i + 1] = t;
      }
  float area = 0;
  for (int k = 0; k < cnt - 1; ++k) {
    float ax = pts[k].x - pts[0].x, ay = pts[k].y - pts[0].y;
    float bx = pts[k + 1].x - pts[0].x, by = pts[k + 1].y - pts[0].y;
    area += ax * by - ay * bx;
  }
  return fabsf(area) / 2.0f;
}

/* boxes_overlap_bev_gpu (iou3d.cpp:66-90): out[i, j] = overlap area of a[i] and b[j], boxes [n, 5] xyxyr */
void orc_tf_boxes_overlap_bev(const float *a, int na, const float *b, int nb, float *out) {
  for (int i = 0; i < na; ++i)
    for (int j = 0; j < nb; ++j) out[(size_t)i * nb + j] = orc_tf_box_overlap(a + i * 5, b + j * 5);
}
