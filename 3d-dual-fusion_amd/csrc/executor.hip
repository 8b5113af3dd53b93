// Native executor for a chain of fused sparse-convolution layers (the sparse 3-D backbones).
//
// The reference walks its backbone layer by layer from Python: per layer a rulebook lookup / build, a D2H read
// of the output count for strided layers, gather/GEMM/scatter launches and separate BN / ReLU / residual passes
// (CP/det3d/models/backbones/scn.py:97-201, TF/mmdet3d/ops/spconv/conv.py:124-204).  With the fused kernels of
// this library a layer is one or two launches of 20-90 us, and the HOST became the bottleneck: ~40 us of
// interpreter work per layer against ~20 us of GPU work (tools/cpu_profile.py).  This executor runs the whole
// chain from one C call: the layer table is built once from the module tree, every intermediate (features,
// split rows, neighbour tables, occupancy directories) comes from one caller-provided arena (bump allocation,
// nothing is freed inside a run), and the only host round trips left are the output counts of the strided
// layers.  The kernels are exactly the ones the per-layer API launches (the extern "C" entry points below call
// the same functions), so results are bit-identical to the module path.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include <mutex>

#include "common.h"

using namespace df3d;

namespace {

struct IndexSet {
  const int32_t *indices = nullptr;
  int n = 0;
  int shape[3] = {0, 0, 0};
  void *grid = nullptr;        // occupancy directory (built lazily)
  size_t grid_bytes = 0;
  int32_t *perm = nullptr;     // rank -> row (NULL when the rows are sorted by flat index)
  bool sorted = false;
};

struct LayerOut {
  float *features = nullptr;
  void *split = nullptr;       // split rows (bf16 hi | lo) of `features`, or its bf16 rows when the layer ran in bf16
  void *rows16 = nullptr;      // bf16 rows of `features` (bf16 layers; converted on demand for fp32 producers)
  int set = -1;
  int channels = 0;
  bool ran = false;            // the convolution of this layer was launched (false: geometry-only layer)
};

struct Bump {
  char *base;
  size_t cap, used = 0;
  bool overflow = false;
  Bump(void *p, size_t bytes) : base((char *)p), cap(bytes) {}
  void *take(size_t bytes) {
    size_t off = align_up(used, 256);
    used = off + bytes;
    if (used > cap) {
      overflow = true;
      return nullptr;
    }
    return base + off;
  }
};

int kvol_of(const int *k) { return k[0] * k[1] * k[2]; }

// Second stream for the geometry work + a small pool of ordering events, created once per HOST THREAD: frames are
// independent, and a caller that keeps several frames in flight drives each from its own thread and stream.
thread_local hipStream_t g_geo_stream = nullptr;
// Sixteen event sets, one per run in rotation: with df3d_backbone_inputs_ready the geometry of run k + 1 records its events while
// the caller's stream may not yet have reached the waits it queued for run k's events -- a re-recorded event must not be one a
// pending wait refers to.
thread_local std::vector<hipEvent_t> g_event_sets[16];
thread_local unsigned g_run_counter = 0;
#define g_events g_event_sets[g_run_counter & 15]
thread_local hipEvent_t g_inputs_ready = nullptr;     // df3d_backbone_inputs_ready: consumed by the next df3d_backbone_run
hipEvent_t order_event(size_t i) {
  while (g_events.size() <= i) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    g_events.push_back(e);
  }
  return g_events[i];
}

}  // namespace

extern "C" int df3d_backbone_inputs_ready(void *event) {
  g_inputs_ready = (hipEvent_t)event;
  return DF3D_OK;
}

extern "C" int df3d_backbone_run(const df3d_layer *layers, int nlayers, const float *features, const int32_t *indices,
                                 int n, int in_channels, int batch, const int *shape, void *arena,
                                 size_t arena_bytes, df3d_layer_view *views, size_t *arena_used, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  hipEvent_t inputs_ready = g_inputs_ready;
  g_inputs_ready = nullptr;
  ++g_run_counter;
  DF3D_CHECK_ARG(layers && nlayers > 0 && features && indices && shape && arena && views,
                 "backbone_run: null argument");
  DF3D_CHECK_ARG(n > 0 && batch > 0, "backbone_run: empty input (n=%d, batch=%d)", n, batch);
  // Geometry (rulebooks, directories, the host round trips for the output counts) runs on its own stream and only
  // ever waits for itself; the convolutions run on the caller's stream and wait, per rulebook, for the event that
  // marks its neighbour table complete.  Index sets depend on coordinates alone, so the geometry of later stages
  // proceeds while the convolutions of earlier stages keep the GPU busy.  DF3D_EXEC_STREAMS=0: one stream.
  static const bool two_streams = !(getenv("DF3D_EXEC_STREAMS") && getenv("DF3D_EXEC_STREAMS")[0] == '0');
  hipStream_t gstream = stream;
  size_t next_event = 0;
  if (two_streams) {
    if (!g_geo_stream) DF3D_HIP(hipStreamCreateWithFlags(&g_geo_stream, hipStreamNonBlocking));
    gstream = g_geo_stream;
    if (inputs_ready) {
      // the caller vouches that the COORDINATES are complete at this event (df3d_backbone_inputs_ready) and that the arena is
      // not in use by earlier work of `stream`: the geometry then does not wait for whatever the caller's stream still has
      // queued (the previous frame), and the host passes its round trips while that frame is still running
      DF3D_HIP(hipStreamWaitEvent(gstream, inputs_ready, 0));
    } else {
      hipEvent_t e = order_event(next_event++);
      DF3D_CHECK_ARG(e != nullptr, "backbone_run: cannot create an event");
      DF3D_HIP(hipEventRecord(e, stream));               // inputs (voxel features / coordinates) are ready
      DF3D_HIP(hipStreamWaitEvent(gstream, e, 0));
    }
  }
  void *gs_ = (void *)gstream;
  Bump mem(arena, arena_bytes);
  std::vector<IndexSet> sets;
  std::vector<LayerOut> outs(nlayers);
  std::vector<std::pair<int, const int32_t *>> rulebooks;     // (rulebook id, nbr table)
  std::vector<int> rulebook_set;                              // index set of the rulebook's outputs
  IndexSet s0;
  s0.indices = indices;
  s0.n = n;
  memcpy(s0.shape, shape, sizeof(s0.shape));
  sets.push_back(s0);
  void *split0 = nullptr;      // split rows of the network input, built on demand
  void *rows16_0 = nullptr;    // its bf16 rows (bf16 layers)
#define DF3D_ARENA_CHECK(ok)                                                                                   \
  do {                                                                                                          \
    if (!(ok)) {                                                                                                \
      if (arena_used) *arena_used = mem.used;                                                                   \
      set_error("backbone_run: arena of %zu bytes is too small (needed more than %zu)", arena_bytes, mem.used); \
      return DF3D_ENOMEM;                                                                                       \
    }                                                                                                           \
  } while (0)
  int32_t *count_dev = (int32_t *)mem.take(256);
  DF3D_ARENA_CHECK(count_dev);

  // ---- per layer: geometry (neighbour table; for strided layers the output index set, with the one host round
  //      trip for its size) on the geometry stream, then the fused convolution on the caller's stream ----
  std::vector<const int32_t *> layer_nbr(nlayers, nullptr);
  // Which layers must emit fp32 rows next to their split rows: exported stages (flag bit 2, set by the caller), residual
  // sources, inputs of layers that do not run on the split-precision kernels.  The first convolution of every residual block
  // feeds exactly one split-precision convolution: its fp32 rows were written and never read (round 2: every output twice).
  std::vector<char> need_f32(nlayers, 0);
  {
    bool flagged = false;
    for (int li = 0; li < nlayers; ++li) flagged = flagged || (layers[li].reserved & 4);
    static const bool write_all = getenv("DF3D_EXEC_F32_ALL") && atoi(getenv("DF3D_EXEC_F32_ALL"));   // A / B of the PMC passes
    if (write_all) flagged = false;
    for (int li = 0; li < nlayers; ++li) {
      const df3d_layer &L = layers[li];
      if (!flagged || (L.reserved & 4) || li == nlayers - 1) need_f32[li] = 1;
      if (L.residual >= 0 && L.residual < nlayers) need_f32[L.residual] = 1;
      const bool split_consumer = L.packed && !(L.reserved & 2) && !(L.reserved & 1) &&
                                  df3d_conv_packed_weight_bytes(kvol_of(L.ksize), L.cin, L.cout) != 0;
      if (L.input >= 0 && L.input < nlayers && !split_consumer) need_f32[L.input] = 1;
    }
  }
  for (int li = 0; li < nlayers; ++li) {
    const df3d_layer &L = layers[li];
    DF3D_CHECK_ARG(L.input >= -1 && L.input < li && L.residual >= -1 && L.residual < li,
                   "backbone_run: layer %d reads a later layer", li);
    const int in_set = L.input < 0 ? 0 : outs[L.input].set;
    const int cin = L.input < 0 ? in_channels : outs[L.input].channels;
    DF3D_CHECK_ARG(cin == L.cin, "backbone_run: layer %d expects %d input channels, gets %d", li, L.cin, cin);
    const int K = kvol_of(L.ksize);
    DF3D_CHECK_ARG(K > 0 && K <= DF3D_MAX_KVOL, "backbone_run: layer %d kernel volume %d unsupported", li, K);

    // ---- neighbour table: shared through the rulebook id, or built here ----
    const int32_t *nbr = nullptr;
    int out_set = -1, n_out = 0;
    if (L.rulebook >= 0)
      for (size_t r = 0; r < rulebooks.size(); ++r)
        if (rulebooks[r].first == L.rulebook) {
          nbr = rulebooks[r].second;
          out_set = rulebook_set[r];
          DF3D_CHECK_ARG(L.kind != 0 || out_set == in_set,
                         "backbone_run: layer %d shares rulebook %d but reads another index set", li, L.rulebook);
        }
    if (!nbr) {
      auto ensure_grid = [&](int si) -> int {
        IndexSet &S = sets[si];
        if (S.grid) return DF3D_OK;
        S.grid_bytes = df3d_grid_bytes(batch, S.shape);
        S.grid = mem.take(S.grid_bytes);
        if (!S.sorted) S.perm = (int32_t *)mem.take((size_t)S.n * 4);
        if (!S.grid || (!S.sorted && !S.perm)) return DF3D_ENOMEM;
        return df3d_grid_build(S.indices, S.n, batch, S.shape, S.grid, S.grid_bytes, S.perm, gs_);
      };
      if (L.kind == 0) {                       // submanifold: outputs = inputs
        int rc = ensure_grid(in_set);
        DF3D_ARENA_CHECK(rc != DF3D_ENOMEM);
        if (rc) return rc;
        const IndexSet &S = sets[in_set];
        int32_t *t = (int32_t *)mem.take((size_t)K * S.n * 4);
        DF3D_ARENA_CHECK(t);
        rc = df3d_subm_neighbors(S.grid, S.perm, S.indices, S.n, batch, S.shape, L.ksize, L.dilation, t, gs_);
        if (rc) return rc;
        nbr = t;
        out_set = in_set;
      } else {                                 // strided: enumerate the active outputs (one host round trip)
        const IndexSet S = sets[in_set];
        IndexSet O;
        for (int d = 0; d < 3; ++d)
          O.shape[d] = (S.shape[d] + 2 * L.padding[d] - L.dilation[d] * (L.ksize[d] - 1) - 1) / L.stride[d] + 1;
        long long fan = 1;
        for (int d = 0; d < 3; ++d) fan *= (L.ksize[d] + L.stride[d] - 1) / L.stride[d];
        long long vol = (long long)batch * O.shape[0] * O.shape[1] * O.shape[2];
        long long cap = (long long)S.n * (fan < K ? fan : K);
        if (cap > vol) cap = vol;
        if (cap < 1) cap = 1;
        O.grid_bytes = df3d_grid_bytes(batch, O.shape);
        O.grid = mem.take(O.grid_bytes);
        int32_t *oi = (int32_t *)mem.take((size_t)cap * 16);
        DF3D_ARENA_CHECK(O.grid && oi);
        int rc = df3d_conv_out_indices(S.indices, S.n, batch, S.shape, O.shape, L.ksize, L.stride, L.padding,
                                       L.dilation, O.grid, O.grid_bytes, oi, (int)cap, count_dev, gs_);
        if (rc) return rc;
        int32_t cnt = 0;
        DF3D_HIP(hipMemcpyAsync(&cnt, count_dev, sizeof(cnt), hipMemcpyDeviceToHost, gstream));
        DF3D_HIP(hipStreamSynchronize(gstream));
        if (cnt > cap) {
          set_error("backbone_run: layer %d produced %d outputs, capacity bound %lld", li, cnt, cap);
          return DF3D_EINVAL;
        }
        if (cnt <= 0) {
          set_error("backbone_run: layer %d has no active outputs", li);
          return DF3D_EINVAL;
        }
        O.indices = oi;
        O.n = cnt;
        O.sorted = true;
        sets.push_back(O);
        out_set = (int)sets.size() - 1;
        rc = ensure_grid(in_set);
        DF3D_ARENA_CHECK(rc != DF3D_ENOMEM);
        if (rc) return rc;
        const IndexSet &Sg = sets[in_set];
        int32_t *t = (int32_t *)mem.take((size_t)K * cnt * 4);
        DF3D_ARENA_CHECK(t);
        rc = df3d_conv_neighbors(Sg.grid, Sg.perm, oi, cnt, batch, Sg.shape, L.ksize, L.stride, L.padding, L.dilation,
                                 t, gs_);
        if (rc) return rc;
        nbr = t;
      }
      if (two_streams) {                       // the convolutions that use this table wait for it
        hipEvent_t e = order_event(next_event++);
        DF3D_CHECK_ARG(e != nullptr, "backbone_run: cannot create an event");
        DF3D_HIP(hipEventRecord(e, gstream));
        DF3D_HIP(hipStreamWaitEvent(stream, e, 0));
      }
      rulebooks.push_back(std::make_pair(L.rulebook, nbr));
      rulebook_set.push_back(out_set);
    }
    layer_nbr[li] = nbr;
    outs[li].set = out_set;
    outs[li].channels = L.cout;

    // ---- features: the fused convolution of this layer, enqueued on the caller's stream (no host wait) ----
    const float *in_feat = L.input < 0 ? features : outs[L.input].features;
    n_out = sets[out_set].n;
    const int n_in = sets[in_set].n;
    LayerOut &o = outs[li];
    df3d_layer_view &v = views[li];
    v.nbr = nbr;
    v.kvol = K;
    if (L.reserved & 1) {                      // geometry-only layer: the caller runs this convolution itself
      const IndexSet &GS = sets[out_set];
      v.features = nullptr;
      v.split = nullptr;
      v.indices = GS.indices;
      v.grid = GS.grid;
      v.grid_bytes = GS.grid_bytes;
      v.n = GS.n;
      v.channels = L.cout;
      v.rows_sorted = GS.sorted ? 1 : 0;
      memcpy(v.shape, GS.shape, sizeof(v.shape));
      continue;
    }
    DF3D_CHECK_ARG(L.input < 0 || outs[L.input].ran, "backbone_run: layer %d reads a geometry-only layer", li);
    o.ran = true;
    const bool split_layer = L.packed && !(L.reserved & 2) && df3d_conv_packed_weight_bytes(K, L.cin, L.cout) != 0;
    if (need_f32[li] || !split_layer) {
      o.features = (float *)mem.take((size_t)n_out * L.cout * 4);
      DF3D_ARENA_CHECK(o.features);
    }
    const float *res = L.residual < 0 ? nullptr : outs[L.residual].features;
    if (L.residual >= 0)
      DF3D_CHECK_ARG(outs[L.residual].set == out_set && outs[L.residual].channels == L.cout,
                     "backbone_run: residual of layer %d lives on another index set", li);
    if (L.packed && (L.reserved & 2)) {          // bf16 rows / bf16 weights (DF3D_CONV_PRECISION=bf16)
      DF3D_CHECK_ARG(df3d_conv_packed_weight_bytes_bf16(K, L.cin, L.cout) != 0,
                     "backbone_run: layer %d has no bf16 kernel (K=%d cin=%d cout=%d)", li, K, L.cin, L.cout);
      void **in16 = L.input < 0 ? &rows16_0 : &outs[L.input].rows16;
      if (!*in16) {
        *in16 = mem.take((size_t)n_in * L.cin * 2);
        DF3D_ARENA_CHECK(*in16);
        int rc = df3d_rows_to_bf16(in_feat, n_in, L.cin, *in16, stream_);
        if (rc) return rc;
      }
      const void *res16 = nullptr;
      if (L.residual >= 0) {
        LayerOut &R = outs[L.residual];
        if (!R.rows16) {
          R.rows16 = mem.take((size_t)n_out * L.cout * 2);
          DF3D_ARENA_CHECK(R.rows16);
          int rc = df3d_rows_to_bf16(R.features, n_out, L.cout, R.rows16, stream_);
          if (rc) return rc;
        }
        res16 = R.rows16;
      }
      o.rows16 = mem.take((size_t)n_out * L.cout * 2);
      DF3D_ARENA_CHECK(o.rows16);
      int rc = df3d_sparse_conv_bf16(*in16, n_in, L.cin, L.packed, K, L.cout, nbr, n_out, L.bias, L.scale, L.shift, res16,
                                     L.relu, o.features, o.rows16, stream_);
      if (rc) return rc;
    } else if (L.packed && df3d_conv_packed_weight_bytes(K, L.cin, L.cout) != 0) {
      void **in_split = L.input < 0 ? &split0 : &outs[L.input].split;
      if (!*in_split) {
        *in_split = mem.take((size_t)n_in * L.cin * 4);
        DF3D_ARENA_CHECK(*in_split);
        int rc = df3d_split_rows(in_feat, n_in, L.cin, *in_split, stream_);
        if (rc) return rc;
      }
      o.split = mem.take((size_t)n_out * L.cout * 4);
      DF3D_ARENA_CHECK(o.split);
      int rc = df3d_sparse_conv_split(*in_split, n_in, L.cin, L.packed, K, L.cout, nbr, n_out, L.bias, L.scale, L.shift,
                                      res, L.relu, o.features, o.split, nullptr, 0, stream_);
      if (rc) return rc;
    } else {
      int rc = df3d_sparse_conv_fused(in_feat, n_in, L.cin, L.weight, K, L.cout, nbr, n_out, L.bias, L.scale, L.shift,
                                      res, L.relu, o.features, stream_);
      if (rc) return rc;
    }
    const IndexSet &OS = sets[out_set];
    v.features = o.features;
    v.split = o.split ? o.split : o.rows16;      // reserved bit 1 tells the caller which format this is
    v.reserved = o.rows16 && !o.split ? 2 : 0;
    v.indices = OS.indices;
    v.grid = OS.grid;          // may still be NULL: directories are built when a later layer needs them
    v.grid_bytes = OS.grid_bytes;
    v.n = OS.n;
    v.channels = L.cout;
    v.rows_sorted = OS.sorted ? 1 : 0;
    memcpy(v.shape, OS.shape, sizeof(v.shape));
  }
  if (two_streams) {                           // exported index sets / directories were written on the geometry stream
    hipEvent_t e = order_event(next_event++);
    DF3D_CHECK_ARG(e != nullptr, "backbone_run: cannot create an event");
    DF3D_HIP(hipEventRecord(e, gstream));
    DF3D_HIP(hipStreamWaitEvent(stream, e, 0));
  }
  // directories may have been built after a view was written: refresh
  for (int li = 0; li < nlayers; ++li) {
    const IndexSet &OS = sets[outs[li].set];
    views[li].grid = OS.grid;
    views[li].grid_bytes = OS.grid_bytes;
  }
  if (arena_used) *arena_used = mem.used;
  return DF3D_OK;
#undef DF3D_ARENA_CHECK
}
