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
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

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

// ---- one run of a layer table: the geometry phase (neighbour tables, index sets, directories -- everything that depends on the
//      COORDINATES alone) and the convolution phase, layer by layer.  df3d_backbone_run interleaves the two (a layer's
//      convolution is queued as soon as its table is); df3d_backbone_geometry / df3d_backbone_convs run them as two calls, so
//      that a caller can build the geometry of frame k + 1 -- with its host round trips -- from another host thread while it
//      still queues frame k (dualfusion/prefetch.py). ----
namespace {

struct RunState {
  const df3d_layer *layers = nullptr;
  int nlayers = 0;
  const int32_t *indices = nullptr;
  int n = 0, in_channels = 0, batch = 0;
  int shape[3] = {0, 0, 0};
  Bump *gmem = nullptr;        // tables, index sets, directories
  Bump *fmem = nullptr;        // features, split rows (the same object as gmem in a single-arena run)
  Bump own_g{nullptr, 0}, own_f{nullptr, 0};
  std::vector<IndexSet> sets;
  std::vector<LayerOut> outs;
  std::vector<std::pair<int, const int32_t *>> rulebooks;     // (rulebook id, nbr table)
  std::vector<int> rulebook_set;                              // index set of the rulebook's outputs
  std::vector<const int32_t *> layer_nbr;
  std::vector<const void *> rulebook_lists;                   // per rulebook: df3d_nbr_row_lists blob or NULL
  std::vector<const void *> layer_lists;                      // per layer: its table's row lists or NULL
  std::vector<int> layer_in_set;
  std::vector<char> need_f32;
  std::vector<char> want_split;   // a two-part split-precision layer reads this layer's rows (round 4: the small-channel kernel emits them)
  int32_t *count_dev = nullptr;
  void *split0 = nullptr;      // split rows of the network input, built on demand
  void *rows16_0 = nullptr;    // its bf16 rows (bf16 layers)
  hipEvent_t done = nullptr;   // split API: the geometry phase is complete on its stream
  hipEvent_t img_done = nullptr;   // frame head: the image projection is complete on the worker's second stream
  int geometry_layers = 0;     // layers whose geometry has been built
  const float *input_features = nullptr;   // df3d_backbone_convs_range: the network input of the first range
  int range_next = 0;                       // ... and the layer the next range starts at
};

int fail_arena(RunState &S, Bump *m, size_t *arena_used, size_t arena_bytes) {
  if (arena_used) *arena_used = m->used;
  set_error("backbone_run: arena of %zu bytes is too small (needed more than %zu)", arena_bytes, m->used);
  (void)S;
  return DF3D_ENOMEM;
}

int state_init(RunState &S, const df3d_layer *layers, int nlayers, const int32_t *indices, int n, int in_channels, int batch,
               const int *shape) {
  S.layers = layers, S.nlayers = nlayers, S.indices = indices, S.n = n, S.in_channels = in_channels, S.batch = batch;
  memcpy(S.shape, shape, sizeof(S.shape));
  S.outs.assign(nlayers, LayerOut());
  S.layer_nbr.assign(nlayers, nullptr);
  S.layer_lists.assign(nlayers, nullptr);
  S.layer_in_set.assign(nlayers, 0);
  IndexSet s0;
  s0.indices = indices;
  s0.n = n;
  memcpy(s0.shape, shape, sizeof(s0.shape));
  S.sets.push_back(s0);
  // Which layers must emit fp32 rows next to their split rows: exported stages (flag bit 2, set by the caller), residual
  // sources, inputs of layers that do not run on the split-precision kernels.  The first convolution of every residual block
  // feeds exactly one split-precision convolution: its fp32 rows were written and never read (round 2: every output twice).
  S.need_f32.assign(nlayers, 0);
  S.want_split.assign(nlayers, 0);
  bool flagged = false;
  for (int li = 0; li < nlayers; ++li) flagged = flagged || (layers[li].reserved & 4);
  static const bool write_all = getenv("DF3D_EXEC_F32_ALL") && atoi(getenv("DF3D_EXEC_F32_ALL"));   // A / B of the PMC passes
  if (write_all) flagged = false;
  for (int li = 0; li < nlayers; ++li) {
    const df3d_layer &L = layers[li];
    if (!flagged || (L.reserved & 4) || li == nlayers - 1) S.need_f32[li] = 1;
    if (L.residual >= 0 && L.residual < nlayers) S.need_f32[L.residual] = 1;
    const bool split_consumer = L.packed && !(L.reserved & 2) && !(L.reserved & 1) &&
                                ((L.reserved & 8) ? df3d_conv_packed_weight_bytes3(kvol_of(L.ksize), L.cin, L.cout) != 0
                                                  : df3d_conv_packed_weight_bytes(kvol_of(L.ksize), L.cin, L.cout) != 0);
    if (L.input >= 0 && L.input < nlayers && !split_consumer) S.need_f32[L.input] = 1;
    if (L.input >= 0 && L.input < nlayers && split_consumer && !(L.reserved & 8)) S.want_split[L.input] = 1;
  }
  return DF3D_OK;
}

// Geometry of layer li on `gstream`.  *new_table: a table was built (the convolutions that use it must wait for `gstream`).
// DF3D_ENOMEM: the geometry arena is too small.
int geo_layer(RunState &S, int li, hipStream_t gstream, bool *new_table) {
  void *gs_ = (void *)gstream;
  Bump &mem = *S.gmem;
  const df3d_layer &L = S.layers[li];
  const int batch = S.batch;
  *new_table = false;
  DF3D_CHECK_ARG(L.input >= -1 && L.input < li && L.residual >= -1 && L.residual < li,
                 "backbone_run: layer %d reads a later layer", li);
  const int in_set = L.input < 0 ? 0 : S.outs[L.input].set;
  const int cin = L.input < 0 ? S.in_channels : S.outs[L.input].channels;
  DF3D_CHECK_ARG(cin == L.cin, "backbone_run: layer %d expects %d input channels, gets %d", li, L.cin, cin);
  const int K = kvol_of(L.ksize);
  DF3D_CHECK_ARG(K > 0 && K <= DF3D_MAX_KVOL, "backbone_run: layer %d kernel volume %d unsupported", li, K);
  if (!S.count_dev) {
    S.count_dev = (int32_t *)mem.take(256);
    if (!S.count_dev) return DF3D_ENOMEM;
  }
  // ---- neighbour table: shared through the rulebook id, or built here ----
  const int32_t *nbr = nullptr;
  const void *lists = nullptr;
  int out_set = -1;
  if (L.rulebook >= 0)
    for (size_t r = 0; r < S.rulebooks.size(); ++r)
      if (S.rulebooks[r].first == L.rulebook) {
        nbr = S.rulebooks[r].second;
        lists = S.rulebook_lists[r];
        out_set = S.rulebook_set[r];
        DF3D_CHECK_ARG(L.kind != 0 || out_set == in_set,
                       "backbone_run: layer %d shares rulebook %d but reads another index set", li, L.rulebook);
      }
  if (!nbr) {
    auto ensure_grid = [&](int si) -> int {
      IndexSet &T = S.sets[si];
      if (T.grid) return DF3D_OK;
      T.grid_bytes = df3d_grid_bytes(batch, T.shape);
      T.grid = mem.take(T.grid_bytes);
      if (!T.sorted) T.perm = (int32_t *)mem.take((size_t)T.n * 4);
      if (!T.grid || (!T.sorted && !T.perm)) return DF3D_ENOMEM;
      return df3d_grid_build(T.indices, T.n, batch, T.shape, T.grid, T.grid_bytes, T.perm, gs_);
    };
    if (L.kind == 0) {                       // submanifold: outputs = inputs
      int rc = ensure_grid(in_set);
      if (rc) return rc;
      const IndexSet &T = S.sets[in_set];
      int32_t *t = (int32_t *)mem.take((size_t)K * T.n * 4);
      if (!t) return DF3D_ENOMEM;
      rc = df3d_subm_neighbors(T.grid, T.perm, T.indices, T.n, batch, T.shape, L.ksize, L.dilation, t, gs_);
      if (rc) return rc;
      nbr = t;
      out_set = in_set;
    } else {                                 // strided: enumerate the active outputs (one host round trip)
      const IndexSet T = S.sets[in_set];
      IndexSet O;
      for (int d = 0; d < 3; ++d)
        O.shape[d] = (T.shape[d] + 2 * L.padding[d] - L.dilation[d] * (L.ksize[d] - 1) - 1) / L.stride[d] + 1;
      long long fan = 1;
      for (int d = 0; d < 3; ++d) fan *= (L.ksize[d] + L.stride[d] - 1) / L.stride[d];
      long long vol = (long long)batch * O.shape[0] * O.shape[1] * O.shape[2];
      long long cap = (long long)T.n * (fan < K ? fan : K);
      if (cap > vol) cap = vol;
      if (cap < 1) cap = 1;
      O.grid_bytes = df3d_grid_bytes(batch, O.shape);
      O.grid = mem.take(O.grid_bytes);
      int32_t *oi = (int32_t *)mem.take((size_t)cap * 16);
      if (!O.grid || !oi) return DF3D_ENOMEM;
      int rc = df3d_conv_out_indices(T.indices, T.n, batch, T.shape, O.shape, L.ksize, L.stride, L.padding,
                                     L.dilation, O.grid, O.grid_bytes, oi, (int)cap, S.count_dev, gs_);
      if (rc) return rc;
      int32_t cnt = 0;
      DF3D_HIP(hipMemcpyAsync(&cnt, S.count_dev, sizeof(cnt), hipMemcpyDeviceToHost, gstream));
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
      S.sets.push_back(O);
      out_set = (int)S.sets.size() - 1;
      rc = ensure_grid(in_set);
      if (rc) return rc;
      const IndexSet &Tg = S.sets[in_set];
      int32_t *t = (int32_t *)mem.take((size_t)K * cnt * 4);
      if (!t) return DF3D_ENOMEM;
      rc = df3d_conv_neighbors(Tg.grid, Tg.perm, oi, cnt, batch, Tg.shape, L.ksize, L.stride, L.padding, L.dilation,
                               t, gs_);
      if (rc) return rc;
      nbr = t;
    }
    *new_table = true;
    // Row lists of the table (round 4) when the layer that builds it runs on the small-channel vector kernel: those tables
    // are ~90 % empty, and the kernel then reads the present pairs instead of all K entries of every row.  Built here, in the
    // geometry phase (the frame head's stream, a frame ahead); layers that share the rulebook share the lists.
    static const bool lists_on = !(getenv("DF3D_ROW_LISTS") && getenv("DF3D_ROW_LISTS")[0] == '0');
    const bool fp32_small = !(L.reserved & 1) && !(L.packed && (L.reserved & (2 | 8))) &&
                            !(L.packed && df3d_conv_packed_weight_bytes(K, L.cin, L.cout) != 0);   // conv_layer's last branch
    const int n_rows = S.sets[out_set].n;
    if (lists_on && fp32_small && L.cin <= 16 && (L.cout == 16 || L.cout == 32) && n_rows >= 2048 &&
        S.sets[in_set].n < (1 << 26)) {
      const size_t lb = df3d_nbr_row_lists_bytes(K, n_rows);
      void *blob = lb ? mem.take(lb) : nullptr;
      if (lb && !blob) return DF3D_ENOMEM;
      if (blob) {
        int rc = df3d_nbr_row_lists(nbr, K, n_rows, S.sets[in_set].n, blob, lb, gs_);
        if (rc) return rc;
        lists = blob;
      }
    }
    S.rulebooks.push_back(std::make_pair(L.rulebook, nbr));
    S.rulebook_lists.push_back(lists);
    S.rulebook_set.push_back(out_set);
  }
  S.layer_nbr[li] = nbr;
  S.layer_lists[li] = lists;
  S.layer_in_set[li] = in_set;
  S.outs[li].set = out_set;
  S.outs[li].channels = L.cout;
  S.geometry_layers = li + 1;
  return DF3D_OK;
}

// the geometry fields of a layer's view
void view_geometry(const RunState &S, int li, df3d_layer_view &v) {
  const IndexSet &OS = S.sets[S.outs[li].set];
  v.nbr = S.layer_nbr[li];
  v.kvol = kvol_of(S.layers[li].ksize);
  v.indices = OS.indices;
  v.grid = OS.grid;          // may still be NULL: directories are built when a later layer needs them
  v.grid_bytes = OS.grid_bytes;
  v.n = OS.n;
  v.channels = S.layers[li].cout;
  v.rows_sorted = OS.sorted ? 1 : 0;
  memcpy(v.shape, OS.shape, sizeof(v.shape));
}

// The fused convolution of layer li, enqueued on the caller's stream (no host wait).  DF3D_ENOMEM: feature arena too small.
int conv_layer(RunState &S, int li, const float *features, df3d_layer_view *views, hipStream_t stream) {
  void *stream_ = (void *)stream;
  Bump &mem = *S.fmem;
  const df3d_layer &L = S.layers[li];
  const int K = kvol_of(L.ksize);
  const int32_t *nbr = S.layer_nbr[li];
  const int in_set = S.layer_in_set[li], out_set = S.outs[li].set;
  const float *in_feat = L.input < 0 ? features : S.outs[L.input].features;
  const int n_out = S.sets[out_set].n;
  const int n_in = S.sets[in_set].n;
  LayerOut &o = S.outs[li];
  df3d_layer_view &v = views[li];
  view_geometry(S, li, v);
  if (L.reserved & 1) {                      // geometry-only layer: the caller runs this convolution itself
    v.features = nullptr;
    v.split = nullptr;
    return DF3D_OK;
  }
  DF3D_CHECK_ARG(L.input < 0 || S.outs[L.input].ran, "backbone_run: layer %d reads a geometry-only layer", li);
  o.ran = true;
  const bool split_layer = L.packed && !(L.reserved & 2) &&
                           ((L.reserved & 8) ? df3d_conv_packed_weight_bytes3(K, L.cin, L.cout) != 0
                                             : df3d_conv_packed_weight_bytes(K, L.cin, L.cout) != 0);
  if (S.need_f32[li] || !split_layer) {
    o.features = (float *)mem.take((size_t)n_out * L.cout * 4);
    if (!o.features) return DF3D_ENOMEM;
  }
  const float *res = L.residual < 0 ? nullptr : S.outs[L.residual].features;
  if (L.residual >= 0)
    DF3D_CHECK_ARG(S.outs[L.residual].set == out_set && S.outs[L.residual].channels == L.cout,
                   "backbone_run: residual of layer %d lives on another index set", li);
  if (L.packed && (L.reserved & 2)) {          // bf16 rows / bf16 weights (DF3D_CONV_PRECISION=bf16)
    DF3D_CHECK_ARG(df3d_conv_packed_weight_bytes_bf16(K, L.cin, L.cout) != 0,
                   "backbone_run: layer %d has no bf16 kernel (K=%d cin=%d cout=%d)", li, K, L.cin, L.cout);
    void **in16 = L.input < 0 ? &S.rows16_0 : &S.outs[L.input].rows16;
    if (!*in16) {
      *in16 = mem.take((size_t)n_in * L.cin * 2);
      if (!*in16) return DF3D_ENOMEM;
      int rc = df3d_rows_to_bf16(in_feat, n_in, L.cin, *in16, stream_);
      if (rc) return rc;
    }
    const void *res16 = nullptr;
    if (L.residual >= 0) {
      LayerOut &R = S.outs[L.residual];
      if (!R.rows16) {
        R.rows16 = mem.take((size_t)n_out * L.cout * 2);
        if (!R.rows16) return DF3D_ENOMEM;
        int rc = df3d_rows_to_bf16(R.features, n_out, L.cout, R.rows16, stream_);
        if (rc) return rc;
      }
      res16 = R.rows16;
    }
    o.rows16 = mem.take((size_t)n_out * L.cout * 2);
    if (!o.rows16) return DF3D_ENOMEM;
    int rc = df3d_sparse_conv_bf16(*in16, n_in, L.cin, L.packed, K, L.cout, nbr, n_out, L.bias, L.scale, L.shift, res16,
                                   L.relu, o.features, o.rows16, stream_);
    if (rc) return rc;
  } else if (L.packed && (L.reserved & 8)) {     // three-part operands ("split3" precision)
    DF3D_CHECK_ARG(df3d_conv_packed_weight_bytes3(K, L.cin, L.cout) != 0,
                   "backbone_run: layer %d has no three-part kernel (K=%d cin=%d cout=%d)", li, K, L.cin, L.cout);
    void **in3 = L.input < 0 ? &S.split0 : &S.outs[L.input].split;
    if (!*in3) {
      *in3 = mem.take((size_t)n_in * L.cin * 6);
      if (!*in3) return DF3D_ENOMEM;
      int rc = df3d_split_rows3(in_feat, n_in, L.cin, *in3, stream_);
      if (rc) return rc;
    }
    o.split = mem.take((size_t)n_out * L.cout * 6);
    if (!o.split) return DF3D_ENOMEM;
    int rc = df3d_conv_rows_split3(*in3, n_in, L.cin, L.cin, 0, L.packed, K, L.cout, 1, nbr, n_out, L.bias, L.scale, L.shift,
                                   res, L.relu, o.features, L.cout, nullptr, o.split, stream_);
    if (rc) return rc;
  } else if (L.packed && df3d_conv_packed_weight_bytes(K, L.cin, L.cout) != 0) {
    void **in_split = L.input < 0 ? &S.split0 : &S.outs[L.input].split;
    if (!*in_split) {
      *in_split = mem.take((size_t)n_in * L.cin * 4);
      if (!*in_split) return DF3D_ENOMEM;
      int rc = df3d_split_rows(in_feat, n_in, L.cin, *in_split, stream_);
      if (rc) return rc;
    }
    o.split = mem.take((size_t)n_out * L.cout * 4);
    if (!o.split) return DF3D_ENOMEM;
    int rc = df3d_sparse_conv_split(*in_split, n_in, L.cin, L.packed, K, L.cout, nbr, n_out, L.bias, L.scale, L.shift,
                                    res, L.relu, o.features, o.split, nullptr, 0, stream_);
    if (rc) return rc;
  } else {
    // a matrix-core layer reads this one (the stride-2 16 -> 32 layer feeds conv2's blocks): the operand split of the rows
    // comes out of the same launch instead of a df3d_split_rows pass in front of the consumer
    if (S.want_split[li] && S.layer_lists[li] && L.cout % 8 == 0) {
      o.split = mem.take((size_t)n_out * L.cout * 4);
      if (!o.split) return DF3D_ENOMEM;
    }
    int rc = df3d_sparse_conv_fused_lists(in_feat, n_in, L.cin, L.weight, K, L.cout, nbr, S.layer_lists[li], n_out, L.bias,
                                          L.scale, L.shift, res, L.relu, o.features, o.split, stream_);
    if (rc) return rc;
  }
  v.features = o.features;
  v.split = o.split ? o.split : o.rows16;      // reserved bit 1 tells the caller which format this is
  v.reserved = o.rows16 && !o.split ? 2 : ((L.reserved & 8) && o.split ? 4 : 0);
  return DF3D_OK;
}

}  // namespace

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
  RunState S;
  S.own_g = Bump(arena, arena_bytes);
  S.gmem = S.fmem = &S.own_g;
  int rc = state_init(S, layers, nlayers, indices, n, in_channels, batch, shape);
  if (rc) return rc;
  // ---- per layer: geometry (neighbour table; for strided layers the output index set, with the one host round
  //      trip for its size) on the geometry stream, then the fused convolution on the caller's stream ----
  for (int li = 0; li < nlayers; ++li) {
    bool new_table = false;
    rc = geo_layer(S, li, gstream, &new_table);
    if (rc == DF3D_ENOMEM) return fail_arena(S, S.gmem, arena_used, arena_bytes);
    if (rc) return rc;
    if (new_table && two_streams) {            // the convolutions that use this table wait for it
      hipEvent_t e = order_event(next_event++);
      DF3D_CHECK_ARG(e != nullptr, "backbone_run: cannot create an event");
      DF3D_HIP(hipEventRecord(e, gstream));
      DF3D_HIP(hipStreamWaitEvent(stream, e, 0));
    }
    rc = conv_layer(S, li, features, views, stream);
    if (rc == DF3D_ENOMEM) return fail_arena(S, S.fmem, arena_used, arena_bytes);
    if (rc) return rc;
  }
  if (two_streams) {                           // exported index sets / directories were written on the geometry stream
    hipEvent_t e = order_event(next_event++);
    DF3D_CHECK_ARG(e != nullptr, "backbone_run: cannot create an event");
    DF3D_HIP(hipEventRecord(e, gstream));
    DF3D_HIP(hipStreamWaitEvent(stream, e, 0));
  }
  // directories may have been built after a view was written: refresh
  for (int li = 0; li < nlayers; ++li) {
    const IndexSet &OS = S.sets[S.outs[li].set];
    views[li].grid = OS.grid;
    views[li].grid_bytes = OS.grid_bytes;
  }
  if (arena_used) *arena_used = S.gmem->used;
  return DF3D_OK;
}

// ---- the two phases as two calls (a frame's geometry built ahead of its convolutions, possibly by another host thread) ----
extern "C" int df3d_backbone_geometry(const df3d_layer *layers, int nlayers, const int32_t *indices, int n, int in_channels,
                                      int batch, const int *shape, void *arena, size_t arena_bytes, void *inputs_ready,
                                      df3d_layer_view *views, size_t *arena_used, void **handle) {
  DF3D_CHECK_ARG(layers && nlayers > 0 && indices && shape && arena && views && handle, "backbone_geometry: null argument");
  DF3D_CHECK_ARG(n > 0 && batch > 0, "backbone_geometry: empty input (n=%d, batch=%d)", n, batch);
  *handle = nullptr;
  if (!g_geo_stream) DF3D_HIP(hipStreamCreateWithFlags(&g_geo_stream, hipStreamNonBlocking));
  hipStream_t gstream = g_geo_stream;
  // the coordinates are complete at `inputs_ready` (a voxeliser's event); NULL: they are complete now (the caller has synchronised)
  if (inputs_ready) DF3D_HIP(hipStreamWaitEvent(gstream, (hipEvent_t)inputs_ready, 0));
  RunState *S = new RunState();
  S->own_g = Bump(arena, arena_bytes);
  S->gmem = &S->own_g;
  int rc = state_init(*S, layers, nlayers, indices, n, in_channels, batch, shape);
  for (int li = 0; li < nlayers && !rc; ++li) {
    bool new_table = false;
    rc = geo_layer(*S, li, gstream, &new_table);
  }
  if (arena_used) *arena_used = S->gmem->used;
  if (rc == DF3D_ENOMEM) set_error("backbone_geometry: arena of %zu bytes is too small (needed more than %zu)", arena_bytes, S->gmem->used);
  if (!rc && hipEventCreateWithFlags(&S->done, hipEventDisableTiming) != hipSuccess) {
    set_error("backbone_geometry: cannot create an event");
    rc = DF3D_EHIP;
  }
  if (!rc && hipEventRecord(S->done, gstream) != hipSuccess) {
    set_error("backbone_geometry: hipEventRecord failed");
    rc = DF3D_EHIP;
  }
  if (rc) {
    if (S->done) (void)hipEventDestroy(S->done);
    delete S;
    return rc;
  }
  for (int li = 0; li < nlayers; ++li) {
    view_geometry(*S, li, views[li]);
    views[li].features = nullptr;
    views[li].split = nullptr;
    views[li].reserved = 0;
  }
  *handle = S;
  return DF3D_OK;
}

extern "C" int df3d_backbone_convs(void *handle, const float *features, void *arena, size_t arena_bytes,
                                   df3d_layer_view *views, size_t *arena_used, void *stream_) {
  DF3D_CHECK_ARG(handle && features && arena && views, "backbone_convs: null argument");
  RunState &S = *(RunState *)handle;
  DF3D_CHECK_ARG(S.geometry_layers == S.nlayers, "backbone_convs: the geometry phase of this handle did not finish");
  hipStream_t stream = (hipStream_t)stream_;
  DF3D_HIP(hipStreamWaitEvent(stream, S.done, 0));
  // a retry after DF3D_ENOMEM starts over on the larger arena
  S.own_f = Bump(arena, arena_bytes);
  S.fmem = &S.own_f;
  S.split0 = S.rows16_0 = nullptr;
  for (auto &o : S.outs) o.features = nullptr, o.split = nullptr, o.rows16 = nullptr, o.ran = false;
  for (int li = 0; li < S.nlayers; ++li) {
    int rc = conv_layer(S, li, features, views, stream);
    if (rc == DF3D_ENOMEM) {
      if (arena_used) *arena_used = S.fmem->used;
      set_error("backbone_convs: arena of %zu bytes is too small (needed more than %zu)", arena_bytes, S.fmem->used);
      return DF3D_ENOMEM;
    }
    if (rc) return rc;
  }
  if (arena_used) *arena_used = S.fmem->used;
  return DF3D_OK;
}

// The convolutions of layers [first, last) of a prepared chain (round 5): a caller whose chain is interrupted by steps that MODIFY
// a stage's features (a fusion layer between two stages) builds the geometry of the WHOLE chain ahead -- it depends on the
// coordinates alone, whatever happens to the features -- and runs the convolutions range by range: `features` of the first range
// (first == 0) is the network input; of a later range it REPLACES the fp32 rows of layer first - 1 (same index set and
// channel count; the operand split of those rows is rebuilt from them).  Ranges must be consecutive and use the same arena.
extern "C" int df3d_backbone_convs_range(void *handle, const float *features, int first, int last, void *arena,
                                         size_t arena_bytes, df3d_layer_view *views, size_t *arena_used, void *stream_) {
  DF3D_CHECK_ARG(handle && features && arena && views, "backbone_convs_range: null argument");
  RunState &S = *(RunState *)handle;
  DF3D_CHECK_ARG(S.geometry_layers == S.nlayers, "backbone_convs_range: the geometry phase of this handle did not finish");
  DF3D_CHECK_ARG(first >= 0 && first < last && last <= S.nlayers, "backbone_convs_range: bad layer range [%d, %d) of %d", first,
                 last, S.nlayers);
  hipStream_t stream = (hipStream_t)stream_;
  if (first == 0) {
    DF3D_HIP(hipStreamWaitEvent(stream, S.done, 0));
    S.own_f = Bump(arena, arena_bytes);
    S.fmem = &S.own_f;
    S.split0 = S.rows16_0 = nullptr;
    for (auto &o : S.outs) o.features = nullptr, o.split = nullptr, o.rows16 = nullptr, o.ran = false;
    S.input_features = features;
  } else {
    DF3D_CHECK_ARG(S.range_next == first && S.fmem == &S.own_f && S.own_f.base == (char *)arena,
                   "backbone_convs_range: range [%d, %d) does not continue the previous one (next layer %d)", first, last,
                   S.range_next);
    LayerOut &p = S.outs[first - 1];
    DF3D_CHECK_ARG(p.ran, "backbone_convs_range: layer %d has not run", first - 1);
    p.features = (float *)features;            // the hook's rows; their operand formats are rebuilt on demand
    p.split = nullptr;
    p.rows16 = nullptr;
  }
  for (int li = first; li < last; ++li) {
    int rc = conv_layer(S, li, S.input_features, views, stream);
    if (rc == DF3D_ENOMEM) {
      if (arena_used) *arena_used = S.fmem->used;
      set_error("backbone_convs_range: arena of %zu bytes is too small (needed more than %zu)", arena_bytes, S.fmem->used);
      return DF3D_ENOMEM;
    }
    if (rc) return rc;
  }
  S.range_next = last;
  if (arena_used) *arena_used = S.fmem->used;
  return DF3D_OK;
}

extern "C" int df3d_backbone_geometry_wait(void *handle, void *stream_) {
  DF3D_CHECK_ARG(handle, "backbone_geometry_wait: null handle");
  DF3D_HIP(hipStreamWaitEvent((hipStream_t)stream_, ((RunState *)handle)->done, 0));
  return DF3D_OK;
}

extern "C" int df3d_frame_head_image_wait(void *handle, void *stream_) {
  DF3D_CHECK_ARG(handle && ((RunState *)handle)->img_done, "frame_head_image_wait: no image projection in this handle");
  DF3D_HIP(hipStreamWaitEvent((hipStream_t)stream_, ((RunState *)handle)->img_done, 0));
  return DF3D_OK;
}

extern "C" int df3d_backbone_release(void *handle) {
  if (!handle) return DF3D_OK;
  RunState *S = (RunState *)handle;
  if (S->done) (void)hipEventDestroy(S->done);
  if (S->img_done) (void)hipEventDestroy(S->img_done);
  delete S;
  return DF3D_OK;
}


// =====================================================================================================================
// Frame head on a worker thread: everything of a frame that depends on its RAW INPUTS alone -- voxelisation (+ mean VFE), every
// rulebook of the backbone, the camera projection / query slots of the fusion adapter -- with ALL its count round trips, off
// the thread that queues the frames.  The reference hides the first of these the same way: its DataLoader worker processes
// voxelise batch k + 1 on the CPU while the GPU step of batch k runs (CP/det3d/datasets/pipelines/preprocess.py `Voxelization`).
// A native thread (not a Python one): the queueing thread is an interpreter, and a second interpreter thread costs it the
// lock at every one of its ~150 C calls per frame (measured: +0.4 ms per frame, tools/debug/host_enqueue.py).
// =====================================================================================================================
namespace {

struct HeadTicket {
  df3d_frame_head_desc d;
  std::vector<const float *> points;
  std::vector<int> num_points;
  Bump mem{nullptr, 0};
  size_t arena_bytes = 0;
  std::vector<df3d_layer_view> views;
  df3d_frame_head_out out;
  RunState *state = nullptr;
  int rc = DF3D_OK;
  std::string err;
  bool done = false;
  std::mutex mu;
  std::condition_variable cv;
};

struct HeadWorker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<HeadTicket *> q;
  bool stop = false;
  int device = 0;
};

thread_local hipStream_t g_img_stream = nullptr;      // the worker's second stream (default priority): image projection

#define HEAD_TAKE(ptr, type, bytes)              \
  do {                                           \
    ptr = (type)t.mem.take(bytes);               \
    if (!ptr) return DF3D_ENOMEM;                \
  } while (0)

int frame_head_run(HeadTicket &t) {
  const df3d_frame_head_desc &d = t.d;
  if (!g_geo_stream) {
    // The worker's ~60 small kernels compete with the full-chip kernels of the frames in flight; at the default priority every
    // one of them waits for workgroup slots (measured with two frames in flight: a head took ~7 ms, the queueing thread waited
    // 1.4 ms per frame for it).  DF3D_HEAD_PRIORITY=0: default priority.
    static const bool high = !(getenv("DF3D_HEAD_PRIORITY") && getenv("DF3D_HEAD_PRIORITY")[0] == '0');
    int least = 0, greatest = 0;
    if (high && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least) {
      DF3D_HIP(hipStreamCreateWithPriority(&g_geo_stream, hipStreamNonBlocking, greatest));
    } else {
      DF3D_HIP(hipStreamCreateWithFlags(&g_geo_stream, hipStreamNonBlocking));
    }
  }
  hipStream_t gs = g_geo_stream;
  void *gs_ = (void *)gs;
  const int B = d.batch, C = d.point_channels;
  memset(&t.out, 0, sizeof(t.out));
  static const bool dbg = getenv("DF3D_HEAD_DEBUG") != nullptr;     // phase by phase, synchronised (localises a device fault)
#define HEAD_PHASE(msg)                                                   \
  do {                                                                    \
    if (dbg) {                                                            \
      hipError_t e__ = hipStreamSynchronize(gs);                          \
      fprintf(stderr, "df3d frame head: %s -> %d\n", msg, (int)e__);     \
      fflush(stderr);                                                     \
    }                                                                     \
  } while (0)
  if (d.inputs_ready) DF3D_HIP(hipStreamWaitEvent(gs, (hipEvent_t)d.inputs_ready, 0));
  // ---- 1. voxelisation + mean VFE of every cloud, ONE round trip for the voxel counts ----
  std::vector<float *> mean(B);
  std::vector<int32_t *> coors(B);
  std::vector<int> cap(B);
  int32_t *counts_dev;
  HEAD_TAKE(counts_dev, int32_t *, (size_t)(B > 64 ? B : 64) * 4);
  for (int b = 0; b < B; ++b) {
    const int P = t.num_points[b];
    cap[b] = (d.max_voxels < 0 || d.max_voxels > P) ? P : d.max_voxels;
    if (cap[b] < 1) cap[b] = 1;
    int32_t *num;
    void *ws;
    const size_t wsb = df3d_hard_voxelize_workspace_bytes(P, d.max_points, cap[b]);
    HEAD_TAKE(mean[b], float *, (size_t)cap[b] * C * 4);
    HEAD_TAKE(coors[b], int32_t *, (size_t)cap[b] * 16);
    HEAD_TAKE(num, int32_t *, (size_t)cap[b] * 4);
    HEAD_TAKE(ws, void *, wsb ? wsb : 1);
    int rc = df3d_hard_voxelize_batched(t.points[b], P, C, d.voxel_size, d.coors_range, d.max_points, cap[b], d.break_at_cap, b,
                                        nullptr, coors[b], num, mean[b], counts_dev + b, ws, wsb, gs_);
    if (rc) return rc;
  }
  HEAD_PHASE("voxelize queued");
  std::vector<int32_t> cnt(B, 0);
  DF3D_HIP(hipMemcpyAsync(cnt.data(), counts_dev, (size_t)B * 4, hipMemcpyDeviceToHost, gs));
  DF3D_HIP(hipStreamSynchronize(gs));
  long long total = 0;
  for (int b = 0; b < B; ++b) total += cnt[b];
  float *feats = mean[0];
  int32_t *ind = coors[0];
  if (B > 1) {                              // the samples' rows packed one after the other (the reference collates on the host)
    HEAD_TAKE(feats, float *, (size_t)(total ? total : 1) * C * 4);
    HEAD_TAKE(ind, int32_t *, (size_t)(total ? total : 1) * 16);
    size_t at = 0;
    for (int b = 0; b < B; ++b) {
      if (cnt[b] > 0) {
        DF3D_HIP(hipMemcpyAsync(feats + at * C, mean[b], (size_t)cnt[b] * C * 4, hipMemcpyDeviceToDevice, gs));
        DF3D_HIP(hipMemcpyAsync(ind + at * 4, coors[b], (size_t)cnt[b] * 16, hipMemcpyDeviceToDevice, gs));
      }
      at += cnt[b];
    }
  }
  t.out.features = feats;
  t.out.coors = ind;
  t.out.n = (int)total;
  RunState *S = new RunState();
  t.state = S;
  if (hipEventCreateWithFlags(&S->done, hipEventDisableTiming) != hipSuccess) {
    set_error("frame_head: cannot create an event");
    return DF3D_EHIP;
  }
  if (d.img_ptrs && d.img_count > 0) {
    // the image-side projection: beside the geometry below, on a stream of default priority
    if (!g_img_stream) DF3D_HIP(hipStreamCreateWithFlags(&g_img_stream, hipStreamNonBlocking));
    HEAD_TAKE(t.out.img_split, void *, (size_t)d.img_count * d.img_pixels * 512);
    HEAD_TAKE(t.out.img_gate, float *, (size_t)d.img_count * d.img_pixels * 4);
    if (hipEventCreateWithFlags(&S->img_done, hipEventDisableTiming) != hipSuccess) {
      set_error("frame_head: cannot create an event");
      return DF3D_EHIP;
    }
    if (d.inputs_ready) DF3D_HIP(hipStreamWaitEvent(g_img_stream, (hipEvent_t)d.inputs_ready, 0));
    int rc = df3d_imgproj_split(d.img_ptrs, d.img_count, d.img_cin, d.img_pixels, d.img_packed, t.out.img_split, t.out.img_gate,
                                (void *)g_img_stream);
    if (rc) return rc;
    DF3D_HIP(hipEventRecord(S->img_done, g_img_stream));
    t.out.img_done = S->img_done;
  }
  if (total > 0 && d.nlayers > 0) {
    // ---- 2. the backbone's geometry ----
    S->gmem = &t.mem;
    int rc = state_init(*S, d.layers, d.nlayers, ind, (int)total, C, B, d.shape);
    HEAD_PHASE("voxel counts read");
    if (dbg) fprintf(stderr, "df3d frame head: n = %lld, arena used %zu of %zu\n", total, t.mem.used, t.arena_bytes);
    for (int li = 0; li < d.nlayers && !rc; ++li) {
      bool new_table = false;
      rc = geo_layer(*S, li, gs, &new_table);
      if (dbg) {
        char b[64];
        snprintf(b, sizeof(b), "geometry layer %d rc %d", li, rc);
        HEAD_PHASE(b);
      }
    }
    if (rc) return rc;
    t.views.assign(d.nlayers, df3d_layer_view());
    for (int li = 0; li < d.nlayers; ++li) {
      memset(&t.views[li], 0, sizeof(df3d_layer_view));
      view_geometry(*S, li, t.views[li]);
    }
    // ---- 3. camera projection of the stages the adapter reads, query slots of the last one ----
    if (d.ncam > 0) {
      DF3D_CHECK_ARG(d.nproj >= 0 && d.nproj <= DF3D_HEAD_MAX_PROJ && d.slots_proj < d.nproj, "frame_head: bad projection list");
      for (int p = 0; p < d.nproj; ++p) {
        const int li = d.proj[p].layer;
        DF3D_CHECK_ARG(li >= 0 && li < d.nlayers, "frame_head: projection %d reads layer %d", p, li);
        const IndexSet &T = S->sets[S->outs[li].set];
        HEAD_TAKE(t.out.grid_xy[p], int32_t *, (size_t)d.ncam * T.n * 8);
        HEAD_TAKE(t.out.mask[p], uint8_t *, (size_t)d.ncam * T.n);
        HEAD_TAKE(t.out.point_inv[p], float *, (size_t)T.n * 12);
        t.out.proj_n[p] = T.n;
        rc = df3d_project_voxels(T.indices, T.n, B, d.ncam, d.proj[p].scale_xyz, d.pc_min, d.lidar2cam, d.intrinsic, d.raw_hw,
                                 d.depth_thres, d.image_scale, d.feat_scale, t.out.grid_xy[p], t.out.mask[p],
                                 t.out.point_inv[p], nullptr, d.aug_inv, gs_);
        if (rc) return rc;
        if (d.proj[p].want_winner && d.feat_h > 0 && d.feat_w > 0) {
          HEAD_TAKE(t.out.winner[p], int32_t *, (size_t)B * d.ncam * d.feat_h * d.feat_w * 4);
          rc = df3d_scatter_winner(T.indices, t.out.grid_xy[p], t.out.mask[p], T.n, B, d.ncam, d.feat_h, d.feat_w,
                                   t.out.winner[p], gs_);
          if (rc) return rc;
        }
      }
      if (d.slots_proj >= 0) {
        const int p = d.slots_proj;
        const IndexSet &T = S->sets[S->outs[d.proj[p].layer].set];
        HEAD_TAKE(t.out.pos, int32_t *, (size_t)d.ncam * T.n * 4);
        HEAD_TAKE(t.out.counts, int32_t *, (size_t)B * d.ncam * 4);
        rc = df3d_query_slots(t.out.mask[p], T.indices, T.n, B, d.ncam, t.out.pos, t.out.counts, gs_);
        if (rc) return rc;
        int32_t *ptot = nullptr;
        if (d.want_pixrow && d.feat_h > 0 && d.feat_w > 0) {   // pixels that carry a query, ranked (df3d_query_pixel_rows)
          const size_t wsb = df3d_query_pixel_rows_workspace_bytes(B, d.ncam, d.feat_h, d.feat_w);
          void *ws;
          HEAD_TAKE(t.out.pixrow, int32_t *, (size_t)B * d.ncam * d.feat_h * d.feat_w * 4);
          HEAD_TAKE(ptot, int32_t *, 256);
          HEAD_TAKE(ws, void *, wsb);
          rc = df3d_query_pixel_rows(T.indices, t.out.grid_xy[p], t.out.mask[p], T.n, B, d.ncam, d.feat_h, d.feat_w, t.out.pixrow,
                                     ptot, ws, wsb, gs_);
          if (rc) return rc;
        }
        std::vector<int32_t> qc((size_t)B * d.ncam, 0);
        DF3D_HIP(hipMemcpyAsync(qc.data(), t.out.counts, qc.size() * 4, hipMemcpyDeviceToHost, gs));
        int32_t tot = 0;
        if (ptot) DF3D_HIP(hipMemcpyAsync(&tot, ptot, 4, hipMemcpyDeviceToHost, gs));
        DF3D_HIP(hipStreamSynchronize(gs));               // the adapter's one host round trip: the longest camera list
        t.out.pixrow_total = tot;
        int mx = 0;
        for (int v : qc) mx = v > mx ? v : mx;
        t.out.max_ne = mx;
      }
    }
  }
  HEAD_PHASE("projection / slots");
  DF3D_HIP(hipEventRecord(S->done, gs));
  return DF3D_OK;
}

void head_worker_main(HeadWorker *w) {
  (void)hipSetDevice(w->device);
  for (;;) {
    HeadTicket *t = nullptr;
    {
      std::unique_lock<std::mutex> lk(w->mu);
      w->cv.wait(lk, [&] { return w->stop || !w->q.empty(); });
      if (w->q.empty()) return;
      t = w->q.front();
      w->q.pop_front();
    }
    int rc = frame_head_run(*t);
    {
      // notify while holding the lock: df3d_frame_head_wait deletes the ticket as soon as it sees `done`, and it can only
      // see it after this scope has released the mutex -- i.e. after the notification (a notify behind the
      // unlock could run on a destroyed condition variable)
      std::lock_guard<std::mutex> lk(t->mu);
      t->rc = rc;
      if (rc) t->err = df3d_last_error();
      t->done = true;
      t->cv.notify_all();
    }
  }
}

}  // namespace

extern "C" void *df3d_head_worker_create(int device) {
  HeadWorker *w = new HeadWorker();
  w->device = device;
  w->th = std::thread(head_worker_main, w);
  return w;
}

extern "C" int df3d_head_worker_destroy(void *worker) {
  if (!worker) return DF3D_OK;
  HeadWorker *w = (HeadWorker *)worker;
  {
    std::lock_guard<std::mutex> lk(w->mu);
    w->stop = true;
  }
  w->cv.notify_all();
  if (w->th.joinable()) w->th.join();
  for (HeadTicket *t : w->q) delete t;
  delete w;
  return DF3D_OK;
}

extern "C" int df3d_frame_head_submit(void *worker, const df3d_frame_head_desc *desc, void *arena, size_t arena_bytes,
                                      void **ticket) {
  DF3D_CHECK_ARG(worker && desc && arena && ticket, "frame_head_submit: null argument");
  DF3D_CHECK_ARG(desc->batch > 0 && desc->points && desc->num_points && desc->point_channels > 0,
                 "frame_head_submit: no point clouds");
  DF3D_CHECK_ARG(desc->nlayers == 0 || desc->layers, "frame_head_submit: layer table missing");
  HeadTicket *t = new HeadTicket();
  t->d = *desc;
  t->points.assign(desc->points, desc->points + desc->batch);
  t->num_points.assign(desc->num_points, desc->num_points + desc->batch);
  t->d.points = nullptr, t->d.num_points = nullptr;          // the caller's arrays are not referenced after this call
  t->mem = Bump(arena, arena_bytes);
  t->arena_bytes = arena_bytes;
  HeadWorker *w = (HeadWorker *)worker;
  {
    std::lock_guard<std::mutex> lk(w->mu);
    w->q.push_back(t);
  }
  w->cv.notify_one();
  *ticket = t;
  return DF3D_OK;
}

extern "C" int df3d_frame_head_wait(void *ticket, df3d_layer_view *views, df3d_frame_head_out *out, size_t *arena_used,
                                    void **handle) {
  DF3D_CHECK_ARG(ticket && out && handle, "frame_head_wait: null argument");
  HeadTicket *t = (HeadTicket *)ticket;
  {
    std::unique_lock<std::mutex> lk(t->mu);
    t->cv.wait(lk, [&] { return t->done; });
  }
  *handle = nullptr;
  if (arena_used) *arena_used = t->mem.used;
  int rc = t->rc;
  if (rc) {
    if (rc == DF3D_ENOMEM) set_error("frame_head: arena of %zu bytes is too small (needed more than %zu)", t->arena_bytes, t->mem.used);
    else set_error("%s", t->err.c_str());
    if (t->state) {
      if (t->state->done) (void)hipEventDestroy(t->state->done);
      if (t->state->img_done) (void)hipEventDestroy(t->state->img_done);
      delete t->state;
    }
    delete t;
    return rc;
  }
  *out = t->out;
  if (views && !t->views.empty()) memcpy(views, t->views.data(), t->views.size() * sizeof(df3d_layer_view));
  RunState *S = t->state;
  S->gmem = nullptr;                       // the ticket's allocator goes away; the geometry phase is over
  *handle = S;
  delete t;
  return DF3D_OK;
}
