"""Measurement aid (bench.py's `roofline_by_kernel`): HIP events around EVERY call into libdf3d_hip.so, recorded on the
stream the call launches its kernels on (the last argument of every launching entry point), plus the byte / flop cost
models of SURVEY.md section 8(d) for the entry points of the hot path.

    t = ApiTimer(); t.start(); step(); records = t.stop()

While started, `_lib.load()` hands out a proxy of the ctypes library; nothing else changes.  The sparse / dense
convolution entries are left to `ops.KernelTimer` (events inside the library, incl. the launches of the native executor,
with the rulebook pair counts their cost model needs)."""
import ctypes

from . import _lib

_NO_STREAM = ("df3d_version", "df3d_last_error", "df3d_device_count", "df3d_device_arch", "df3d_timing_", "df3d_ffn_set_precision",
              "df3d_conv_tile_count", "df3d_bn_rows_supported", "df3d_bn_rows_scratch_doubles", "df3d_debug_")
CONV_ENTRIES = ("df3d_sparse_conv_fused", "df3d_sparse_conv_grouped", "df3d_sparse_conv_fused_tiled", "df3d_sparse_conv_split", "df3d_sparse_conv_bf16",
                "df3d_conv_rows_split", "df3d_backbone_run")


def _ops_stream():
    from . import ops
    return ops._stream()


def _ival(a):
    if isinstance(a, bool):
        return int(a)
    if isinstance(a, int):
        return a
    if isinstance(a, (ctypes.c_int, ctypes.c_longlong, ctypes.c_size_t, ctypes.c_uint)):
        return int(a.value)
    if isinstance(a, float):
        return a
    if isinstance(a, ctypes.c_float):
        return float(a.value)
    return None


class _Proxy(object):
    def __init__(self, real, timer):
        self.__dict__["_real"], self.__dict__["_timer"], self.__dict__["_cache"] = real, timer, {}

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if not name.startswith("df3d_") or name.endswith("_bytes") or any(name.startswith(p) for p in _NO_STREAM):
            return fn
        w = self._cache.get(name)
        if w is None:
            timer = self._timer

            def w(*args, _fn=fn, _name=name):
                return timer._call(_name, _fn, args)
            self._cache[name] = w
        return w


class ApiTimer(object):
    def __init__(self):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
        self.hip.hipDeviceSynchronize.argtypes = []
        self._raw, self.records, self._real = [], [], None

    def _event(self):
        e = ctypes.c_void_p()
        if self.hip.hipEventCreate(ctypes.byref(e)) != 0:
            raise _lib.Df3dError("hipEventCreate failed")
        return e

    def _call(self, name, fn, args):
        stream = args[-1] if args else None
        if isinstance(stream, int):
            stream = ctypes.c_void_p(stream)
        # only calls whose last argument IS torch's current stream are bracketed (an entry point without a stream argument
        # ends in some other pointer: recording an event "on" it would be a wild pointer dereference inside the runtime)
        if not isinstance(stream, ctypes.c_void_p) or (stream.value or 0) != (_ops_stream().value or 0):
            return fn(*args)
        extra = None
        if name == "df3d_ffn_fused_jobs":                       # rows live in the job structs
            try:
                arr = getattr(args[0], "_obj", args[0])        # ctypes.byref(array) keeps the array in ._obj
                extra = sum(int(arr[i].rows) for i in range(int(_ival(args[1]))))
            except Exception:                                   # noqa: BLE001
                extra = None
        e0, e1 = self._event(), self._event()
        self.hip.hipEventRecord(e0, stream)
        rc = fn(*args)
        self.hip.hipEventRecord(e1, stream)
        self._raw.append((name, e0, e1, tuple(_ival(a) for a in args), extra))
        return rc

    def start(self):
        self._real = _lib.load()
        _lib._lib = _Proxy(self._real, self)
        return self

    def stop(self):
        _lib._lib = self._real
        self.hip.hipDeviceSynchronize()
        ms = ctypes.c_float()
        for name, e0, e1, ints, extra in self._raw:
            self.hip.hipEventElapsedTime(ctypes.byref(ms), e0, e1)
            self.records.append(dict(name=name, ms=float(ms.value), args=ints, extra=extra))
            self.hip.hipEventDestroy(e0)
            self.hip.hipEventDestroy(e1)
        self._raw = []
        return self.records


# ------------------------------------------------------------------------------------------------ cost models
# SURVEY.md section 8(d).  Each returns (bytes, flops, bound, what) from the entry point's integer arguments
# (positions as declared in include/df3d_hip.h); None where a quantity is unknown at the call.
def _msda(a, value_bytes=4):
    N, S, M, D, Lq, L, P = a[10:17]
    by = min(N * S * M * D, N * Lq * M * L * P * 4 * D) * value_bytes + N * Lq * M * L * P * 12 + N * Lq * M * D * 4
    return by, 10 * N * Lq * M * L * P * D, "hbm", "bilinear sampling + weighted sum of %d x %d queries, %d heads x %d points" % (N, Lq, M, L * P)


def _ffn(a, rows):
    d_model, d_ffn = a[2], a[3]
    if rows is None:
        return None, None, "mfma", "fused feed-forward"
    return rows * d_model * 8 + 2 * d_model * d_ffn * 4, 4 * rows * d_model * d_ffn, "mfma", "LayerNorm(x + W2 relu(W1 x)) of %d query rows" % rows


def _imgproj(a):
    nimg, cin, S = a[1], a[2], a[3]
    return nimg * S * (cin * 4 + 512 + 4), 2 * nimg * S * cin * 144, "hbm", "u = Wcat[144 x %d] img of %d maps x %d pixels -> split rows + gate" % (cin, nimg, S)


def _value_gemm(a, out_bytes=4):
    nimg, S = a[2], a[3]
    return nimg * S * (512 * 2 + 256 * out_bytes), 2 * nimg * S * 128 * 256 + 4 * nimg * S * 128, "hbm", "GroupNorm moments + folded value GEMM [%d x %d px, 128] -> 256" % (nimg, S)


def _assemble(a):
    n, C, Ci, B, ncam, H, W, max_ne = a[9:17]
    return B * ncam * max_ne * (C + Ci + 5) * 4, 0, "hbm", "gather of %d x %d query slots: %d voxel + %d image channels" % (B * ncam, max_ne, C, Ci)


def _rows(elems_in, elems_out, what):
    return (elems_in + elems_out) * 4, 0, "hbm", what


MODELS = {
    "df3d_hard_voxelize": lambda a, x: (20 * a[1] + a[6] * 0, 0, "hbm", "hash + scan + gather of %d points (bytes completed by the caller with M * 36)" % a[1]),
    "df3d_hard_voxelize_batched": lambda a, x: (20 * a[1], 0, "hbm", "hash + scan + gather of %d points (bytes completed by the caller with M * 36)" % a[1]),
    "df3d_ms_deform_attn_fused": lambda a, x: _msda(a, 4),
    "df3d_ms_deform_attn_fused_bf16": lambda a, x: _msda(a, 2),
    "df3d_ffn_fused_jobs": lambda a, x: _ffn(a, x),
    "df3d_imgproj_split": lambda a, x: _imgproj(a),
    "df3d_value_fold_gemm": lambda a, x: _value_gemm(a, 4),
    "df3d_value_fold_gemm_bf16": lambda a, x: _value_gemm(a, 2),
    "df3d_assemble_queries2": lambda a, x: _assemble(a),
    "df3d_sparse_to_dense": lambda a, x: (a[2] * a[3] * 4, 0, "hbm", "dense(): %d rows x %d channels scattered (+ the zero fill of the dense tensor)" % (a[2], a[3])),
    "df3d_sparse_to_dense_rows": lambda a, x: (a[2] * a[3] * 4, 0, "hbm", "dense() as pixel rows: %d rows x %d channels (+ the zero fill)" % (a[2], a[3])),
    "df3d_head_final_conv": lambda a, x: (a[2] * a[3] * a[4] * (a[1] * 4 + a[10] * 4), 2 * a[2] * a[3] * a[4] * a[5] * 9 * 64 * 4, "hbm",
                                          "final 3x3 convs of %d head branches on %d pixels" % (a[5], a[2] * a[3] * a[4])),
    "df3d_head_final_conv_packed": lambda a, x: (a[2] * a[3] * a[4] * (a[1] * 4 + a[10] * 4), 2 * a[2] * a[3] * a[4] * a[5] * 9 * 64 * 4, "hbm",
                                                 "final 3x3 convs of %d head branches on %d pixels (taps as matrix-core columns)" % (a[5], a[2] * a[3] * a[4])),
    "df3d_split_rows": lambda a, x: _rows(a[1] * a[2], a[1] * a[2], "fp32 rows -> split rows (%d x %d)" % (a[1], a[2])),
    "df3d_actr_prep": lambda a, x: _rows(3 * a[3] * a[4], 2 * a[3] * a[4], "query + position sums (%d rows)" % a[3]),
    "df3d_add_layernorm": lambda a, x: _rows(2 * a[5] * a[6], a[5] * a[6], "add + LayerNorm (%d rows)" % a[5]),
    "df3d_add_layernorm_split": lambda a, x: _rows(2 * a[5] * a[6], 2 * a[5] * a[6], "add + LayerNorm + split rows (%d rows)" % a[5]),
    "df3d_bigate_sum": lambda a, x: _rows(2 * a[6] * a[7], 2 * a[6] * a[7], "bidirectional gate (%d rows)" % a[6]),
    "df3d_rows_groupnorm": lambda a, x: _rows(2 * a[1] * a[2] * a[3], a[1] * a[2] * a[3], "GroupNorm over query rows (two passes)"),
    "df3d_fusion_writeback": lambda a, x: _rows(a[5] * a[6] + a[7] * a[8] * a[6], a[5] * a[6], "additive write-back of %d voxels" % a[5]),
    "df3d_rows_linear": lambda a, x: (a[3] * (a[4] + a[6]) * 4 + a[4] * a[6] * 4, 2 * a[3] * a[4] * a[6], "hbm",
                                      "query linear %d -> %d over %d rows (one input stream counted)" % (a[4], a[6], a[3])),
    "df3d_lt_layer": lambda a, x: (2 * a[1] * a[2] * a[3] * 4, a[1] * a[2] * 2 * (3 * 64 * 64 + 64 * 64 + 2 * 128 * 64 + 2 * 32 * 64), "hbm",
                                   "LocalTransformer layer over %d groups of %d tokens" % (a[2], a[1])),
    "df3d_lt_layer_gather": lambda a, x: (2 * 32 * a[3] * 64 * 4 + 32 * a[3] * 20, 32 * a[3] * 2 * (3 * 64 * 64 + 64 * 64 + 2 * 128 * 64 + 2 * 32 * 64 + 32 * 64), "hbm",
                                          "LocalTransformer layer with gather + positional MLP, %d groups" % a[3]),
    "df3d_lt_layer_scatter": lambda a, x: (2 * 32 * a[1] * 64 * 4 + 32 * a[1] * 8, 32 * a[1] * 2 * (3 * 64 * 64 + 64 * 64 + 2 * 128 * 64 + 2 * 32 * 64), "hbm",
                                           "LocalTransformer layer with winner write-back, %d groups" % a[1]),
    "df3d_voxel_image_sample": lambda a, x: (a[1] * (16 + a[9] * 4 * 5), 0, "hbm", "point-fusion sampling of %d voxels x %d channels" % (a[1], a[9])),
    "df3d_project_voxels": lambda a, x: (a[1] * 16 + a[1] * a[3] * (8 + 1 + 12 + 4), 0, "hbm", "projection of %d voxels into %d cameras" % (a[1], a[3])),
    "df3d_query_slots": lambda a, x: (a[2] * a[4] * 5 + a[2] * 16, 0, "hbm", "per-camera query slots of %d voxels" % a[2]),
}
RULEBOOK_ENTRIES = ("df3d_grid_build", "df3d_subm_neighbors", "df3d_conv_out_indices", "df3d_conv_neighbors",
                    "df3d_conv_transpose_out_indices", "df3d_conv_transpose_neighbors", "df3d_conv2d_neighbors", "df3d_nbr_to_pairs",
                    "df3d_pairs_to_nbr", "df3d_invert_neighbors")


def summarize(records):
    """records of ONE step -> {entry: dict(ms, calls, bytes, flops, bound, what)} (convolution entries excluded)."""
    out = {}
    for r in records:
        name = r["name"]
        if name in CONV_ENTRIES:
            continue
        key = "rulebook" if name in RULEBOOK_ENTRIES else name
        e = out.setdefault(key, dict(ms=0.0, calls=0, bytes=0, flops=0, bound=None, what=None, unknown=False))
        e["ms"] += r["ms"]
        e["calls"] += 1
        m = MODELS.get(name)
        if m is not None:
            try:
                by, fl, bound, what = m(r["args"], r["extra"])
            except Exception:                                   # noqa: BLE001  (an argument was not an integer)
                by = fl = None
                bound = what = None
            if by is None:
                e["unknown"] = True
            else:
                e["bytes"] += by
                e["flops"] += fl or 0
            e["bound"], e["what"] = bound, what
        elif key == "rulebook":
            e["bound"], e["what"] = "hbm", "occupancy directories, output index sets and neighbour tables (bytes from the pair counts)"
        else:
            e["unknown"] = True
    return out
