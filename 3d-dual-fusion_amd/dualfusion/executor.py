"""Native execution of a sparse backbone's convolution chain (csrc/executor.hip, df3d_backbone_run).

The module tree stays the API (same classes, parameter names and checkpoints as the reference); for inference the
chain of SparseSequential / SubMConv3d / SparseConv3d / BatchNorm1d / ReLU / basic-block modules is flattened once
into a layer table and every forward is ONE native call instead of ~40 us of interpreter work per layer.  Anything
the table cannot express (training mode, 1x1 convs, down-sampled residuals, unknown modules) makes `compile`
return None and the caller keeps the per-module path; both paths launch the same kernels, so the outputs are
bit-identical."""
import ctypes
import os
import threading
import time

import torch
from torch import nn

from . import _lib
from . import ops as _ops
from .spconv.conv import SparseConvolution
from .spconv.modules import SparseSequential, can_fold, fold_batchnorm
from .spconv.structure import DirectoryCache, SparseConvTensor


class _Layer(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("input", ctypes.c_int), ("residual", ctypes.c_int),
                ("rulebook", ctypes.c_int), ("cin", ctypes.c_int), ("cout", ctypes.c_int),
                ("ksize", ctypes.c_int * 3), ("stride", ctypes.c_int * 3), ("padding", ctypes.c_int * 3),
                ("dilation", ctypes.c_int * 3), ("relu", ctypes.c_int), ("reserved", ctypes.c_int),
                ("weight", ctypes.c_void_p), ("packed", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p)]


class _View(ctypes.Structure):
    _fields_ = [("features", ctypes.c_void_p), ("split", ctypes.c_void_p), ("indices", ctypes.c_void_p),
                ("grid", ctypes.c_void_p), ("grid_bytes", ctypes.c_size_t), ("n", ctypes.c_int),
                ("channels", ctypes.c_int), ("rows_sorted", ctypes.c_int), ("shape", ctypes.c_int * 3),
                ("nbr", ctypes.c_void_p), ("kvol", ctypes.c_int), ("reserved", ctypes.c_int)]


class Unsupported(Exception):
    pass


class _Spec(object):
    """One fused layer: conv + optional folded BN + optional residual + optional ReLU."""

    def __init__(self, conv, bn, relu, inp, residual, rulebook):
        self.conv, self.bn, self.relu, self.input, self.residual, self.rulebook = conv, bn, relu, inp, residual, rulebook
        self.geometry_only = False


def _is_basic_block(m):
    return all(hasattr(m, a) for a in ("conv1", "bn1", "conv2", "bn2")) and hasattr(m, "downsample")


class BackbonePlan(object):
    def __init__(self, stages, geometry_stages=()):
        """stages: list of (name, module); the output of every stage is exported under its name.
        geometry_stages: [(name of the stage it reads, module)]: modules that run AFTER the caller has modified that
        stage's features (a fusion step): only their rulebooks are built here (geometry depends on the coordinates
        alone) and handed to the module path, which then needs no host round trip."""
        self.specs = []
        self.exports = {}
        self.geometry = []         # (input stage name, layer index)
        self._keys = {}
        self._set = 0              # index-set counter (changes at every strided conv)
        cur = -1
        for name, module in stages:
            cur = self._walk(module, cur)
            self.exports[name] = cur
        for src, module in geometry_stages:
            first = len(self.specs)
            self._walk(module, self.exports[src])
            if len(self.specs) != first + 1:
                raise Unsupported("geometry stage must be a single convolution group")
            self.specs[first].geometry_only = True
            self.geometry.append((src, first))
        self._table = None
        self._sig = None
        self._keep = None
        self._arena_bytes = 256 << 20
        self._ring = []            # decoupled runs: up to three persistent [arena, event recorded at the start of the NEXT run]
        self._ring_last = 0
        self._geo_bytes = 384 << 20
        self._frames = _FrameRing()      # build_geometry() / run_convs(): arenas of the frames in flight
        self._lock = threading.Lock()

    # ---------------------------------------------------------------- module tree -> layer specs
    def _rulebook_id(self, conv):
        if conv.subm:
            key = ("subm", conv.indice_key if conv.indice_key is not None else ("auto", self._set),
                   tuple(conv.kernel_size), tuple(conv.dilation))
        else:
            if conv.indice_key is None:
                return -1
            key = ("sparse", conv.indice_key)
        return self._keys.setdefault(key, len(self._keys))

    def _add(self, conv, bn, relu, inp, residual):
        if not isinstance(conv, SparseConvolution) or conv.ndim != 3 or conv.conv1x1 or conv.inverse \
                or conv.transposed or conv.training:
            raise Unsupported("conv %r" % (conv,))
        if bn is not None and not can_fold(bn):
            raise Unsupported("BatchNorm in training mode")
        rb = self._rulebook_id(conv)
        self.specs.append(_Spec(conv, bn, relu, inp, residual, rb))
        if not conv.subm:
            self._set += 1
        return len(self.specs) - 1

    def _walk(self, m, cur):
        if isinstance(m, SparseSequential):
            mods = list(m._modules.values())
            i = 0
            while i < len(mods):
                c = mods[i]
                if isinstance(c, SparseConvolution):
                    bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm1d) else None
                    j = i + (2 if bn is not None else 1)
                    relu = j < len(mods) and isinstance(mods[j], nn.ReLU)
                    cur = self._add(c, bn, relu, cur, -1)
                    i = j + (1 if relu else 0)
                elif isinstance(c, (nn.Identity, nn.Dropout)):
                    i += 1
                else:
                    cur = self._walk(c, cur)
                    i += 1
            return cur
        if _is_basic_block(m):
            if m.downsample is not None:
                raise Unsupported("basic block with a downsample branch")
            a = self._add(m.conv1, m.bn1, True, cur, -1)
            if cur < 0:
                raise Unsupported("basic block on the network input")
            return self._add(m.conv2, m.bn2, True, a, cur)
        if isinstance(m, SparseConvolution):
            return self._add(m, None, False, cur, -1)
        raise Unsupported("module %s" % type(m).__name__)

    # ---------------------------------------------------------------- layer specs -> C table
    def _signature(self):
        """Change detector for the cached table: (storage address, version counter) of EVERY tensor the table is folded
        from -- filters, conv biases, BatchNorm weight / bias / running_mean / running_var.  Version counters are bumped by
        every in-place update (load_state_dict, optimizer steps, `bn.bias.add_()` under no_grad ...); addresses catch re-assigned
        parameters."""
        slots = self.__dict__.get("_sig_slots")
        if slots is None:
            # (dict, key) of every tensor, looked up in the modules' own _parameters / _buffers dicts: this runs twice per frame
            # and nn.Module.__getattr__ costs more than the check itself
            slots = []
            for s in self.specs:
                slots += [(s.conv._parameters, "weight"), (s.conv._parameters, "bias")]
                if s.bn is not None:
                    slots += [(s.bn._parameters, "weight"), (s.bn._parameters, "bias"),
                              (s.bn._buffers, "running_mean"), (s.bn._buffers, "running_var")]
            self.__dict__["_sig_slots"] = slots
        sig = []
        for d, k in slots:
            t = d.get(k)
            sig.append(None if t is None else (t.data_ptr(), t._version))
        return (tuple(sig), _ops.CONV_PRECISION)

    def _build_table(self):
        n = len(self.specs)
        table = (_Layer * n)()
        keep = []
        exported = set(self.exports.values())
        for i, s in enumerate(self.specs):
            c = s.conv
            L = table[i]
            L.kind = 0 if c.subm else 1
            L.input, L.residual, L.rulebook = s.input, s.residual, s.rulebook
            L.cin, L.cout = c.in_channels, c.out_channels
            for d in range(3):
                L.ksize[d], L.stride[d] = int(c.kernel_size[d]), int(c.stride[d])
                L.padding[d], L.dilation[d] = int(c.padding[d]), int(c.dilation[d])
            if c.subm:                       # SubM geometry: stride 1, pad = k/2 whatever the module was given
                for d in range(3):
                    L.stride[d], L.padding[d] = 1, int(c.kernel_size[d]) // 2
            L.relu = int(bool(s.relu))
            L.reserved = 1 if s.geometry_only else 0
            if i in exported:
                L.reserved |= 4                      # fp32 rows of this layer are read by the caller
            K = int(c.kernel_size[0] * c.kernel_size[1] * c.kernel_size[2])
            w = c.weight.detach().contiguous().view(K, c.in_channels, c.out_channels)
            if w.dtype != torch.float32:
                raise Unsupported("non-fp32 weights")
            keep.append(w)
            L.weight = w.data_ptr()
            L.packed = None
            if _ops.CONV_PRECISION == "bf16" and _ops.conv_bf16_supported(K, c.in_channels, c.out_channels):
                p = c._packed_weight_bf16(c.weight.detach(), K)
                keep.append(p)
                L.packed = p.data_ptr()
                L.reserved |= 2                      # bf16 rows / weights for this layer
            elif _ops.conv_split_supported(K, c.in_channels, c.out_channels):
                p = c._packed_weight(c.weight.detach(), K)
                keep.append(p)
                L.packed = p.data_ptr()
                if _ops.CONV_PRECISION == "split3":
                    L.reserved |= 8                  # three-part rows / filters
            if c.bias is not None:
                L.bias = c.bias.detach().data_ptr()
            if s.bn is not None:
                scale, shift = fold_batchnorm(s.bn)
                keep += [scale, shift]
                L.scale, L.shift = scale.data_ptr(), shift.data_ptr()
        self._table, self._keep = table, keep

    # ---------------------------------------------------------------- run
    def run(self, features, coors, batch_size, spatial_shape):
        """-> {stage name: SparseConvTensor} whose tensors are views into one arena."""
        lib = _lib.load()
        sig = self._signature()
        if self._table is None or sig != self._sig:
            self._build_table()
            self._sig = sig
        feats = features.contiguous()
        if feats.dtype != torch.float32:
            feats = feats.float()
        coors = coors.contiguous()
        n = feats.shape[0]
        nl = len(self.specs)
        views = (_View * nl)()
        used = ctypes.c_size_t(0)
        shp = (ctypes.c_int * 3)(*[int(v) for v in spatial_shape])
        # Resident inputs (DF3D_EXEC_DECOUPLE=0 switches this off): coordinates produced on the voxeliser's own stream carry the event that marks them complete; the
        # geometry stream then waits for that event only (df3d_backbone_inputs_ready) and the rulebooks of this frame (with the
        # host's round trips for the output counts) proceed while the previous frame still runs on the current stream.  The
        # geometry stream writes into the arena WITHOUT being ordered behind the current stream, so in this mode the arena must
        # never be a block the caching allocator recycles (a freed tensor of the previous frame whose kernels are still queued
        # would be handed out as "free" -- found as garbage output counts and memory faults): three PERSISTENT arenas are
        # used in rotation, each guarded by an event recorded when the run after its own starts; a stage tensor of frame k
        # therefore stays valid until frame k + 3 starts.
        ready = getattr(coors, "_df3d_ready", None) if os.environ.get("DF3D_EXEC_DECOUPLE", "1") == "1" else None
        slot = None
        if ready is not None:
            mark = torch.cuda.current_stream(feats.device).record_event()
            if self._ring and self._ring[self._ring_last][1] is None:
                self._ring[self._ring_last][1] = mark
            if len(self._ring) < 3:
                self._ring.append([None, None])
                slot = len(self._ring) - 1
            else:
                slot = (self._ring_last + 1) % 3
            if self._ring[slot][1] is not None:
                self._ring[slot][1].synchronize()              # the frame that used this arena three runs ago has finished
                self._ring[slot][1] = None
        while True:
            if slot is None:
                arena = torch.empty((self._arena_bytes,), dtype=torch.uint8, device=feats.device)
            else:
                if self._ring[slot][0] is None or self._ring[slot][0].numel() < self._arena_bytes:
                    torch.cuda.synchronize(feats.device)       # (re)allocation: rare, and the old block must be idle
                    self._ring[slot][0] = torch.empty((self._arena_bytes,), dtype=torch.uint8, device=feats.device)
                arena = self._ring[slot][0]
                lib.df3d_backbone_inputs_ready(ctypes.c_void_p(ready.cuda_event))
            rc = lib.df3d_backbone_run(self._table, nl, _ops._ptr(feats), _ops._ptr(coors), n, feats.shape[1],
                                       int(batch_size), shp, _ops._ptr(arena), self._arena_bytes, views,
                                       ctypes.byref(used), _ops._stream())
            if rc == _lib.DF3D_ENOMEM and self._arena_bytes < (64 << 30):
                self._arena_bytes *= 2
                if ready is not None:
                    torch.cuda.synchronize(feats.device)   # the failed attempt's geometry kernels still write the small arena
                continue
            _lib.check(rc, "df3d_backbone_run")
            break
        if slot is not None:
            self._ring_last = slot
        return self._export(views, (arena,), coors, batch_size)

    def _export(self, views, arenas, coors, batch_size, features=True, only=None, state=None):
        """views (filled by the native call) -> {stage name: SparseConvTensor} whose tensors are views into the arenas.
        features=False: the geometry only (index tensors, directories); the feature tensors are attached by `run_convs`.
        only = (first, last): the stages whose layer lies in that range (`run_convs_range`); `state`: the (indice_dict,
        directories) pair the stages of one frame share."""
        spans = [(a.data_ptr(), a.data_ptr() + a.numel(), a) for a in arenas]

        def view(ptr, nbytes, dtype, shape):
            for lo, hi, a in spans:
                if lo <= ptr < hi:
                    off = ptr - lo
                    return a[off:off + nbytes].view(dtype).view(shape)
            raise _lib.Df3dError("executor: a view points outside the arenas")

        out = {}
        idict, dirs = state if state is not None else ({}, DirectoryCache())
        for name, li in self.exports.items():
            if only is not None and not (only[0] <= li < only[1]):
                continue
            v = views[li]
            # SubM stages of the first index set keep the caller's index tensor
            ind = coors if v.indices == coors.data_ptr() else view(v.indices, v.n * 16, torch.int32, (v.n, 4))
            f = view(v.features, v.n * v.channels * 4, torch.float32, (v.n, v.channels)) if features else None
            t = SparseConvTensor(f, ind, [v.shape[0], v.shape[1], v.shape[2]], batch_size)
            t.indice_dict, t._directories = idict, dirs
            t._indices_synced = True      # the host has waited for these coordinates (count round trip)
            if features and v.split and (v.reserved & 2):
                t._bf16 = (f, view(v.split, v.n * v.channels * 2, torch.bfloat16, (v.n, v.channels)))
            elif features and v.split:
                pb = 6 if (v.reserved & 4) else 4
                t._split = (f, view(v.split, v.n * v.channels * pb, torch.uint8, (v.n, v.channels * pb)))
            if v.grid and v.rows_sorted:
                # the occupancy directory the executor built is handed to the module path (e.g. a conv that runs
                # after a fusion step): SparseConvTensor.directory() finds it by the identity of `indices`
                blob = view(v.grid, v.grid_bytes, torch.uint8, (v.grid_bytes,))
                dirs.put(ind, _ops.GridDirectory(blob, None, batch_size, list(t.spatial_shape)))
            out[name] = t
        for src, li in self.geometry:
            if src not in out:
                continue
            v = views[li]
            from .spconv.structure import Rulebook
            x = out[src]
            outids = view(v.indices, v.n * 16, torch.int32, (v.n, 4))
            nbr = view(v.nbr, v.kvol * v.n * 4, torch.int32, (v.kvol, v.n))
            rb = Rulebook(outids, x.indices, nbr, list(x.spatial_shape), [v.shape[0], v.shape[1], v.shape[2]],
                          out_rows_sorted=True)
            x._prebuilt[id(self.specs[li].conv)] = rb
        return out

    # ---------------------------------------------------------------- the run as two calls (geometry ahead of the convolutions)
    def build_geometry(self, coors, in_channels, batch_size, spatial_shape, ready=None):
        """Phase 1 (`df3d_backbone_geometry`): every rulebook / index set of the chain from the COORDINATES alone, on the calling
        thread's geometry stream, behind `ready` (event: coordinates complete).  Meant for a helper thread that works a frame
        ahead (dualfusion/prefetch.py): the call blocks for the strided layers' output counts.  -> PreparedGeometry whose
        `.stages` are SparseConvTensors WITHOUT features (indices / directories only); hand it to `run_convs`."""
        lib = _lib.load()
        with self._lock:
            sig = self._signature()
            if self._table is None or sig != self._sig:
                self._build_table()
                self._sig = sig
            table, keep = self._table, self._keep
            slot = self._frames.acquire()
        coors = coors.contiguous()
        n, nl = coors.shape[0], len(self.specs)
        views = (_View * nl)()
        used = ctypes.c_size_t(0)
        shp = (ctypes.c_int * 3)(*[int(v) for v in spatial_shape])
        handle = ctypes.c_void_p(0)
        while True:
            arena = slot.arena("geo", self._geo_bytes, coors.device)
            rc = lib.df3d_backbone_geometry(table, nl, _ops._ptr(coors), n, int(in_channels), int(batch_size), shp,
                                            _ops._ptr(arena), arena.numel(),
                                            ctypes.c_void_p(ready.cuda_event) if ready is not None else None, views,
                                            ctypes.byref(used), ctypes.byref(handle))
            if rc == _lib.DF3D_ENOMEM and self._geo_bytes < (64 << 30):
                self._geo_bytes *= 2
                torch.cuda.synchronize(coors.device)           # the failed attempt's kernels still write the small arena
                continue
            _lib.check(rc, "df3d_backbone_geometry")
            break
        geo = PreparedGeometry(self, handle, views, slot, coors, int(batch_size), (table, keep), sig)
        geo.spatial_shape = [int(v) for v in spatial_shape]
        geo.stages = self._export(views, (arena,), coors, batch_size, features=False)
        return geo

    def run_convs(self, geo, features):
        """Phase 2 (`df3d_backbone_convs`): the fused convolutions of a PreparedGeometry on the current stream.  Same result as
        `run` (the same kernels on the same tables)."""
        lib = _lib.load()
        if geo.plan is not self or geo.handle is None:
            raise _lib.Df3dError("executor.convs: the geometry belongs to another plan or was already consumed")
        feats = features.contiguous()
        if feats.dtype != torch.float32:
            feats = feats.float()
        if self._signature() != geo.sig:
            # the parameters changed between the two phases (an optimizer step, load_state_dict): the table the geometry call
            # saw is stale -- run the frame through the one-call path instead
            geo.release()
            return self.run(feats, geo.coors, geo.batch_size, geo.spatial_shape)
        mark = torch.cuda.current_stream(feats.device).record_event()
        self._frames.consumed(geo.slot, mark)
        used = ctypes.c_size_t(0)
        while True:
            arena = geo.slot.arena("feat", self._arena_bytes, feats.device)
            rc = lib.df3d_backbone_convs(geo.handle, _ops._ptr(feats), _ops._ptr(arena), arena.numel(), geo.views,
                                         ctypes.byref(used), _ops._stream())
            if rc == _lib.DF3D_ENOMEM and self._arena_bytes < (64 << 30):
                self._arena_bytes *= 2
                torch.cuda.synchronize(feats.device)
                continue
            _lib.check(rc, "df3d_backbone_convs")
            break
        out = self._export(geo.views, (arena, geo.slot.arenas["geo"]), geo.coors, geo.batch_size)
        geo.release()
        return out


    def run_convs_range(self, geo, features, first, last):
        """`df3d_backbone_convs_range`: layers [first, last) of a PreparedGeometry of THIS plan on the current stream; `features`
        = the network input (first == 0) or the rows that replace layer first - 1's output (a hook modified them).  Ranges are
        consecutive; the last one (last == len(specs)) releases the geometry handle.  -> {stage name: SparseConvTensor} of
        the stages inside the range, or None when the arena ran out in a LATER range (the caller finishes the frame on the
        one-call path; the arena is larger from the next frame on)."""
        lib = _lib.load()
        if geo.plan is not self or geo.handle is None:
            raise _lib.Df3dError("executor.convs_range: the geometry belongs to another plan or was already consumed")
        feats = features.contiguous()
        if feats.dtype != torch.float32:
            feats = feats.float()
        nl = len(self.specs)
        if first == 0:
            if self._signature() != geo.sig:
                geo.release()
                return None
            mark = torch.cuda.current_stream(feats.device).record_event()
            self._frames.consumed(geo.slot, mark)
            geo._range_state = ({}, DirectoryCache())
        used = ctypes.c_size_t(0)
        while True:
            arena = geo.slot.arena("feat", self._arena_bytes, feats.device)
            rc = lib.df3d_backbone_convs_range(geo.handle, _ops._ptr(feats), int(first), int(last), _ops._ptr(arena),
                                               arena.numel(), geo.views, ctypes.byref(used), _ops._stream())
            if rc == _lib.DF3D_ENOMEM and self._arena_bytes < (64 << 30):
                self._arena_bytes *= 2
                if first == 0:
                    torch.cuda.synchronize(feats.device)
                    continue
                geo.release()
                return None
            _lib.check(rc, "df3d_backbone_convs_range")
            break
        geo._range_keep = getattr(geo, "_range_keep", []) + [feats]        # the replaced rows are read by later launches
        out = self._export(geo.views, (arena, geo.slot.arenas["geo"]), geo.coors, geo.batch_size, only=(first, last),
                           state=geo._range_state)
        if last >= nl:
            geo.release()
        return out


class _FrameSlot(object):
    """Persistent arenas of one frame in flight + the event after which they may be rewritten."""

    def __init__(self):
        self.arenas = {}
        self.guard = None          # event recorded (on the consumer's stream) when the NEXT frame's convolutions start
        self.pending = False       # handed out, not yet guarded

    def arena(self, kind, nbytes, device):
        a = self.arenas.get(kind)
        if a is None or a.numel() < nbytes:
            # (Re)allocation: rare (the first frames).  The device must be idle: the block the caching allocator hands out may be
            # one it has just recycled from a stream whose kernels are still queued -- legal for work ordered behind that
            # stream, but these arenas are written by streams that are not (the frame-head worker's, other frames' streams)
            torch.cuda.synchronize(device)
            self.arenas[kind] = None
            with torch.cuda.stream(torch.cuda.default_stream(device)):
                a = self.arenas[kind] = torch.empty((nbytes,), dtype=torch.uint8, device=device)
        return a


class _FrameRing(object):
    """Frame slots in rotation.  Slot of frame k is rewritten by frame k + N: by then every reader of frame k has been
    QUEUED (the guard event is recorded on the consumer's stream when frame k + 1's convolutions start, i.e. after the host
    has queued everything of frame k) and the writer waits for that event on the host."""

    def __init__(self, n=4):
        self.slots = [_FrameSlot() for _ in range(n)]
        self.next = 0
        self.last_consumed = None
        self.blocked_s = 0.0       # host time spent waiting for a slot (the GPU is a whole ring behind)

    def acquire(self):
        slot = self.slots[self.next]
        self.next = (self.next + 1) % len(self.slots)
        if slot.guard is not None:
            if not slot.guard.query():                 # the host is ahead of the GPU by the whole ring: back-pressure, not work
                t0 = time.perf_counter()
                slot.guard.synchronize()
                self.blocked_s += time.perf_counter() - t0
        elif slot.pending:
            torch.cuda.synchronize()       # handed out and never guarded (a dropped frame): wait for everything
        slot.guard, slot.pending = None, True
        return slot

    def consumed(self, slot, mark):
        """`mark` was just recorded on the consumer's stream, BEFORE anything of `slot`'s frame is queued there: it guards the
        slot of the frame consumed before."""
        prev, self.last_consumed = self.last_consumed, slot
        if prev is not None and prev is not slot:
            prev.guard, prev.pending = mark, False


class PreparedGeometry(object):
    def __init__(self, plan, handle, views, slot, coors, batch_size, keep, sig):
        self.plan, self.handle, self.views, self.slot, self.coors = plan, handle, views, slot, coors
        self.batch_size, self.keep, self.sig = batch_size, keep, sig
        self.stages = None
        self.spatial_shape = None

    def wait(self):
        """The CURRENT stream waits (on the device) for the geometry: before anything other than `plan.run_convs` reads the index
        tensors of `.stages`."""
        _lib.check(_lib.load().df3d_backbone_geometry_wait(self.handle, _ops._stream()), "df3d_backbone_geometry_wait")

    def release(self):
        if self.handle is not None:
            _lib.load().df3d_backbone_release(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:             # noqa: BLE001  (interpreter shutdown)
            pass


def compile_stages(stages, geometry_stages=()):
    """BackbonePlan for [(name, module), ...] or None when the chain cannot be expressed."""
    try:
        return BackbonePlan(stages, geometry_stages)
    except Unsupported:
        return None


class SegmentedRunner(object):
    """Executor plans for a backbone whose conv chain is interrupted by feature-modifying hooks (fusion layers):
    the stages are cut into contiguous segments, each one native call; the hooks run between them on the module
    path.  `cuts` = indices i such that a hook runs AFTER stage i."""

    def __init__(self, stages, cuts=(), geometry_module=None):
        self.stages = list(stages)
        bounds = sorted(set(int(c) for c in cuts if 0 <= int(c) < len(self.stages) - 1))
        starts = [0] + [c + 1 for c in bounds]
        ends = [c + 1 for c in bounds] + [len(self.stages)]
        self.segments = []
        for k, (a, b) in enumerate(zip(starts, ends)):
            seg = self.stages[a:b]
            geo = [(seg[-1][0], geometry_module)] if (geometry_module is not None and b == len(self.stages)) else ()
            plan = compile_stages(seg, geometry_stages=geo)
            if plan is None:
                raise Unsupported("segment %d" % k)
            self.segments.append((a, b, plan))
        # Round 5: ONE plan over the whole chain as well -- its geometry (every rulebook, the geometry-only tail) depends on the
        # coordinates alone and can be built a frame ahead whatever the hooks do to the features; the convolutions then run
        # range by range (`BackbonePlan.run_convs_range`).  Layer ranges of the segments inside that plan:
        geo = [(self.stages[-1][0], geometry_module)] if geometry_module is not None else ()
        self.full = compile_stages(self.stages, geometry_stages=geo) if len(self.segments) > 1 else self.segments[0][2]
        self.ranges = []
        if self.full is not None:
            for k, (a, b, _) in enumerate(self.segments):
                lo = 0 if a == 0 else self.full.exports[self.stages[a - 1][0]] + 1
                hi = len(self.full.specs) if b == len(self.stages) else self.full.exports[self.stages[b - 1][0]] + 1
                self.ranges.append((lo, hi))

    def _run_ranges(self, x, hook, prepared):
        """The frame on the prepared geometry of the whole chain; None if it has to fall back to the per-segment path."""
        outs = {}
        feats = x.features
        for k, (a, b, _) in enumerate(self.segments):
            lo, hi = self.ranges[k]
            res = self.full.run_convs_range(prepared, feats, lo, hi)
            if res is None:
                return None if k == 0 else (outs, k)
            for i in range(a, b):
                name = self.stages[i][0]
                t = res[name]
                if hook is not None:
                    t = hook(i, name, t)
                outs[name] = t
            feats = outs[self.stages[b - 1][0]].features
        return outs, len(self.segments)

    def run(self, x, hook=None, prepared=None):
        """x: SparseConvTensor (network input).  hook(stage_index, name, tensor) -> tensor is called after every
        stage in order (identity when None) -- stages inside a segment see their output after the segment ran,
        which is fine for hooks that only READ; hooks that MODIFY features must sit at a cut.
        prepared: PreparedGeometry of the FIRST segment's plan (a frame head built ahead, dualfusion/prefetch.py).
        Returns {stage name: tensor}."""
        outs, k0 = {}, 0
        if prepared is not None and len(self.segments) > 1 and prepared.plan is self.full:
            got = self._run_ranges(x, hook, prepared)
            if got is not None:
                outs, k0 = got
                if k0 == len(self.segments):
                    return outs
                x = outs[self.stages[self.segments[k0][0] - 1][0]]      # (the arena ran out: the rest on the segment plans)
            prepared = None
        for k, (a, b, plan) in enumerate(self.segments):
            if k < k0:
                continue
            if k == 0 and prepared is not None and prepared.plan is plan:
                res = plan.run_convs(prepared, x.features)
            else:
                res = plan.run(x.features, x.indices, x.batch_size, x.spatial_shape)
            for i in range(a, b):
                name = self.stages[i][0]
                t = res[name]
                if hook is not None:
                    t = hook(i, name, t)
                outs[name] = t
            x = outs[self.stages[b - 1][0]]
        return outs


def build_runner(stages, cuts=(), geometry_module=None):
    try:
        return SegmentedRunner(stages, cuts, geometry_module)
    except Unsupported:
        return None
