"""The head of the NEXT frame while the current one is queued: `df3d_frame_head_*` (csrc/executor.hip) behind a small
Python handle.

Everything at the head of a frame depends on the raw inputs alone -- voxelisation (points), every rulebook of the sparse
backbone (voxel coordinates), the camera projection / query slots of the fusion adapter (coordinates + calibration) -- and
every one of those steps ends in a host round trip for a COUNT (voxels, active outputs of the strided layers, longest camera
list).  On the thread that queues the frames these waits cost ~1 ms of a ~3 ms step and leave the convolutions waiting for
their tables.  The reference hides the first of them by voxelising inside the DataLoader's worker processes while the GPU
step of the previous batch runs (CP/det3d/datasets/pipelines/preprocess.py `Voxelization`, torch DataLoader prefetch); here
a NATIVE worker thread per detector does all of them on the GPU (its own HIP stream): `submit` builds the job description
and returns at once, `take` hands the finished head to the queueing thread, whose stream then waits ON THE DEVICE for the
worker's last kernel.  (A Python helper thread was built first and measured: it works, bit-identically, but costs the
queueing thread the interpreter lock at every one of its ~150 C calls per frame -- 2.9 -> 3.5 ms per step.)"""
import ctypes
import time

import numpy as np
import torch

from . import _lib
from . import ops as _ops
from .executor import PreparedGeometry, _View

MAX_PROJ = 4


class _Project(ctypes.Structure):
    _fields_ = [("layer", ctypes.c_int), ("scale_xyz", ctypes.c_float * 3), ("want_winner", ctypes.c_int)]


class _Desc(ctypes.Structure):
    _fields_ = [("batch", ctypes.c_int), ("point_channels", ctypes.c_int), ("points", ctypes.c_void_p),
                ("num_points", ctypes.c_void_p), ("voxel_size", ctypes.c_float * 3), ("coors_range", ctypes.c_float * 6),
                ("max_points", ctypes.c_int), ("max_voxels", ctypes.c_int), ("break_at_cap", ctypes.c_int),
                ("layers", ctypes.c_void_p), ("nlayers", ctypes.c_int), ("shape", ctypes.c_int * 3), ("ncam", ctypes.c_int),
                ("lidar2cam", ctypes.c_void_p), ("intrinsic", ctypes.c_void_p), ("raw_hw", ctypes.c_void_p),
                ("depth_thres", ctypes.c_void_p), ("image_scale", ctypes.c_float), ("feat_scale", ctypes.c_void_p),
                ("aug_inv", ctypes.c_void_p), ("pc_min", ctypes.c_float * 3), ("nproj", ctypes.c_int),
                ("proj", _Project * MAX_PROJ), ("slots_proj", ctypes.c_int), ("inputs_ready", ctypes.c_void_p),
                ("img_ptrs", ctypes.c_void_p), ("img_count", ctypes.c_int), ("img_cin", ctypes.c_int),
                ("img_pixels", ctypes.c_int), ("img_packed", ctypes.c_void_p), ("feat_h", ctypes.c_int),
                ("feat_w", ctypes.c_int), ("want_pixrow", ctypes.c_int)]


class _Out(ctypes.Structure):
    _fields_ = [("features", ctypes.c_void_p), ("coors", ctypes.c_void_p), ("n", ctypes.c_int), ("max_ne", ctypes.c_int),
                ("grid_xy", ctypes.c_void_p * MAX_PROJ), ("mask", ctypes.c_void_p * MAX_PROJ),
                ("point_inv", ctypes.c_void_p * MAX_PROJ), ("proj_n", ctypes.c_int * MAX_PROJ), ("pos", ctypes.c_void_p),
                ("counts", ctypes.c_void_p), ("img_split", ctypes.c_void_p), ("img_gate", ctypes.c_void_p),
                ("img_done", ctypes.c_void_p), ("winner", ctypes.c_void_p * MAX_PROJ), ("pixrow", ctypes.c_void_p),
                ("pixrow_total", ctypes.c_int)]


class Prepared(object):
    """One frame's finished head: voxel features / coordinates, the backbone's PreparedGeometry, the fusion adapter's
    prepared dict (fusion.VoxelWithPointProjection.use_prepared)."""

    def __init__(self):
        self.feats = self.coors = self.geometry = self.fusion = None

    def hand_over(self):
        """The CURRENT stream waits (on the device) for the worker's last kernel.  Every tensor of the head lives in the
        frame slot's persistent arena, so the caching allocator has nothing to learn."""
        if self.geometry is not None:
            self.geometry.wait()
            if self.fusion is not None and "both" in self.fusion:
                _lib.check(_lib.load().df3d_frame_head_image_wait(self.geometry.handle, _ops._stream()),
                           "df3d_frame_head_image_wait")


class _Ticket(object):
    def __init__(self):
        self.handle = self.slot = self.plan = self.keep = self.sig = self.job = None
        self.cam = None


class FrameHead(object):
    def __init__(self, device):
        self.device = torch.device(device)
        self.lib = _lib.load()
        self.worker = ctypes.c_void_p(self.lib.df3d_head_worker_create(int(self.device.index or 0)))
        if not self.worker:
            raise _lib.Df3dError("df3d_head_worker_create failed")
        self._pending = {}
        # host seconds; take_wait_s = waiting for the worker, slot_wait_s = waiting for a frame slot (GPU a whole ring behind)
        self.stats = {"submits": 0, "submit_s": 0.0, "takes": 0, "take_wait_s": 0.0, "take_s": 0.0, "slot_wait_s": 0.0}

    # ------------------------------------------------------------------ submit
    def submit(self, key, plan, points_list, vox, shape, cam=None, owner=None):
        """owner: objects whose id() is part of `key` (the batch_dict): referenced until the head is taken, so that the id
        cannot be reused by another object meanwhile."""
        t0 = time.perf_counter()
        try:
            t = self._submit_job(key, plan, points_list, vox, shape, cam)
            t.owner = owner
            return t
        finally:
            self.stats["submits"] += 1
            self.stats["submit_s"] += time.perf_counter() - t0

    def _submit_job(self, key, plan, points_list, vox, shape, cam=None):
        """vox: dict(voxel_size, coors_range, max_points, max_voxels, break_at_cap); shape: sparse shape (z, y, x) of the
        backbone input; cam: None or dict(inp=<fusion._gather_inputs result>, pc_min, image_scale, levels=[(level index, layer
        index, scale_xyz)], slots_level, ready=<torch event or None>)."""
        t = _Ticket()
        t.plan, t.cam = plan, cam
        with plan._lock:
            sig = plan._signature()
            if plan._table is None or sig != plan._sig:
                plan._build_table()
                plan._sig = sig
            t.keep, t.sig = (plan._table, plan._keep), sig
            b0 = plan._frames.blocked_s
            t.slot = plan._frames.acquire()
            self.stats["slot_wait_s"] += plan._frames.blocked_s - b0
        B = len(points_list)
        d = _Desc()
        d.batch, d.point_channels = B, int(points_list[0].shape[1])
        ptrs = (ctypes.c_void_p * B)(*[p.data_ptr() for p in points_list])
        nums = (ctypes.c_int * B)(*[int(p.shape[0]) for p in points_list])
        d.points, d.num_points = ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(nums, ctypes.c_void_p)
        for i in range(3):
            d.voxel_size[i] = float(vox["voxel_size"][i])
            d.shape[i] = int(shape[i])
        for i in range(6):
            d.coors_range[i] = float(vox["coors_range"][i])
        d.max_points, d.max_voxels, d.break_at_cap = int(vox["max_points"]), int(vox["max_voxels"]), int(bool(vox["break_at_cap"]))
        d.layers, d.nlayers = ctypes.cast(plan._table, ctypes.c_void_p), len(plan.specs)
        d.slots_proj = -1
        keep = [points_list]
        if cam is not None:
            inp = cam["inp"]
            d.ncam = int(inp["ncam"])
            d.lidar2cam, d.intrinsic = inp["l2c"].data_ptr(), inp["intr"].data_ptr()
            d.raw_hw, d.depth_thres, d.feat_scale = inp["raw_hw"].data_ptr(), inp["thres"].data_ptr(), inp["feat_scale"].data_ptr()
            d.image_scale = float(np.float32(cam["image_scale"]))
            if inp.get("aug_inv") is not None:
                d.aug_inv = inp["aug_inv"].data_ptr()
            for i in range(3):
                d.pc_min[i] = float(np.float32(cam["pc_min"][i]))
            d.nproj = len(cam["levels"])
            for j, (lvl, layer, scale) in enumerate(cam["levels"]):
                d.proj[j].layer = int(layer)
                for i in range(3):
                    d.proj[j].scale_xyz[i] = float(scale[i])
                d.proj[j].want_winner = int(lvl in cam.get("winner_levels", ()))
                if lvl == cam["slots_level"]:
                    d.slots_proj = j
            d.feat_h, d.feat_w = int(inp["h"]), int(inp["w"])
            d.want_pixrow = int(bool(cam.get("want_pixrow")))
            if cam.get("ready") is not None:
                d.inputs_ready = cam["ready"].cuda_event
                keep.append(cam["ready"])
            img = cam.get("image_projection")
            if img is not None:                 # dict(packed=<df3d_imgproj_pack tensor>, cin=, pixels=): project the maps as well
                d.img_ptrs, d.img_count = inp["img_ptrs"].data_ptr(), len(inp["imgs"])
                d.img_cin, d.img_pixels, d.img_packed = int(img["cin"]), int(img["pixels"]), img["packed"].data_ptr()
                keep.append(img["packed"])
            keep.append(inp)
        t.job = (d, ptrs, nums, keep)
        self._submit(t)
        self._pending[key] = t
        return t

    def _submit(self, t):
        arena = t.slot.arena("geo", t.plan._geo_bytes, self.device)
        t.arena = arena
        handle = ctypes.c_void_p(0)
        rc = self.lib.df3d_frame_head_submit(self.worker, ctypes.byref(t.job[0]), _ops._ptr(arena), arena.numel(),
                                             ctypes.byref(handle))
        _lib.check(rc, "df3d_frame_head_submit")
        t.handle = handle

    # ------------------------------------------------------------------ take
    def take(self, key):
        t = self._pending.pop(key, None)
        if t is None:
            return None
        plan = t.plan
        nl = len(plan.specs)
        t_take = time.perf_counter()
        while True:
            views, out = (_View * nl)(), _Out()
            used, handle = ctypes.c_size_t(0), ctypes.c_void_p(0)
            t0 = time.perf_counter()
            rc = self.lib.df3d_frame_head_wait(t.handle, views, ctypes.byref(out), ctypes.byref(used), ctypes.byref(handle))
            self.stats["take_wait_s"] += time.perf_counter() - t0
            t.handle = None
            if rc == _lib.DF3D_ENOMEM and plan._geo_bytes < (64 << 30):
                plan._geo_bytes = max(2 * plan._geo_bytes, int(used.value) + (64 << 20))
                torch.cuda.synchronize(self.device)            # the failed attempt's kernels still write the small arena
                self._submit(t)
                continue
            _lib.check(rc, "df3d_frame_head_wait")
            break
        arena = t.arena
        base = arena.data_ptr()

        def view(ptr, nbytes, dtype, shape):
            off = ptr - base
            return arena[off:off + nbytes].view(dtype).view(shape)

        prep = Prepared()
        n, C = int(out.n), int(t.job[0].point_channels)
        prep.feats = view(out.features, n * C * 4, torch.float32, (n, C))
        prep.coors = view(out.coors, n * 16, torch.int32, (n, 4))
        if n == 0:
            self.lib.df3d_backbone_release(handle)
            return prep
        B = int(t.job[0].batch)
        geo = PreparedGeometry(plan, handle, views, t.slot, prep.coors, B, t.keep, t.sig)
        geo.spatial_shape = [int(v) for v in t.job[0].shape]
        geo.stages = plan._export(views, (arena,), prep.coors, B, features=False)
        prep.geometry = geo
        cam = t.cam
        if cam is not None:
            ncam = int(t.job[0].ncam)
            proj, winners = {}, {}
            hw = int(t.job[0].feat_h) * int(t.job[0].feat_w)
            for j, (lvl, layer, scale) in enumerate(cam["levels"]):
                m = int(out.proj_n[j])
                if out.winner[j]:
                    winners[lvl] = view(out.winner[j], B * ncam * hw * 4, torch.int32,
                                        (B * ncam, int(t.job[0].feat_h), int(t.job[0].feat_w)))
                proj[lvl] = (view(out.grid_xy[j], ncam * m * 8, torch.int32, (ncam, m, 2)),
                             view(out.mask[j], ncam * m, torch.uint8, (ncam, m)),
                             view(out.point_inv[j], m * 12, torch.float32, (m, 3)))
            early = None
            if int(t.job[0].slots_proj) >= 0:
                m = int(out.proj_n[int(t.job[0].slots_proj)])
                early = (view(out.pos, ncam * m * 4, torch.int32, (ncam, m)), int(out.max_ne),
                         view(out.counts, B * ncam * 4, torch.int32, (B * ncam,)))
            prep.fusion = dict(inp=cam["inp"], proj=proj, early=early, winner=winners)
            if out.pixrow:
                prep.fusion["pixrow"] = (view(out.pixrow, B * ncam * hw * 4, torch.int32, (B * ncam, hw)), int(out.pixrow_total))
            if out.img_split:
                ni, px = int(t.job[0].img_count), int(t.job[0].img_pixels)
                prep.fusion["both"] = (view(out.img_split, ni * px * 512, torch.uint8, (ni, px, 512)),
                                       view(out.img_gate, ni * px * 4, torch.float32, (ni, px)))
        self.stats["takes"] += 1
        self.stats["take_s"] += time.perf_counter() - t_take
        return prep

    def drop_all(self):
        for key in list(self._pending):
            prep = self.take(key)
            if prep is not None and prep.geometry is not None:
                prep.geometry.release()

    def close(self):
        if self.worker:
            self.drop_all()
            self.lib.df3d_head_worker_destroy(self.worker)
            self.worker = None
            # the last job's kernels may still be writing the frame slots' arenas on the worker's stream (a stream the caching
            # allocator knows nothing about): nothing may recycle that memory before they are done
            torch.cuda.synchronize(self.device)

    def __del__(self):
        try:
            self.close()
        except Exception:             # noqa: BLE001  (interpreter shutdown)
            pass
