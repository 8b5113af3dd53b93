#!/usr/bin/env python3
"""Build the REAL reference CPU code into oracle/_ref/ (test infrastructure only).

This is the recipe SURVEY.md §8(c) describes.  It compiles the reference's own
sources from where they lie under /root/reference; nothing of them is copied
into the repository: torch's hipify step needs a writable tree, so the sources
are staged in a throw-away directory under $TMPDIR, and ONLY the resulting
shared objects are placed in oracle/_ref/ (git-ignored, but not gpurun-ignored,
so the binaries travel to the GPU box like our own .so files).

Built here:
  oracle/_ref/sparse_conv_ext.so  <- TF/mmdet3d/ops/spconv/src/{all,indice,reordering,maxpool}.cc
                                     + {reordering,maxpool}_cuda.cu  (spconv v1, CPU + GPU gather paths)
      `src/indice_cuda.cu` does not compile under hipcc (tensorview.h keys
      TV_HOST_DEVICE on __CUDACC__).  We do NOT write a stand-in for it: the
      library is linked with those three GPU functors left undefined and is
      loaded with RTLD_LAZY (see oracle/ref.py); the CPU code path never
      touches them.
  oracle/_ref/voxel_layer.so      <- TF/mmdet3d/ops/voxel/src/{voxelization.cpp,
                                     voxelization_cpu.cpp, scatter_points_cpu.cpp} (CPU only)

  oracle/_ref/iou3d_nms_cuda.so  <- CP/det3d/ops/iou3d_nms/src/{iou3d_cpu.cpp, iou3d_nms.cpp, iou3d_nms_api.cpp,
                                     iou3d_nms_kernel.cu} (rotated BEV IoU / NMS: the CPU path `boxes_iou_bev_cpu`
                                     and, through torch's own hipify step, the reference's GPU kernels
                                     `boxes_overlap_bev_gpu / boxes_iou_bev_gpu / nms_gpu / nms_normal_gpu`, which
                                     the -m gpu tests run beside ours on the MI355X)

  oracle/_ref/iou3d_cuda.so      <- TF/mmdet3d/ops/iou3d/src/{iou3d.cpp, iou3d_kernel.cu}: the TransFusion tree's
                                     `boxes_overlap_bev_gpu / boxes_iou_bev_gpu / nms_gpu / nms_normal_gpu` (GPU only; the
                                     overlap kernel the Hungarian matcher's 3-D IoU runs on).  tests/test_gpu_tfloss.py runs it
                                     beside ours and beside the oracle's restatement on the MI355X.

Nothing outside tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may load these libraries.
"""
import os
import shutil
import sys
import tempfile

REF = "/root/reference/TransFusion/mmdet3d/ops"
REF_CP = "/root/reference/CenterPoint/det3d/ops"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")


def have_reference():
    return os.path.isdir(os.path.join(REF, "spconv", "src"))


def build(verbose=False):
    if not have_reference():
        print("[oracle/_ref] /root/reference absent: keeping prebuilt files", file=sys.stderr)
        return False
    os.makedirs(OUT, exist_ok=True)
    want = [os.path.join(OUT, "sparse_conv_ext.so"), os.path.join(OUT, "voxel_layer.so"),
            os.path.join(OUT, "iou3d_nms_cuda.so"), os.path.join(OUT, "iou3d_cuda.so")]
    if all(os.path.exists(w) for w in want) and not os.environ.get("DF3D_REBUILD_REF"):
        return True
    os.environ.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
    os.environ.setdefault("MAX_JOBS", "8")
    from torch.utils.cpp_extension import load

    stage = tempfile.mkdtemp(prefix="df3d_ref_")
    try:
        # ---- spconv v1 -------------------------------------------------
        sp = os.path.join(stage, "spconv")
        os.makedirs(sp)
        for d in ("include", "src"):
            shutil.copytree(os.path.join(REF, "spconv", d), os.path.join(sp, d))
        os.system("chmod -R u+w %s" % stage)
        bdir = os.path.join(sp, "out")
        os.makedirs(bdir)
        srcs = [os.path.join(sp, "src", f) for f in
                ("all.cc", "reordering.cc", "reordering_cuda.cu", "indice.cc", "maxpool.cc", "maxpool_cuda.cu")]
        try:
            load(name="sparse_conv_ext", sources=srcs, extra_include_paths=[os.path.join(sp, "include")],
                 extra_cflags=["-w", "-std=c++17"], extra_cuda_cflags=["-w", "-std=c++17"],
                 build_directory=bdir, with_cuda=True, verbose=verbose)
        except ImportError as e:  # expected: undefined GPU rulebook functors at RTLD_NOW
            if "undefined symbol" not in str(e):
                raise
        shutil.copy(os.path.join(bdir, "sparse_conv_ext.so"), want[0])

        # ---- voxel_layer (CPU) ----------------------------------------
        vx = os.path.join(stage, "voxel")
        os.makedirs(vx)
        for f in ("voxelization.cpp", "voxelization.h", "voxelization_cpu.cpp", "scatter_points_cpu.cpp"):
            shutil.copy(os.path.join(REF, "voxel", "src", f), vx)
        bdir = os.path.join(vx, "out")
        os.makedirs(bdir)
        load(name="voxel_layer", sources=[os.path.join(vx, f) for f in
                                         ("voxelization.cpp", "voxelization_cpu.cpp", "scatter_points_cpu.cpp")],
             extra_cflags=["-w"], build_directory=bdir, verbose=verbose)
        shutil.copy(os.path.join(bdir, "voxel_layer.so"), want[1])

        # ---- iou3d_nms (CPU IoU + hipified GPU kernels) -----------------
        io = os.path.join(stage, "iou3d_nms")
        shutil.copytree(os.path.join(REF_CP, "iou3d_nms", "src"), io)
        os.system("chmod -R u+w %s" % io)
        bdir = os.path.join(io, "out")
        os.makedirs(bdir)
        load(name="iou3d_nms_cuda", sources=[os.path.join(io, f) for f in
                                            ("iou3d_cpu.cpp", "iou3d_nms.cpp", "iou3d_nms_api.cpp", "iou3d_nms_kernel.cu")],
             extra_include_paths=[io], extra_cflags=["-w"], extra_cuda_cflags=["-w"], build_directory=bdir,
             with_cuda=True, verbose=verbose)
        shutil.copy(os.path.join(bdir, "iou3d_nms_cuda.so"), want[2])

        # ---- TransFusion tree's iou3d (hipified GPU kernels only) --------
        it = os.path.join(stage, "iou3d_tf")
        shutil.copytree(os.path.join(REF, "iou3d", "src"), it)
        os.system("chmod -R u+w %s" % it)
        bdir = os.path.join(it, "out")
        os.makedirs(bdir)
        load(name="iou3d_cuda", sources=[os.path.join(it, f) for f in ("iou3d.cpp", "iou3d_kernel.cu")],
             extra_include_paths=[it], extra_cflags=["-w"], extra_cuda_cflags=["-w"], build_directory=bdir,
             with_cuda=True, verbose=verbose)
        shutil.copy(os.path.join(bdir, "iou3d_cuda.so"), want[3])
    finally:
        shutil.rmtree(stage, ignore_errors=True)
    return True


if __name__ == "__main__":
    ok = build(verbose="-v" in sys.argv)
    print("oracle/_ref:", sorted(os.listdir(OUT)) if os.path.isdir(OUT) else None, "built" if ok else "skipped")
