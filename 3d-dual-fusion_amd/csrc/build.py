#!/usr/bin/env python3
"""Build libdf3d_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["common.hip", "voxelize.hip", "rulebook.hip", "spconv.hip", "spconv_split.hip", "dense.hip", "msda.hip", "pointops.hip",
           "fusion.hip", "actr.hip", "ffn.hip", "imgproj.hip", "executor.hip", "nms.hip", "dettail.hip", "topk.hip", "pool.hip", "tfhead.hip", "xattn.hip",
           "spconv_bwd.hip"]
OUT = os.path.join(HERE, "libdf3d_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc"]


def build(force=False, verbose=False):
    srcs = [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    deps = srcs + [os.path.join(HERE, "common.h"), os.path.join(HERE, "..", "..", "include", "df3d_hip.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("DF3D_HIPCC_FLAGS", "").split()          # e.g. -DDF3D_OS_EXPERIMENTS for the tuning flags
    cmd = [hipcc] + FLAGS + extra + ["-o", OUT] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="-f" in sys.argv, verbose=True))
