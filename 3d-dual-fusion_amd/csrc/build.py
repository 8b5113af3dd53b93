#!/usr/bin/env python3
"""Build libdf3d_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

Every .hip source is compiled to its own object (in parallel, only when it or a header changed) and the objects are
linked into the shared library; `python build.py -f` rebuilds everything, `-v` prints the commands."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["common.hip", "voxelize.hip", "rulebook.hip", "spconv.hip", "spconv_split.hip", "dense.hip", "msda.hip", "pointops.hip",
           "fusion.hip", "actr.hip", "ffn.hip", "imgproj.hip", "executor.hip", "nms.hip", "dettail.hip", "topk.hip", "pool.hip", "tfhead.hip", "xattn.hip",
           "spconv_bwd.hip", "loss.hip", "headconv.hip", "bnrows.hip", "tfloss.hip", "rowlinear.hip", "ltlayer.hip", "mvx.hip"]
OUT = os.path.join(HERE, "libdf3d_hip.so")
OBJ_DIR = os.path.join(HERE, "build")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc"]
HEADERS = [os.path.join(HERE, "common.h"), os.path.join(HERE, "..", "..", "include", "df3d_hip.h")]


def _newer(target, deps):
    return os.path.exists(target) and all(os.path.getmtime(target) >= os.path.getmtime(d) for d in deps)


def build(force=False, verbose=False):
    srcs = [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("DF3D_HIPCC_FLAGS", "").split()          # e.g. -DDF3D_OS_EXPERIMENTS for the tuning flags
    stamp = os.path.join(OBJ_DIR, "flags.txt")
    flags_txt = " ".join(CFLAGS + extra)
    if os.path.isdir(OBJ_DIR) and (not os.path.exists(stamp) or open(stamp).read() != flags_txt):
        force = True                                                 # built with other flags before
    if not force and _newer(OUT, srcs + HEADERS):
        return OUT
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs, jobs = [], []
    for src in srcs:
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not _newer(obj, [src] + HEADERS):
            jobs.append([hipcc] + CFLAGS + extra + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        return cmd, r.returncode, r.stdout + r.stderr

    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4) or 1) as pool:
        for cmd, rc, log in pool.map(run, jobs):
            if rc != 0:
                sys.stderr.write(log)
                raise subprocess.CalledProcessError(rc, cmd)
            if verbose and log.strip():
                print(log)
    with open(stamp, "w") as f:
        f.write(flags_txt)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", OUT] + objs
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    return OUT


if __name__ == "__main__":
    print(build(force="-f" in sys.argv, verbose="-v" in sys.argv))
