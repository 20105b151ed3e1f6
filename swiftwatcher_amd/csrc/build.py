"""Build libswk.so (HIP, gfx950 only) in-tree: swiftwatcher_amd/libswk.so.

    python swiftwatcher_amd/csrc/build.py [--force]

hipcc cross-compiles for gfx950 without a GPU.  -ffp-contract=off: the bilateral filter's
float arithmetic must round exactly like a scalar C evaluation (no implicit FMA).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, "libswk.so")
SOURCES = ["swk_api.hip", "ialm.hip", "ialm_mfma.hip", "ialm_mstate.hip", "ialm_small.hip", "ialm_refine.hip", "ialm_gram8.hip", "filters.hip", "ccl.hip", "classify_input.hip", "cnn_aux.hip", "cnn_conv1.hip", "cnn_conv1x1.hip", "cnn_expand_bf16.hip", "cnn_conv3x3.hip", "cnn_wino3x3.hip", "cnn_poolsq.hip", "tracker.cpp", "roi_mask.cpp", "host_stage.cpp"]
HEADERS = ["swk_internal.h", "ialm_small_dev.h", os.path.join(ROOT, "include", "swk.h"), os.path.join(ROOT, "include", "swk_debug.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I", HERE]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [h if os.path.isabs(h) else os.path.join(HERE, h) for h in HEADERS]
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(HERE, src)
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        if force or _stale(obj, [sp] + hdrs + [os.path.abspath(__file__)]):
            cmd = [hipcc] + FLAGS + ["-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
