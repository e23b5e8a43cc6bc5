"""Build libupsnet_hip.so (hand-written HIP kernels for gfx950 / MI355X) in-tree.

    python -m upsnet_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU. The selection / sampling kernels must reproduce fp32
decisions bit-for-bit, hence -ffp-contract=off (no FMA contraction).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(CSRC, "libupsnet_hip.so")
SOURCES = ["capi.cpp", "roi_align.hip", "nms.hip", "deform_conv.hip", "backward.hip", "conv.hip", "conv_wino.hip", "conv_bf16.hip", "proposal.hip", "detect.hip", "panoptic.hip", "fcn_head.hip", "preprocess.hip", "postprocess.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h"))]
    deps.append(os.path.join(INCLUDE, "upsnet_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "hipcc")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc] + FLAGS + ["-I" + INCLUDE, "-I" + CSRC, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % src)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
