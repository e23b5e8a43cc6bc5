"""Build libupsnet_hip.so (hand-written HIP kernels for gfx950 / MI355X) in-tree.

    python -m upsnet_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU. The selection / sampling kernels must reproduce fp32
decisions bit-for-bit, hence -ffp-contract=off (no FMA contraction).
"""
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(CSRC, "libupsnet_hip.so")
STAMP = LIB + ".srchash"   # git-ignored, travels with the .so
SOURCES = ["capi.cpp", "fill.hip", "roi_align.hip", "nms.hip", "deform_conv.hip", "deform_fused.hip", "deform_fused_bf16.hip", "backward.hip", "conv.hip", "conv1x1.hip", "conv1x1_ksw.hip", "conv1x1_pair.hip", "conv_wino.hip", "conv_wino36.hip", "conv_bf16.hip", "bottleneck_bf16.hip", "conv3x3_wreg_bf16.hip", "conv1x1_wreg_bf16.hip", "stem_pool_bf16.hip", "stem_pool.hip", "proposal.hip", "detect.hip", "panoptic.hip", "fcn_head.hip", "preprocess.hip", "postprocess.hip"]
# -fno-slp-vectorize: the SLP vectorizer turns adjacent scalar fp32 adds / multiplies (the bilinear blend of the deformable kernels)
# into v_pk_add_f32 / v_pk_mul_f32, and a packed fp32 VALU instruction beside MFMAs costs far more than its issue slot
# (/opt/skills/guides/MI355X_MICROARCH.md); same IEEE operations either way, so results do not change.
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


_COMMENT = re.compile(rb"//[^\n]*|/\*.*?\*/", re.S)


def _source_hash():
    """sha1 over the compile flags and every source / header the library is built from, comments and whitespace runs removed (mtimes
    are not trusted: a shipped .so and a checked-out source can carry the same timestamp; and a comment edit must neither rebuild the
    library nor orphan the PMC profile that is stamped with this hash, tools/make_pmc_json.py)."""
    h = hashlib.sha1(" ".join(FLAGS + SOURCES).encode())
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h")))
    deps.append(os.path.join(INCLUDE, "upsnet_hip.h"))
    for d in deps:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(b" ".join(_COMMENT.sub(b" ", f.read()).split()))
    return h.hexdigest()


def _stale():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != _source_hash()


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
    with open(STAMP, "w") as f:
        f.write(_source_hash() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
