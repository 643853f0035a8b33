"""Build libssdnerf_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

``-ffp-contract=off`` is part of the arithmetic contract: fused multiply-adds exist only
where the sources call ``__builtin_fmaf``, so integer outputs (sample counts, alive flags) are
reproducible bit for bit against the CPU oracle.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.environ.get("SSDNERF_LIB_DIR") or os.path.join(HERE, "lib")     # SSDNERF_LIB_DIR + SSDNERF_EXTRA_FLAGS: side builds for A/B runs
LIB_PATH = os.path.join(LIB_DIR, "libssdnerf_hip.so")
SOURCES = ["raymarching_ops.hip", "shencoder.hip", "decode.hip", "render_fused.hip", "render_queue.hip", "shade_mfma.hip", "ddim.hip", "groupnorm.hip", "conv_igemm.hip", "attention.hip", "raygen.hip"]
HEADERS = ["common.h", "sh_basis.h", "decode_core.h", "decode_bwd_math.h", "gn_bwd_math.h", os.path.join("..", "..", "include", "ssdnerf_hip.h")]
FLAGS = os.environ.get("SSDNERF_EXTRA_FLAGS", "").split() + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-result"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    objs = []
    for src in SOURCES:
        obj = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        cmd = [_hipcc()] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-o", LIB_PATH] + objs
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
