"""Build libssdnerf_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

``-ffp-contract=off`` is part of the arithmetic contract: fused multiply-adds exist only
where the sources call ``__builtin_fmaf``, so integer outputs (sample counts, alive flags) are
reproducible bit for bit against the CPU oracle.

Every source goes  hipcc -S (device listing) -> ``asm_postpass.pad_trans_use`` (two wait states behind every transcendental instruction:
the toolchain pads that hazard to one, which is not always enough on gfx950 with two waves on a SIMD -- see asm_postpass.py and
profiles/r03/hazard.txt) -> assembler -> lld -> offload bundle -> host object that embeds it.  ``SSDNERF_NO_POSTPASS=1`` builds the
compiler's own code (A/B runs only).  ``lib/postpass_report.json`` records what the pass did per source.
"""
from __future__ import annotations

import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.environ.get("SSDNERF_LIB_DIR") or os.path.join(HERE, "lib")     # SSDNERF_LIB_DIR + SSDNERF_EXTRA_FLAGS: side builds for A/B runs
LIB_PATH = os.path.join(LIB_DIR, "libssdnerf_hip.so")
SOURCES = ["raymarching_ops.hip", "shencoder.hip", "decode.hip", "render_fused.hip", "render_queue.hip", "shade_mfma.hip", "ddim.hip", "groupnorm.hip", "conv_igemm.hip", "attention.hip", "raygen.hip"]
LLVM_BIN = os.environ.get("SSDNERF_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TRANS_USE_WAIT_STATES = int(os.environ.get("SSDNERF_TRANS_USE_WAIT_STATES", "4"))
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
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__), os.path.join(HERE, "asm_postpass.py")]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)


def _compile_with_postpass(src: str, obj: str, verbose: bool) -> dict:
    """device listing -> post-pass -> device object -> code object -> fat binary -> host object embedding it (the steps `hipcc -c` runs
    internally, with the listing edited in between); returns the post-pass statistics of this source"""
    from .asm_postpass import closest_trans_use, pad_trans_use
    stem = obj[:-2]
    dev_s, dev_o, dev_out, fatbin = stem + ".dev.s", stem + ".dev.o", stem + ".dev.out", stem + ".hipfb"
    _run([_hipcc()] + FLAGS + ["-S", "--cuda-device-only", src, "-o", dev_s], verbose)
    with open(dev_s) as f:
        listing = f.read()
    patched, stats = pad_trans_use(listing, TRANS_USE_WAIT_STATES)
    closest = closest_trans_use(patched)
    assert closest >= TRANS_USE_WAIT_STATES, f"{src}: a transcendental -> use pair is still {closest} slots apart after the post-pass"
    stats["closest_pair_after"] = closest if closest < (1 << 30) else None
    with open(dev_s, "w") as f:
        f.write(patched)
    _run([os.path.join(LLVM_BIN, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", dev_s, "-o", dev_o], verbose)
    _run([os.path.join(LLVM_BIN, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-plugin-opt=-amdgpu-internalize-symbols",
          "-plugin-opt=mcpu=gfx950", "-o", dev_out, dev_o], verbose)
    _run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
          "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", f"-input={dev_out}", f"-output={fatbin}"], verbose)
    _run([_hipcc()] + FLAGS + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fatbin, "-c", src, "-o", obj], verbose)
    for tmp in (dev_s, dev_o, dev_out, fatbin):
        os.remove(tmp)
    return stats


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    postpass = os.environ.get("SSDNERF_NO_POSTPASS", "0") != "1"
    objs, report = [], {"wait_states": TRANS_USE_WAIT_STATES if postpass else None, "sources": {}}
    for src in SOURCES:
        obj = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        if postpass:
            report["sources"][src] = _compile_with_postpass(os.path.join(CSRC, src), obj, verbose)
        else:
            _run([_hipcc()] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj], verbose)
        objs.append(obj)
    _run([_hipcc(), "--offload-arch=gfx950", "-shared", "-o", LIB_PATH] + objs, verbose)
    with open(os.path.join(LIB_DIR, "postpass_report.json"), "w") as f:
        json.dump(report, f, indent=1)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
