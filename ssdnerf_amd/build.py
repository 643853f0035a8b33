"""Build libssdnerf_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

``-ffp-contract=off`` is part of the arithmetic contract: fused multiply-adds exist only
where the sources call ``__builtin_fmaf``, so integer outputs (sample counts, alive flags) are
reproducible bit for bit against the CPU oracle.

Every source goes  hipcc -S (device listing) -> ``asm_postpass.unpack_cross_half`` -> assembler -> lld -> checks on the LINKED code object -> offload bundle ->
host object that embeds it.  The post-pass has ONE job since round 6: every packed fp32 instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) whose op_sel /
op_sel_hi bits read ACROSS the halves of a VGPR source pair is replaced by the two plain instructions that compute the same two IEEE results.  On gfx950, with
two waves on a SIMD, such an instruction occasionally loses the product term of its low half in lanes 48-63: the cause of the run-to-run differences of the
fused render that rounds 2 - 5 chased as a transcendental -> use and a swap -> MFMA hazard (asm_postpass.py, the r06 block; DESIGN.md section 5.5;
profiles/r06/README.md).  The build FAILS if one such instruction is left in a linked code object.  The two padding rules of r03 / r05 (issue slots behind a
transcendental, between a lane swap and a matrix instruction) are still in asm_postpass.py for experiments and are OFF by default (``SSDNERF_TRANS_USE_WAIT_STATES=1``
is the toolchain's own distance, ``SSDNERF_SWAP_MFMA_WAIT_STATES=0``): with the crossed instructions gone, the compiler's own code is clean in every arrangement that
used to fail and over 10^5 renders (profiles/r06/g_*, e_*).  ``SSDNERF_KEEP_PACKED_CROSS_HALF=1`` keeps the compiler's instructions (positive-control builds only);
``SSDNERF_NO_POSTPASS=1`` builds with plain `hipcc -c`.  ``lib/postpass_report.json`` records what the pass did per source, the settings, and the toolchain.

The pass parses the device listing syntax of ROCm 7.2 and re-states `hipcc -c`'s internal device steps, so it is PINNED to the toolchains it was
validated on (``VALIDATED_HIP_VERSIONS``): another `hipcc --version` fails the build loudly (``SSDNERF_ALLOW_UNVALIDATED_TOOLCHAIN=1`` overrides
after the hazard runs of tools/ubench/gen_trans_use_hazard.py and tests/test_render_gpu.py::test_fused_render_is_reproducible_bit_for_bit have been
repeated on it).
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
SOURCES = ["raymarching_ops.hip", "shencoder.hip", "decode.hip", "render_fused.hip", "render_queue.hip", "shade_mfma.hip", "ddim.hip", "groupnorm.hip", "conv_igemm.hip", "attention.hip", "raygen.hip", "marching_cubes.hip"]
LLVM_BIN = os.environ.get("SSDNERF_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TRANS_USE_WAIT_STATES = int(os.environ.get("SSDNERF_TRANS_USE_WAIT_STATES", "1"))     # 1 = the toolchain's own distance: the r03 rule is off (r06)
SWAP_MFMA_WAIT_STATES = int(os.environ.get("SSDNERF_SWAP_MFMA_WAIT_STATES", "0"))    # asm_postpass.SWAP_MFMA_WAIT_STATES (0 = rule off: the default since r06)
UNPACK_CROSS_HALF = os.environ.get("SSDNERF_KEEP_PACKED_CROSS_HALF", "0") != "1"      # r06 (asm_postpass.unpack_cross_half); =1 keeps the compiler's instructions (positive-control builds)
VALU_MFMA_WAIT_STATES = int(os.environ.get("SSDNERF_VALU_MFMA_WAIT_STATES", "0"))    # asm_postpass.VALU_MFMA_WAIT_STATES (r06; 0 = rule off)
HEADERS = ["common.h", "sh_basis.h", "decode_core.h", "decode_bwd_math.h", "gn_bwd_math.h", os.path.join("..", "..", "include", "ssdnerf_hip.h")]
VALIDATED_HIP_VERSIONS = ("7.2.",)              # prefixes of `hipcc --version`'s "HIP version:" the post-pass + hazard analysis were validated on (r03 / r04)
FLAGS = os.environ.get("SSDNERF_EXTRA_FLAGS", "").split() + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-result"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _settings() -> dict:
    """what, besides the sources, decides the bytes of the library: compared with the shipped report by needs_build()"""
    return {"wait_states": TRANS_USE_WAIT_STATES if os.environ.get("SSDNERF_NO_POSTPASS", "0") != "1" else None,
            "extra_flags": os.environ.get("SSDNERF_EXTRA_FLAGS", ""), "swap_mfma_wait_states": SWAP_MFMA_WAIT_STATES,
            "valu_mfma_wait_states": VALU_MFMA_WAIT_STATES, "unpack_cross_half": UNPACK_CROSS_HALF}


def toolchain() -> dict:
    try:
        text = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True, check=True).stdout
    except Exception as e:                                           # noqa: BLE001
        return {"hip_version": None, "validated": False, "error": repr(e)}
    hip = next((l.split(":", 1)[1].strip() for l in text.splitlines() if l.startswith("HIP version")), None)
    clang = next((l.strip() for l in text.splitlines() if "clang version" in l), None)
    return {"hip_version": hip, "clang": clang, "validated": bool(hip) and hip.startswith(VALIDATED_HIP_VERSIONS)}


def render_build_id() -> dict:
    """what decides the render kernels of the library that is loaded: a hash over the render path's sources, and the build settings and the post-pass's effect on those sources from the
    shipped report.  tools/make_traffic_json.py stores it with a profiling session; bench.py quotes a session's HBM traffic only if it matches the library it times."""
    import hashlib
    h = hashlib.sha256()
    for f in ["shade_mfma.hip", "render_queue.hip", "common.h", "decode_core.h", "sh_basis.h"]:
        h.update(open(os.path.join(CSRC, f), "rb").read())
    try:
        with open(os.path.join(LIB_DIR, "postpass_report.json")) as f:
            rep = json.load(f)
        settings = rep.get("settings")
        # what the post-pass DID to the two render sources (not the text of asm_postpass.py: a comment there changes no instruction)
        effect = {k: {kk: rep["sources"][k].get(kk) for kk in ("packed_cross_half_split", "pairs_closer_than_required", "trans_instructions")}
                  for k in ("shade_mfma.hip", "render_queue.hip") if k in rep.get("sources", {})}
    except Exception:                                                # noqa: BLE001
        settings, effect = None, None
    return {"render_csrc_sha16": h.hexdigest()[:16], "build_settings": settings, "postpass_effect": effect}


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    try:
        with open(os.path.join(LIB_DIR, "postpass_report.json")) as f:
            if json.load(f).get("settings") != _settings():
                return True                                          # SSDNERF_TRANS_USE_WAIT_STATES / SSDNERF_NO_POSTPASS / SSDNERF_EXTRA_FLAGS changed
    except Exception:                                                # noqa: BLE001
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__), os.path.join(HERE, "asm_postpass.py")]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)


def assemble_and_link(dev_s: str, dev_o: str, dev_out: str, verbose: bool = False) -> None:
    """device listing -> device object -> linked code object (the two device steps `hipcc -c` runs internally after its -S stage)"""
    _run([os.path.join(LLVM_BIN, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", dev_s, "-o", dev_o], verbose)
    _run([os.path.join(LLVM_BIN, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-plugin-opt=-amdgpu-internalize-symbols",
          "-plugin-opt=mcpu=gfx950", "-o", dev_out, dev_o], verbose)


def _compile_with_postpass(src: str, obj: str, verbose: bool) -> dict:
    """device listing -> post-pass -> device object -> code object -> fat binary -> host object embedding it (the steps `hipcc -c` runs
    internally, with the listing edited in between); returns the post-pass statistics of this source"""
    from .asm_postpass import closest_trans_use, pad_trans_use, verify_code_object
    stem = obj[:-2]
    dev_s, dev_o, dev_out, fatbin = stem + ".dev.s", stem + ".dev.o", stem + ".dev.out", stem + ".hipfb"
    _run([_hipcc()] + FLAGS + ["-S", "--cuda-device-only", src, "-o", dev_s], verbose)
    with open(dev_s) as f:
        listing = f.read()
    from . import asm_postpass
    asm_postpass.SWAP_MFMA_WAIT_STATES = SWAP_MFMA_WAIT_STATES
    asm_postpass.VALU_MFMA_WAIT_STATES = VALU_MFMA_WAIT_STATES
    split_stats = {"packed_cross_half_split": None}
    if UNPACK_CROSS_HALF:                                            # r06: no packed fp32 instruction whose halves read across a VGPR source pair (asm_postpass.py, the r06 block)
        listing, split_stats = asm_postpass.unpack_cross_half(listing)
    patched, stats = pad_trans_use(listing, TRANS_USE_WAIT_STATES)
    stats.update(split_stats)
    closest = closest_trans_use(patched)
    if closest < TRANS_USE_WAIT_STATES:                              # (an explicit raise: `python -O` drops asserts)
        raise RuntimeError(f"{src}: a transcendental -> use pair is still {closest} slots apart after the post-pass")
    stats["closest_pair_after"] = closest if closest < (1 << 30) else None
    with open(dev_s, "w") as f:
        f.write(patched)
    assemble_and_link(dev_s, dev_o, dev_out, verbose)
    # the same rule on what the device will execute, by a scanner that shares nothing with the listing parser (raises on a violation)
    try:
        stats["code_object_check"] = verify_code_object(dev_out, TRANS_USE_WAIT_STATES, os.path.join(LLVM_BIN, "llvm-objdump"), SWAP_MFMA_WAIT_STATES)
    except RuntimeError as e:
        # the scanner also counts an OVERWRITE of a transcendental's destination as a mention (conservative); experimental side builds
        # (sensitivity probes with inline assembly) may ask for a warning instead -- never the library build
        if os.environ.get("SSDNERF_POSTPASS_VERIFY", "strict") != "warn" or LIB_DIR == os.path.join(HERE, "lib") and "variants" not in obj:
            raise
        print(f"warning (SSDNERF_POSTPASS_VERIFY=warn): {e}", file=sys.stderr)
        stats["code_object_check"] = {"warning": str(e)}
    if UNPACK_CROSS_HALF:                                            # ... and none in what the device will execute
        dis = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", dev_out], check=True, capture_output=True, text=True).stdout
        left = asm_postpass.scan_packed_cross_half(dis)
        if left:
            raise RuntimeError(f"{src}: packed fp32 instructions with a crossed VGPR source are left in the linked code object: {left}")
        stats["code_object_check"]["packed_cross_half"] = 0
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
    tc = toolchain()
    if postpass and not tc["validated"] and os.environ.get("SSDNERF_ALLOW_UNVALIDATED_TOOLCHAIN", "0") != "1":
        raise RuntimeError(f"ssdnerf_amd.build: the assembly post-pass was validated on HIP {VALIDATED_HIP_VERSIONS}*, this hipcc reports "
                           f"{tc.get('hip_version')!r} ({tc.get('clang')}).  The pass parses the device listing and re-states hipcc's device link line; "
                           "re-run the hazard checks on this toolchain (build.py docstring), then set SSDNERF_ALLOW_UNVALIDATED_TOOLCHAIN=1 or extend "
                           "VALIDATED_HIP_VERSIONS.")
    objs, report = [], {"wait_states": TRANS_USE_WAIT_STATES if postpass else None, "settings": _settings(), "toolchain": tc, "sources": {}}
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


def build_variant(name: str, source: str, extra_flags) -> str:
    """Side build for A/B runs (no GPU needed): recompile ONE source with extra flags -- through the same post-pass -- and link it with the
    in-tree objects into <repo>/.variants/<name>/libssdnerf_hip.so (git-ignored, shipped by gpurun).  Use on the GPU box with
    SSDNERF_HIP_LIB=.variants/<name>/libssdnerf_hip.so."""
    global FLAGS
    build()
    out_dir = os.path.join(os.path.dirname(HERE), ".variants", name)
    os.makedirs(out_dir, exist_ok=True)
    obj = os.path.join(out_dir, source.replace(".hip", ".o"))
    saved = FLAGS
    FLAGS = list(extra_flags) + FLAGS
    try:
        stats = _compile_with_postpass(os.path.join(CSRC, source), obj, False)
    finally:
        FLAGS = saved
    objs = [obj if s == source else os.path.join(LIB_DIR, s.replace(".hip", ".o")) for s in SOURCES]
    lib = os.path.join(out_dir, "libssdnerf_hip.so")
    _run([_hipcc(), "--offload-arch=gfx950", "-shared", "-o", lib] + objs, False)
    os.remove(obj)
    print(f"built {lib} ({source} {' '.join(extra_flags)}): {stats}")
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:                                      # python -m ssdnerf_amd.build --variant NAME --source shade_mfma.hip -- -DSM_X=1 ...
        i = sys.argv.index("--variant")
        extra = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
        build_variant(sys.argv[i + 1], sys.argv[sys.argv.index("--source") + 1], extra)
    else:
        print(build(force="--force" in sys.argv, verbose=True))
