#!/usr/bin/env python
"""tools/build_ws_variant.py NAME WAIT_STATES [-D flags...] -- side build of shade_mfma.hip with another SSDNERF_TRANS_USE_WAIT_STATES (the assembly
post-pass's distance between a transcendental and the first reader of its result) into .variants/NAME/; the other objects come from the in-tree build.
For the dose-response runs of the transcendental -> use hazard (tools/repro_check.py, profiles/r05/m_*)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name, ws, flags = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
if flags and flags[0].startswith("swap="):                       # NAME WAIT_STATES swap=N ...: the post-pass's second rule (asm_postpass.SWAP_MFMA_WAIT_STATES)
    os.environ["SSDNERF_SWAP_MFMA_WAIT_STATES"] = flags.pop(0).split("=")[1]
if flags and flags[0].startswith("valu="):                       # ... valu=N: the third rule (asm_postpass.VALU_MFMA_WAIT_STATES, r06)
    os.environ["SSDNERF_VALU_MFMA_WAIT_STATES"] = flags.pop(0).split("=")[1]
if flags and flags[0] == "keepcross":                            # ... keepcross: leave the compiler's packed fp32 instructions with crossed halves alone (r06 positive controls)
    flags.pop(0)
    os.environ["SSDNERF_KEEP_PACKED_CROSS_HALF"] = "1"
os.environ["SSDNERF_TRANS_USE_WAIT_STATES"] = str(ws)
if ws == 0:
    os.environ["SSDNERF_NO_POSTPASS"] = "1"
from ssdnerf_amd import build as b
b.FLAGS = flags + b.FLAGS
out_dir = os.path.join(os.path.dirname(b.HERE), ".variants", name)
os.makedirs(out_dir, exist_ok=True)
obj = os.path.join(out_dir, "shade_mfma.o")
if ws == 0:
    b._run([b._hipcc()] + b.FLAGS + ["-c", os.path.join(b.CSRC, "shade_mfma.hip"), "-o", obj], False)
    st = "compiler's own code"
else:
    st = b._compile_with_postpass(os.path.join(b.CSRC, "shade_mfma.hip"), obj, False)
objs = [obj if s == "shade_mfma.hip" else os.path.join(b.LIB_DIR, s.replace(".hip", ".o")) for s in b.SOURCES]
b._run([b._hipcc(), "--offload-arch=gfx950", "-shared", "-o", os.path.join(out_dir, "libssdnerf_hip.so")] + objs, False)
print(name, ws, flags, st)
