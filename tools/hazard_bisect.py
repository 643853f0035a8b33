#!/usr/bin/env python
"""tools/hazard_bisect.py NAME LO HI [INSIDE] [-D flags...] -- side build of shade_mfma.hip for the hazard bisect of r05: the
assembly post-pass pads every transcendental -> use pair to the build's 4 wait states, EXCEPT the sites LO .. HI-1 of `k_shade_mfma<float, 2, 6>` (readers
the pass would pad at INSIDE = 7 wait states, in listing order), which get INSIDE.  With the head grouping that fails massively at 4 and not at all at 7
(-DSM_QB={0,2,4,7,10,13,16}), halving the range of sites that still cures it walks towards the instruction pair that matters.  Prints the number of sites."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rest = sys.argv[4:]
inside = int(rest.pop(0)) if rest and not rest[0].startswith("-") else 7
from ssdnerf_amd import asm_postpass, build as b
asm_postpass.SITE_FILTER = ("k_shade_mfmaIfLi2ELi6E", lo, hi, inside)
b.FLAGS = rest + b.FLAGS
out_dir = os.path.join(os.path.dirname(b.HERE), ".variants", name)
os.makedirs(out_dir, exist_ok=True)
obj = os.path.join(out_dir, "shade_mfma.o")
os.environ["SSDNERF_POSTPASS_VERIFY"] = "warn"
st = b._compile_with_postpass(os.path.join(b.CSRC, "shade_mfma.hip"), obj, False)
objs = [obj if s == "shade_mfma.hip" else os.path.join(b.LIB_DIR, s.replace(".hip", ".o")) for s in b.SOURCES]
b._run([b._hipcc(), "--offload-arch=gfx950", "-shared", "-o", os.path.join(out_dir, "libssdnerf_hip.so")] + objs, False)
print(name, lo, hi, inside, {k: st[k] for k in ("pairs_closer_than_required", "lengthened_in_place", "inserted")})
