#!/usr/bin/env python
"""tools/isa_census.py -- static instruction census of a kernel in a hipcc -S listing, per section (line ranges of the listing, or the
whole kernel): VALU / packed / transcendental / MFMA / LDS / VMEM / s_nop counts and the lane-spill traffic (v_readlane / v_writelane).
Used with the PMC totals (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU) to budget cycles per loop iteration.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Issdnerf_amd/csrc -Iinclude -S --cuda-device-only \\
          ssdnerf_amd/csrc/shade_mfma.hip -o /tmp/sm.s
    python tools/isa_census.py /tmp/sm.s k_shade_mfmaIfLi4E [start:end ...]"""
import collections
import sys


def main():
    path, kern = sys.argv[1], sys.argv[2]
    s = open(path).read()
    i = s.index(kern)
    i = s.index("\n", s.index(":", i))
    body = s[i:s.index(".end_amdhsa_kernel", i)]
    lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
    ranges = [tuple(int(v) for v in a.split(":")) for a in sys.argv[3:]] or [(0, len(lines))]
    for k, l in enumerate(lines):
        if l.startswith(".LBB") or l.startswith(("s_cbranch", "s_branch")):
            print(f"{k:6d}  {l}")
    for a, b in ranges:
        c = collections.Counter(x.split()[0] for x in lines[a:b] if not x.startswith("."))
        trans = sum(n for k, n in c.items() if k.startswith(("v_exp", "v_rcp", "v_rsq", "v_sqrt", "v_log", "v_sin", "v_cos")))
        mfma = sum(n for k, n in c.items() if k.startswith("v_mfma"))
        pk = sum(n for k, n in c.items() if k.startswith("v_pk_"))
        valu = sum(n for k, n in c.items() if k.startswith("v_")) - mfma
        print(f"[{a}:{b}] instructions={b - a} VALU={valu} (transcendental {trans}, packed {pk}) MFMA={mfma} "
              f"LDS={sum(n for k, n in c.items() if k.startswith('ds_'))} VMEM={sum(n for k, n in c.items() if k.startswith(('global_', 'buffer_', 'flat_')))} "
              f"s_nop={c['s_nop']} readlane={c['v_readlane_b32']} writelane={c['v_writelane_b32']} v_mov={c['v_mov_b32_e32']}")
        print("   top:", ", ".join(f"{k} {n}" for k, n in c.most_common(12)))


if __name__ == "__main__":
    main()
