#!/usr/bin/env python
"""tools/asm_patch.py in.s out.s MODE [ARG] -- edit a hipcc device listing (-S --cuda-device-only) before it is assembled (tools/asm_patch_build.sh).
Keeps the compiler's schedule exactly and only adds idle issue slots at named places, which is what decides a wait-state question
(an `asm volatile("s_nop")` in the source is NOT pinned to the instruction it is meant to pad: the scheduler moves MFMAs across it).

MODE (distances are issue slots, an instruction = 1, `s_nop N` = N + 1, measured inside a straight-line run):
  none              copy (control build: must reproduce the unpatched behaviour)
  mfma_raw N        in front of every v_mfma whose A / B / C source has a non-MFMA writer closer than N slots: s_nop up to distance N
  mfma_raw_swap N   the same, only where that writer is v_permlane*_swap
  mfma_raw_valu N   the same, only where that writer is any other VALU instruction (v_mov / v_perm_b32 / ...)
  swap_raw N        in front of every v_permlane*_swap whose operand has a VALU writer closer than N slots
  all_mfma N        s_nop (N - 1) in front of EVERY v_mfma (blunt control)
Prints how many pads were inserted."""
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from isa_hazard_scan import parse, regs  # noqa: E402


def main():
    src, dst, mode = sys.argv[1], sys.argv[2], sys.argv[3]
    n_arg = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    out, run, pads = [], [], 0
    in_kernel = False
    for raw in open(src).read().split("\n"):
        t = raw.strip()
        if t.startswith(".amdhsa_kernel") or t.startswith(".end_amdhsa_kernel"):
            in_kernel = False
        if re.match(r"^_Z[\w$.]*:", t):
            in_kernel, run = True, []
        if not in_kernel or mode == "none":
            out.append(raw)
            continue
        if t.startswith(".LBB") or re.match(r"^[A-Za-z_.$][\w.$]*:", t):
            run = []
            out.append(raw)
            continue
        ins = parse(raw)
        if ins is None:
            out.append(raw)
            continue
        need = 0
        is_mfma = ins["op"].startswith("v_mfma")
        is_swap = ins["op"].startswith(("v_permlane32_swap", "v_permlane16_swap"))
        if is_mfma and mode == "all_mfma":
            need = n_arg
        elif (is_mfma and mode.startswith("mfma_raw")) or (is_swap and mode == "swap_raw"):
            want = set()
            for o in (ins["ops"][1:4] if is_mfma else ins["ops"][:2]):
                want |= regs(o)
            d = 0
            for prev in reversed(run):
                d += prev["slots"]
                if d >= n_arg:
                    break
                if prev["writes"] & want and not prev["op"].startswith("v_mfma") and prev["op"].startswith("v_"):
                    w_swap = prev["op"].startswith(("v_permlane32_swap", "v_permlane16_swap"))
                    if mode in ("mfma_raw", "swap_raw") or (mode == "mfma_raw_swap" and w_swap) or (mode == "mfma_raw_valu" and not w_swap):
                        need = max(need, n_arg - d)
                    # keep looking: a farther writer of another operand may still be inside the window
        if need > 0:
            pads += 1
            k = need
            while k > 0:
                step = min(k, 8)
                out.append(f"\ts_nop {step - 1}")
                run.append(dict(op="s_nop", ops=[str(step - 1)], writes=set(), reads=set(), slots=step, text=f"s_nop {step - 1}"))
                k -= step
        out.append(raw)
        run.append(ins)
        if ins["op"].startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
            run = []
    open(dst, "w").write("\n".join(out))
    print(f"asm_patch {mode} {n_arg}: {pads} pads inserted")


if __name__ == "__main__":
    main()
