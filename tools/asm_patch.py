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
  trans_use N [LO-HI]   at least N wait states between a transcendental VALU instruction (v_exp / v_rcp / v_rsq / v_sqrt / v_log / v_sin / v_cos) and
                    the first VALU instruction that READS its result (the toolchain's VALUTransUseHazard pads this to 1); optionally only for
                    consumers whose instruction index lies in [LO, HI)
  trans_use_inplace N [LO-HI]   the same, but ONLY by lengthening the compiler's own s_nop (byte-for-byte the same code layout)
  shift_at K N      N x `s_nop 0` in front of the K-th instruction of every kernel (K = 0: the first): moves all code behind it by 4 N bytes
                    (placement experiments: does ANY shift change the behaviour, and from where on?)
  mfma_raw_only N SITES   as mfma_raw, but only at the qualifying sites whose index (per kernel, in program order, from 0) is in the comma list
                    SITES -- bisects WHICH (writer -> v_mfma) pair matters; the sites are printed
Prints how many pads were inserted."""
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from isa_hazard_scan import parse, regs  # noqa: E402


def main():
    src, dst, mode = sys.argv[1], sys.argv[2], sys.argv[3]
    n_arg = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    only = None
    if mode == "mfma_raw_only":
        only = set(int(v) for v in sys.argv[5].split(",")) if len(sys.argv) > 5 and sys.argv[5] != "none" else set()
        mode = "mfma_raw"
    shift_k = n_arg if mode == "shift_at" else -1
    shift_n = int(sys.argv[5]) if mode == "shift_at" else 0
    icount = 0
    site = -1
    kernel_name = ""
    out, run, pads = [], [], 0
    in_kernel = False
    for raw in open(src).read().split("\n"):
        t = raw.strip()
        if t.startswith(".amdhsa_kernel") or t.startswith(".end_amdhsa_kernel"):
            in_kernel = False
        if re.match(r"^_Z[\w$.]*:", t):
            in_kernel, run, site, kernel_name, icount = True, [], -1, t.split(":")[0], 0
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
        if mode in ("trans_use", "trans_use_inplace"):
            lo, hi = (int(v) for v in sys.argv[5].split("-")) if len(sys.argv) > 5 and "-" in sys.argv[5] else (0, 1 << 30)
            need_t = 0
            if ins["op"].startswith("v_") and lo <= icount < hi:
                d = 0
                for prev in reversed(run):
                    if d >= n_arg:
                        break
                    if prev["op"].startswith(("v_exp_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_log_", "v_sin_", "v_cos_")) and prev["writes"] & ins["reads"]:
                        need_t = max(need_t, n_arg - d)
                    if prev["writes"] & ins["reads"] and not prev["op"].startswith(("v_exp_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_log_", "v_sin_", "v_cos_")):
                        pass
                    d += prev["slots"]
            if need_t > 0:
                m = re.match(r"^\s*s_nop (\d+)\s*$", out[-1]) if out else None
                if m and run and run[-1]["op"] == "s_nop" and int(m.group(1)) + need_t <= 7:
                    # the compiler's own pad sits right here: lengthen it IN PLACE (same 4 bytes: the code layout does not move)
                    k = int(m.group(1)) + need_t
                    out[-1] = f"\ts_nop {k}"
                    run[-1] = dict(op="s_nop", ops=[str(k)], writes=set(), reads=set(), slots=k + 1, text=f"s_nop {k}")
                    inplace = globals().setdefault("_inplace", [0])
                    inplace[0] += 1
                elif mode == "trans_use":
                    out.append(f"\ts_nop {need_t - 1}")
                    run.append(dict(op="s_nop", ops=[str(need_t - 1)], writes=set(), reads=set(), slots=need_t, text=f"s_nop {need_t - 1}"))
                else:
                    pads -= 1                                   # trans_use_inplace: never move code
                pads += 1
            icount += 1
            out.append(raw)
            run.append(ins)
            if ins["op"].startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                run = []
            continue
        if mode == "shift_at":
            if icount == shift_k:
                out.extend(["\ts_nop 0"] * shift_n)
                pads += shift_n
            icount += 1
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
                        why = (d, prev["text"])
                    # keep looking: a farther writer of another operand may still be inside the window
        if need > 0 and only is not None:
            site += 1
            print(f"  {kernel_name[:40]} site {site}: {why[1]}  -> {why[0]} slots ->  {ins['text']}{'   [PADDED]' if site in only else ''}")
            if site not in only:
                need = 0
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
    print(f"asm_patch {mode} {n_arg}: {pads} pads inserted" + (f" ({globals()['_inplace'][0]} of them by lengthening an existing s_nop in place)" if "_inplace" in globals() else ""))


if __name__ == "__main__":
    main()
