"""Assembly post-pass of the library build: at least N issue slots (build.TRANS_USE_WAIT_STATES, 4) between a transcendental VALU instruction and the
instruction that reads its result.

Why (round 3, profiles/r03/hazard.txt): on gfx950 the quarter-rate instructions (v_exp / v_rcp / v_rsq / v_sqrt / v_log / v_sin / v_cos: 16 lanes per
pass) hand their result to a following VALU instruction through a software-managed hazard; ROCm 7.2's hazard recogniser (VALUTransUseHazard) pads it
to ONE wait state -- the `s_nop 0` in `v_exp_f32 a; v_exp_f32 b; s_nop 0; v_pk_add_f32 ..a:b..`.  With two waves on a SIMD that is not always enough:
the consumer occasionally reads the register before the last 16-lane pass has been written.  In the shading kernel this showed as renders that
differed run to run on groups of exactly 16 neighbouring rays (r02: "fixed" by scheduling barriers whose only effect was to move code).  The bisect
that names the pair: any 4-byte shift of the instruction stream in front of instruction 2 400 of `k_shade_mfma<float, 2>` hid the failure, a 64-byte
shift did not, and lengthening ONLY the compiler's own `s_nop 0` behind transcendentals to `s_nop 1` -- byte-for-byte the same code layout -- gave
0 differing renders of 200 at every placement tried, against 40 of 40 without.  With another instruction order 2 slots still failed 5 times in
120 renders, 3 never did; the build uses 4.  One wait state suffices in an isolated loop at every placement (tools/ubench/trans_use_hazard.hip), so
the toolchain's table is not wrong in general; the margin is what is missing.

What: for every kernel in a `hipcc -S --cuda-device-only` listing, walk the instructions -- across labels, branches and loop back-edges too: a block
starts with what its predecessors may have left in flight --; for every instruction that reads a VGPR whose
most recent writer (within the window) is a transcendental, make the number of issue slots between the two at least ``wait_states`` -- by
lengthening an `s_nop` that already sits directly in front of the reader (the usual case: no code moves), else by inserting one.  `s_nop N`
counts N + 1 slots, every other instruction 1.  The pass is idempotent.  ``build.py`` runs it on every source of the library (the rule is a
property of the hardware, not of one kernel) and records the counts in ``lib/postpass_report.json``.
"""
from __future__ import annotations

import re
from typing import Dict, List, Set, Tuple

TRANS = ("v_exp_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_log_", "v_sin_", "v_cos_")
_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
_LABEL = re.compile(r"^[A-Za-z_.$][\w.$]*:")


def _vregs(tok: str) -> Set[int]:
    out: Set[int] = set()
    for m in _REG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def _split(rest: str) -> List[str]:
    ops, depth, cur = [], 0, ""
    for ch in rest:
        depth += ch == "["
        depth -= ch == "]"
        if ch == "," and depth == 0:
            ops.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops


def _parse(line: str):
    """(opcode, VGPRs written, VGPRs read, issue slots) of one listing line, or None for directives / comments / labels"""
    text = line.split(";")[0].strip()
    if not text or text.startswith(".") or _LABEL.match(text):
        return None
    parts = text.split(None, 1)
    op = parts[0]
    ops = _split(parts[1]) if len(parts) > 1 else []
    if op == "s_nop":
        return op, set(), set(), int(ops[0], 0) + 1
    if not op.startswith(("v_", "ds_", "global_", "buffer_", "flat_", "scratch_")):
        return op, set(), set(), 1
    stores = op.startswith(("ds_write", "ds_add", "global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic"))
    swaps = op.startswith(("v_permlane32_swap", "v_permlane16_swap", "v_swap"))
    no_vdst = stores or op.startswith(("v_cmp", "v_cmpx", "v_nop", "v_readlane", "v_readfirstlane"))
    writes: Set[int] = set()
    reads: Set[int] = set()
    for i, o in enumerate(ops):
        if i == 0 and not no_vdst:
            writes |= _vregs(o)
            if swaps or op.startswith(("v_mac", "v_fmac", "v_pk_fmac", "v_dot2c", "v_writelane")):
                reads |= _vregs(o)
        elif i == 1 and swaps:
            writes |= _vregs(o)
            reads |= _vregs(o)
        else:
            reads |= _vregs(o)
    return op, writes, reads, 1


_BRANCH = ("s_cbranch", "s_branch")
_NO_FALLTHROUGH = ("s_branch", "s_endpgm", "s_setpc", "s_swappc")


def _pending_trans(run: List[list], window: int, extra_slots: int = 0) -> Dict[int, int]:
    """VGPRs whose most recent writer within `window` issue slots of the end of `run` is a transcendental -> the slots issued since then"""
    pending: Dict[int, int] = {}
    seen: Set[int] = set()
    d = extra_slots
    for prev in reversed(run):
        if d >= window:
            break
        fresh = prev[1] - seen
        if fresh and prev[0].startswith(TRANS):
            for r in fresh:
                pending[r] = d
        seen |= prev[1]
        d += prev[3]
    return pending


def _history(pendings: List[Dict[int, int]]) -> List[list]:
    """a synthetic run that stands for 'any of these predecessors came before': every pending register at its SMALLEST distance"""
    merged: Dict[int, int] = {}
    for p in pendings:
        for r, d in p.items():
            merged[r] = min(d, merged.get(r, 1 << 30))
    run: List[list] = []
    last = None
    for d in sorted(set(merged.values()), reverse=True):       # oldest first
        if last is not None and last - d > 0:
            run.append(["(gap)", set(), set(), last - d, -1])
        run.append(["v_exp_(predecessor)", {r for r, dd in merged.items() if dd == d}, set(), 0, -1])
        last = d
    if last:
        run.append(["(gap)", set(), set(), last, -1])
    return run


def _walk(listing: str, wait_states: int, edit: bool):
    """one walk over the kernels of a listing, control flow included: a block that is entered by a branch (loop back-edges too) or by fall-through
    starts with the transcendental results its predecessors may have left in flight.  edit=False: only measure (returns the closest pair)."""
    lines = listing.split("\n")
    # pass 1: what is pending at every branch, per target label (the branch itself is one issue slot)
    at_label: Dict[str, List[Dict[int, int]]] = {}
    run: List[list] = []
    for raw in lines:
        t = raw.strip()
        m = _LABEL.match(t)
        if m:
            continue                                            # (fall-through keeps the run; pass 2 merges the branch predecessors in)
        ins = _parse(raw)
        if ins is None:
            continue
        op, writes, reads, slots = ins
        run.append([op, writes, reads, slots, -1])
        if op.startswith(_BRANCH):
            target = t.split()[-1]
            pend = _pending_trans(run, wait_states + 8)
            if pend:
                at_label.setdefault(target, []).append(pend)
        if op.startswith(_NO_FALLTHROUGH):
            run = []
    # pass 2
    out: List[str] = []
    run = []
    stats = dict(trans_instructions=0, pairs_closer_than_required=0, lengthened_in_place=0, inserted=0)
    closest = 1 << 30
    in_kernel = False
    for raw in lines:
        t = raw.strip()
        if t.startswith((".amdhsa_kernel", ".end_amdhsa_kernel")):
            in_kernel = False
        elif re.match(r"^[\w$.]+:\s*(;.*)?$", t) and not t.startswith(".L"):
            in_kernel, run = True, []                           # a function / kernel entry label
        if not in_kernel:
            out.append(raw)
            continue
        m = _LABEL.match(t)
        if m:
            name = t.split(":")[0]
            preds = list(at_label.get(name, []))
            fall = _pending_trans(run, wait_states + 8)
            if fall:
                preds.append(fall)
            run = _history(preds)
            out.append(raw)
            continue
        ins = _parse(raw)
        if ins is None:
            out.append(raw)
            continue
        op, writes, reads, slots = ins
        if op.startswith(TRANS):
            stats["trans_instructions"] += 1
        need = 0
        if reads and op.startswith("v_"):
            d = 0
            pending = set(reads)
            for prev in reversed(run):
                if d >= max(wait_states, 16) or not pending:
                    break
                hit = prev[1] & pending
                if hit:
                    if prev[0].startswith(TRANS):
                        closest = min(closest, d)
                        need = max(need, wait_states - d)
                    pending -= hit                              # the nearest writer decides
                d += prev[3]
        if need > 0 and edit:
            stats["pairs_closer_than_required"] += 1
            if run and run[-1][0] == "s_nop" and run[-1][4] >= 0 and run[-1][3] + need <= 8:
                run[-1][3] += need
                out[run[-1][4]] = f"\ts_nop {run[-1][3] - 1}"
                stats["lengthened_in_place"] += 1
            else:
                out.append(f"\ts_nop {need - 1}")
                run.append(["s_nop", set(), set(), need, len(out) - 1])
                stats["inserted"] += 1
        out.append(raw)
        run.append([op, writes, reads, slots, len(out) - 1])
        if op.startswith(_NO_FALLTHROUGH):
            run = []
    return "\n".join(out), stats, closest


def pad_trans_use(listing: str, wait_states: int = 2) -> Tuple[str, Dict[str, int]]:
    out, stats, _ = _walk(listing, wait_states, True)
    return out, stats


def closest_trans_use(listing: str) -> int:
    """smallest number of issue slots between a transcendental and the first VALU reader of its result in the listing, across branches and
    fall-through as well (a large number if there is no such pair): the invariant the build asserts after the pass"""
    return _walk(listing, 0, False)[2]
