"""Assembly post-pass of the library build.

ROUND 6 -- what the pass is FOR now (the block at the end of this file: ``unpack_cross_half``, ``scan_packed_cross_half``): every packed fp32 instruction whose
op_sel / op_sel_hi bits read across the halves of a VGPR source pair is split into two plain instructions, and the build fails if one is left in a linked code
object.  That instruction kind is what made the fused render differ from run to run with two waves on a SIMD (rounds 2 - 5); the per-ray trace hashes and the
in-kernel re-evaluation that found it are tools/trace_check.py and tools/blend_check.py.

The rest of this docstring and the walk below are the two PADDING rules of rounds 3 and 5, kept for experiments and OFF in the default build (build.py): at least
N issue slots (``wait_states``) between a transcendental VALU instruction and the instruction that reads its result, and (``SWAP_MFMA_WAIT_STATES``) between a
`v_permlane32_swap` and a matrix instruction that reads one of the swapped registers.  Both were adopted on dose-response evidence that round 6 explains as timing
side effects -- padding moves the two waves of a SIMD against each other -- of the packed instructions above: with those split, every arrangement that failed at
some padding is clean at the toolchain's own distances (profiles/r06/e_loud_arrangements_split.txt).

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

r04 (the r03 advisor's holes): (1) a call (`s_swappc`) or a function entry leaves EVERY VGPR pending -- the callee / caller may have produced any of
them with a transcendental -- instead of clearing the history; (2) the per-label pending sets are iterated to a fixpoint, so what a predecessor's
predecessor left in flight reaches a block through a short intermediate block; (3) the single-index register syntax `v[7]` is parsed; (4) LDS-DMA
loads (`buffer_load ... lds`, `global_load_lds_*`) have no VGPR destination: their first operand is an address that is READ; (5) the invariant is
re-checked on the LINKED CODE OBJECT by ``verify_code_object`` -- `llvm-objdump -d` of what the device will run, tokenised by a second, deliberately
dumb scanner (any mention of the register by a VALU instruction counts, whichever operand) that shares no code with the listing parser.
"""
from __future__ import annotations

import re
from typing import Dict, List, Set, Tuple

import os as _os
TRANS = ("v_exp_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_log_", "v_sin_", "v_cos_") + tuple(x for x in _os.environ.get("SSDNERF_POSTPASS_EXTRA_PRODUCERS", "").split(",") if x)
_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]|\bv\[(\d+)\]")
ALL_VGPRS = frozenset(range(512))
_LABEL = re.compile(r"^[A-Za-z_.$][\w.$]*:")


def _vregs(tok: str) -> Set[int]:
    out: Set[int] = set()
    for m in _REG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        elif m.group(4) is not None:
            out.add(int(m.group(4)))
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
    lds_dma = op.startswith("global_load_lds") or (op.startswith(("buffer_load", "global_load")) and re.search(r"\blds\b", text.split(None, 1)[1] if len(parts) > 1 else ""))
    no_vdst = stores or lds_dma or op.startswith(("v_cmp", "v_cmpx", "v_nop", "v_readlane", "v_readfirstlane"))
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
_NO_FALLTHROUGH = ("s_branch", "s_endpgm", "s_setpc")
_CALL = ("s_swappc",)


def _all_pending() -> list:
    """history entry for 'any VGPR may just have been written by a transcendental' (function entry, return from a call)"""
    return ["v_exp_(unknown code)", set(ALL_VGPRS), set(), 0, -1]


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


#: Bisect aid (tools/hazard_bisect.py, r05): (kernel name substring, first site, last site + 1, wait states inside the range).  A "site" is a reader that the
#: pass would pad at `inside` wait states, counted in listing order inside the named kernel; sites in the range get `inside`, every other pair the build's own
#: distance.  None in product builds.
SITE_FILTER = None
SITE_COUNT: Dict[str, int] = {}                 # sites seen per kernel by the last filtered walk (the bisect's upper bound)

#: Second rule (r05, last hours; build.SWAP_MFMA_WAIT_STATES): issue slots between a `v_permlane32_swap` / `v_permlane16_swap` and a matrix instruction that reads one
#: of the swapped registers as its A or B operand.  The toolchain pads that pair like any VALU write -> MFMA operand read (2 wait states); the 12 000-render soaks of
#: the shading kernel (profiles/r05/zz_soak_reproducibility.txt) show one quarter-wave with a stale low-order operand term about once in 3 000 renders, whatever the
#: transcendental rule's distance, and the swaps that put the features' split terms into B-operand order are the only VALU producers within 24 slots of a matrix
#: instruction's operand read in that kernel (one of them at 2).  0 = rule off.  Straight-line code only (a label clears the swap history).
SWAP_MFMA_WAIT_STATES = 0
_SWAPS = ("v_permlane32_swap", "v_permlane16_swap")
#: Third rule (r06; build.VALU_MFMA_WAIT_STATES): the second rule for EVERY VALU producer -- issue slots between any VALU instruction (a matrix instruction is not
#: one) that writes a VGPR and a matrix instruction that reads that VGPR as its A or B operand.  The toolchain's own distance for the pair is 2 wait states.  0 = off.
#: Loads (LDS, global) are not VALU producers: their results are counted (`s_waitcnt`), not timed.
VALU_MFMA_WAIT_STATES = 0


def _walk(listing: str, wait_states: int, edit: bool):
    """one walk over the kernels of a listing, control flow included: a block that is entered by a branch (loop back-edges too) or by fall-through
    starts with the transcendental results its predecessors may have left in flight.  edit=False: only measure (returns the closest pair)."""
    lines = listing.split("\n")
    kernels = {m.group(1) for m in re.finditer(r"^\s*\.amdhsa_kernel\s+(\S+)", listing, re.M)}   # entry points: the hardware starts them with nothing in flight

    def entry_state(label_line: str) -> list:
        return [] if label_line.split(":")[0] in kernels else [_all_pending()]
    # pass 1: what is pending at every branch, per target label (the branch itself is one issue slot).  A block's own start state depends on the
    # pending sets of ITS predecessors, so the sets are iterated to a fixpoint (distances only shrink and are bounded by the window)
    window = wait_states + 8
    at_label: Dict[str, Dict[int, int]] = {}
    for _ in range(16):
        new_at: Dict[str, Dict[int, int]] = {}
        run: List[list] = []
        for raw in lines:
            t = raw.strip()
            if _LABEL.match(t):
                name = t.split(":")[0]
                if re.match(r"^[\w$.]+:\s*(;.*)?$", t) and not t.startswith(".L"):
                    run = entry_state(t)                        # a function / kernel entry
                else:
                    preds = [at_label[name]] if name in at_label else []
                    fall = _pending_trans(run, window)
                    if fall:
                        preds.append(fall)
                    run = _history(preds)
                continue
            ins = _parse(raw)
            if ins is None:
                continue
            op, writes, reads, slots = ins
            run.append([op, writes, reads, slots, -1])
            if op.startswith(_CALL):
                run = [_all_pending()]
            if op.startswith(_BRANCH):
                target = t.split(";")[0].split()[-1]
                pend = _pending_trans(run, window)
                if pend:
                    cur = new_at.setdefault(target, {})
                    for r, d in pend.items():
                        cur[r] = min(d, cur.get(r, 1 << 30))
            if op.startswith(_NO_FALLTHROUGH):
                run = []
        if new_at == at_label:
            break
        at_label = new_at
    # pass 2
    out: List[str] = []
    run = []
    stats = dict(trans_instructions=0, pairs_closer_than_required=0, lengthened_in_place=0, inserted=0)
    closest = 1 << 30
    in_kernel = False
    site, kernel_name = 0, ""
    for raw in lines:
        t = raw.strip()
        if t.startswith((".amdhsa_kernel", ".end_amdhsa_kernel")):
            in_kernel = False
        elif re.match(r"^[\w$.]+:\s*(;.*)?$", t) and not t.startswith(".L"):
            in_kernel, run = True, entry_state(t)               # a device FUNCTION's caller may have left anything in flight; a kernel starts clean
            kernel_name, site = t.split(":")[0], 0
            out.append(raw)
            continue
        if not in_kernel:
            out.append(raw)
            continue
        m = _LABEL.match(t)
        if m:
            name = t.split(":")[0]
            preds = [at_label[name]] if name in at_label else []
            fall = _pending_trans(run, window)
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
            need_inside = 0
            for prev in reversed(run):
                if d >= max(wait_states, 16) or not pending:
                    break
                hit = prev[1] & pending
                if hit:
                    if prev[0].startswith(TRANS):
                        closest = min(closest, d)
                        need = max(need, wait_states - d)
                        if SITE_FILTER is not None:
                            need_inside = max(need_inside, SITE_FILTER[3] - d)
                    pending -= hit                              # the nearest writer decides
                d += prev[3]
            if SITE_FILTER is not None and edit and SITE_FILTER[0] in kernel_name and need_inside > 0:
                if SITE_FILTER[1] <= site < SITE_FILTER[2]:
                    need = max(need, need_inside)
                site += 1
                SITE_COUNT[kernel_name] = site
        if (SWAP_MFMA_WAIT_STATES > 0 or VALU_MFMA_WAIT_STATES > 0) and op.startswith("v_mfma"):
            ops_ = _split(t.split(";")[0].split(None, 1)[1])
            pending = (_vregs(ops_[1]) | _vregs(ops_[2])) if len(ops_) >= 3 else set()
            d = 0
            for prev in reversed(run):
                if d >= max(SWAP_MFMA_WAIT_STATES, VALU_MFMA_WAIT_STATES) or not pending:
                    break
                hit = prev[1] & pending
                if hit:
                    if prev[0].startswith(_SWAPS) and SWAP_MFMA_WAIT_STATES - d > 0:
                        need = max(need, SWAP_MFMA_WAIT_STATES - d)
                        stats["swap_mfma_pairs_padded"] = stats.get("swap_mfma_pairs_padded", 0) + 1
                    if prev[0].startswith("v_") and not prev[0].startswith("v_mfma") and VALU_MFMA_WAIT_STATES - d > 0:
                        need = max(need, VALU_MFMA_WAIT_STATES - d)
                        stats["valu_mfma_pairs_padded"] = stats.get("valu_mfma_pairs_padded", 0) + 1
                    pending -= hit
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
        if op.startswith(_CALL):
            run = [_all_pending()]                              # whatever the callee did last
        if op.startswith(_NO_FALLTHROUGH):
            run = []
    return "\n".join(out), stats, closest


def pad_trans_use(listing: str, wait_states: int = 4) -> Tuple[str, Dict[str, int]]:
    out, stats, _ = _walk(listing, wait_states, True)
    return out, stats


def closest_trans_use(listing: str) -> int:
    """smallest number of issue slots between a transcendental and the first VALU reader of its result in the listing, across branches and
    fall-through as well (a large number if there is no such pair): the invariant the build asserts after the pass"""
    return _walk(listing, 0, False)[2]


# ---------------------------------------------------------------------------------------------- independent check on the linked code object
_DIS_REG = re.compile(r"(?<![\w.])v(\d+)(?![\w\[])|(?<![\w.])v\[(\d+)(?::(\d+))?\]")


def _mentions(operands: str) -> Set[int]:
    regs: Set[int] = set()
    for a, lo, hi in _DIS_REG.findall(operands):
        if a:
            regs.add(int(a))
        else:
            regs.update(range(int(lo), int(hi or lo) + 1))
    return regs


def verify_code_object(path: str, wait_states: int, objdump: str = "/opt/rocm/lib/llvm/bin/llvm-objdump", swap_mfma_wait_states: int = 0) -> Dict[str, int]:
    """Check the trans -> use rule on the instructions the device will execute: `llvm-objdump -d --symbolize-operands` of the linked code object.
    Deliberately NOT the listing parser: every instruction is (mnemonic, set of VGPRs it mentions anywhere); from each transcendental, every
    control-flow path (both sides of conditional branches, back-edges included) is followed for ``wait_states`` issue slots, and the first VALU
    instruction on a path that mentions a destination register of the transcendental must not be closer than that.  Mentions as a destination count
    too (conservative: an overwrite this close would be flagged; none exists in the library).  ``swap_mfma_wait_states`` > 0 checks the second rule the
    same way: from each `v_permlane32_swap` / `v_permlane16_swap`, the first MATRIX instruction on a path that mentions one of the two swapped registers
    (any operand: conservative).  Returns counts; raises RuntimeError on a violation."""
    import subprocess
    text = subprocess.run([objdump, "-d", "--symbolize-operands", path], check=True, capture_output=True, text=True).stdout
    ins: List[Tuple[str, str]] = []
    labels: Dict[str, int] = {}
    for line in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        body = line.split("//")[0].strip()
        if not body or not line.startswith(("\t", " ")):
            continue
        parts = body.split(None, 1)
        ins.append((parts[0], parts[1] if len(parts) > 1 else ""))

    def closest_reader(i: int, dst: Set[int], limit: int, reader_prefix: str):
        """(slots, index) of the nearest instruction starting with `reader_prefix` that mentions a register of `dst`, over every path of `limit` slots from i + 1"""
        best, at = 1 << 30, None
        stack = [(i + 1, 0, frozenset(dst))]
        seen = set()
        while stack:
            j, used, live = stack.pop()
            while j < len(ins) and used < limit and live:
                if (j, used, live) in seen:
                    break
                seen.add((j, used, live))
                o, args = ins[j]
                if o == "s_nop":
                    used += int(args.strip(), 0) + 1
                    j += 1
                    continue
                if o.startswith("v_"):
                    hit = _mentions(args) & live
                    if hit:
                        if o.startswith(reader_prefix) and used < best:
                            best, at = used, j
                        live = live - hit                            # (a VALU instruction that is not a reader re-defines or consumes the register: the nearest mention decides)
                if o.startswith(("s_endpgm", "s_setpc")):
                    break
                if o.startswith("s_branch"):
                    tgt = args.split()[0]
                    j = labels.get(tgt, len(ins))
                    used += 1
                    continue
                if o.startswith("s_cbranch"):
                    tgt = args.split()[0]
                    if tgt in labels:
                        stack.append((labels[tgt], used + 1, live))
                used += 1
                j += 1
        return best, at

    n_trans = n_swaps = 0
    closest = swap_closest = 1 << 30
    worst = swap_worst = None
    for i, (op, operands) in enumerate(ins):
        if op.startswith(TRANS):
            n_trans += 1
            d, j = closest_reader(i, _mentions(operands.split(",")[0]), wait_states + 4, "v_")      # (4 slots beyond the rule, so that the report shows the actual margin)
            if d < closest:
                closest, worst = d, (i, j)
        elif swap_mfma_wait_states > 0 and op.startswith(_SWAPS):
            n_swaps += 1
            d, j = closest_reader(i, _mentions(operands), swap_mfma_wait_states + 4, "v_mfma")
            if d < swap_closest:
                swap_closest, swap_worst = d, (i, j)
    if closest < wait_states:
        a, b = worst
        raise RuntimeError(f"{path}: `{ins[a][0]} {ins[a][1]}` is read {closest} issue slots later by `{ins[b][0]} {ins[b][1]}` in the linked code object "
                           f"(instructions {a} and {b}); the build requires {wait_states}")
    if swap_closest < swap_mfma_wait_states:
        a, b = swap_worst
        raise RuntimeError(f"{path}: `{ins[a][0]} {ins[a][1]}` is read {swap_closest} issue slots later by `{ins[b][0]} {ins[b][1]}` in the linked code object "
                           f"(instructions {a} and {b}); the build requires {swap_mfma_wait_states} between a lane swap and a matrix instruction")
    out = dict(trans_instructions=n_trans, closest_pair=None if closest == 1 << 30 else closest)
    if swap_mfma_wait_states > 0:
        out.update(swap_instructions=n_swaps, closest_swap_mfma_pair=None if swap_closest == 1 << 30 else swap_closest)
    return out


# ---------------------------------------------------------------------------------------------- r06: packed fp32 instructions with cross-half operand selection
#: Round 6 (profiles/r06/README.md).  The run-to-run differences of the shading kernel that need two waves per SIMD -- r02's "groups of 16
#: neighbouring rays", r03's transcendental theory, r05's swap theory -- were traced, with per-ray hashes of every stage and an in-kernel re-evaluation
#: (tools/trace_check.py, tools/blend_check.py), to ONE kind of instruction: a packed fp32 operation (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) whose `op_sel` / `op_sel_hi`
#: bits make a half of the result read the OTHER half of a VGPR source pair (the compiler's way of broadcasting one weight to both halves).  On gfx950 with two waves on a
#: SIMD the low half of such an instruction's result occasionally comes out, in lanes 48-63 only, as if the product term were absent.  The same arithmetic as plain
#: v_mul / v_fma, or packed with the broadcast materialised in a register pair, never does.  The library therefore contains NO such instruction: the sources avoid them
#: (explicit pairs or pinned plain instructions) and this scan, run by build.py on every listing and on the linked code object, fails the build if the compiler forms one.
_PK_F32 = ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32")
_SEL = re.compile(r"\b(op_sel|op_sel_hi):\[([01,]+)\]")


def packed_cross_half(line: str):
    """None, or (mnemonic, [source operand indices whose halves are crossed]) for one listing / disassembly line: a packed fp32 instruction that reads, for its low
    result, the high register of a VGPR source pair (op_sel bit 1) or, for its high result, the low register (op_sel_hi bit 0)"""
    text = line.split(";")[0].split("//")[0].strip()
    parts = text.split(None, 1)
    if len(parts) < 2 or not parts[0].startswith(_PK_F32):
        return None
    mods = dict((m.group(1), [int(x) for x in m.group(2).split(",")]) for m in _SEL.finditer(parts[1]))
    body = _SEL.sub("", parts[1])
    body = re.sub(r"\b(neg_lo|neg_hi):\[[01,]+\]|\bclamp\b", "", body)
    ops = _split(body)
    srcs = ops[1:]
    n = len(srcs)
    sel = mods.get("op_sel", [0] * n) + [0] * n
    sel_hi = mods.get("op_sel_hi", [1] * n) + [1] * n
    crossed = [i for i, o in enumerate(srcs) if _vregs(o) and (sel[i] == 1 or sel_hi[i] == 0)]
    return (parts[0], crossed) if crossed else None


def scan_packed_cross_half(text: str) -> Dict[str, int]:
    """kernel / function name -> number of packed fp32 instructions with a crossed VGPR source, over a device listing (`hipcc -S`) or an `llvm-objdump -d` text"""
    out: Dict[str, int] = {}
    cur = "?"
    for raw in text.split("\n"):
        t = raw.strip()
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", t) or re.match(r"^([A-Za-z_$][\w$.]*):\s*(;.*)?$", t)
        if m and not m.group(1).startswith((".L", "L")):
            cur = m.group(1)
            continue
        if packed_cross_half(raw) is not None:
            out[cur] = out.get(cur, 0) + 1
    return out


_PLAIN = {"v_pk_fma_f32": "v_fma_f32", "v_pk_mul_f32": "v_mul_f32_e64", "v_pk_add_f32": "v_add_f32_e64"}
_PAIR = re.compile(r"^([vs])\[(\d+):(\d+)\]$")
_BITS = re.compile(r"\b(op_sel|op_sel_hi|neg_lo|neg_hi):\[([01,]+)\]")


def split_packed_cross_half(line: str):
    """The two plain instructions that compute what a crossed packed fp32 instruction computes (see above), as listing lines; None if `line` is not such an
    instruction.  D.lo = f(src_i[op_sel_i]), D.hi = f(src_i[op_sel_hi_i]) element-wise -- a packed fp32 operation is two independent IEEE operations, so the results are
    bit-identical.  Raises ValueError for a form that cannot be split in place (destination overlapping a source of the other half both ways; a non-zero constant read
    as a high half)."""
    if packed_cross_half(line) is None:
        return None
    indent = line[: len(line) - len(line.lstrip())]
    text = line.split(";")[0].strip()
    op, rest = text.split(None, 1)
    bits = dict((m.group(1), [int(x) for x in m.group(2).split(",")]) for m in _BITS.finditer(rest))
    clamp = bool(re.search(r"\bclamp\b", rest))
    body = re.sub(r"\bclamp\b", "", _BITS.sub("", rest))
    ops = _split(body)
    dst, srcs = ops[0], ops[1:]
    n = len(srcs)
    sel = (bits.get("op_sel", []) + [0] * n)[:n]
    sel_hi = (bits.get("op_sel_hi", []) + [1] * n)[:n]
    neg_lo = (bits.get("neg_lo", []) + [0] * n)[:n]
    neg_hi = (bits.get("neg_hi", []) + [0] * n)[:n]
    md = _PAIR.match(dst)
    if not md or md.group(1) != "v":
        raise ValueError(f"cannot split `{text}`: destination is not a VGPR pair")
    d_lo, d_hi = int(md.group(2)), int(md.group(3))

    def pick(o: str, which: int):
        m = _PAIR.match(o)
        if m:
            return f"{m.group(1)}{int(m.group(2)) + which}", (m.group(1), int(m.group(2)) + which)
        if which == 1 and o.strip() not in ("0", "0.0"):
            raise ValueError(f"cannot split `{text}`: constant operand `{o}` read as a high half")
        return o.strip(), None
    lo, hi = [], []
    for i, o in enumerate(srcs):
        a, ra = pick(o, sel[i])
        b, rb = pick(o, sel_hi[i])
        lo.append((("-" if neg_lo[i] else "") + a, ra))
        hi.append((("-" if neg_hi[i] else "") + b, rb))
    plain = _PLAIN[next(p for p in _PK_F32 if op.startswith(p))]
    tail = " clamp" if clamp else ""
    lo_ins = f"{indent}{plain} v{d_lo}, " + ", ".join(t for t, _ in lo) + tail
    hi_ins = f"{indent}{plain} v{d_hi}, " + ", ".join(t for t, _ in hi) + tail
    lo_first_ok = all(r != ("v", d_lo) for _, r in hi)          # the high half must not read what the low half has just overwritten
    hi_first_ok = all(r != ("v", d_hi) for _, r in lo)
    if lo_first_ok:
        return [lo_ins, hi_ins]
    if hi_first_ok:
        return [hi_ins, lo_ins]
    # both halves read what the other writes: fine if they compute the same value (x * y into both halves of [x:y], x + y and y + x, ...) -- one operation and a copy
    lt, ht = [t for t, _ in lo], [t for t, _ in hi]
    same = (sorted(lt[:2]) == sorted(ht[:2]) and lt[2:] == ht[2:])          # (add, mul and the product of an fma commute: the IEEE result does not depend on the order)
    if same:
        return [lo_ins, f"{indent}v_mov_b32_e32 v{d_hi}, v{d_lo}"]
    raise ValueError(f"cannot split `{text}` in place: each half's destination is a source of the other half")


def unpack_cross_half(listing: str) -> Tuple[str, Dict[str, int]]:
    """every crossed packed fp32 instruction of a device listing as two plain instructions; returns (listing, {'split': n})"""
    out, n = [], 0
    for raw in listing.split("\n"):
        two = split_packed_cross_half(raw)
        if two is None:
            out.append(raw)
        else:
            out.extend(two)
            n += 1
    return "\n".join(out), {"packed_cross_half_split": n}


def device_code_objects(path: str, arch: str = "gfx950") -> List[bytes]:
    """the device code objects embedded in a host object / shared library built by hipcc: every clang offload bundle (`__CLANG_OFFLOAD_BUNDLE__`) in the file, the
    entries whose target triple names `arch`"""
    import struct
    data = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out: List[bytes] = []
    at = data.find(magic)
    while at >= 0:
        n = struct.unpack_from("<Q", data, at + len(magic))[0]
        p = at + len(magic) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24: p + 24 + tl].decode("ascii", "replace")
            p += 24 + tl
            if arch in triple and size > 0:
                out.append(data[at + off: at + off + size])
        at = data.find(magic, at + len(magic))
    return out


def disassemble_library(path: str, objdump: str = "/opt/rocm/lib/llvm/bin/llvm-objdump", arch: str = "gfx950") -> str:
    """`llvm-objdump -d` of every device code object embedded in `path`, concatenated"""
    import subprocess
    import tempfile
    text = ""
    with tempfile.TemporaryDirectory() as td:
        for i, blob in enumerate(device_code_objects(path, arch)):
            f = f"{td}/co{i}.out"
            with open(f, "wb") as fh:
                fh.write(blob)
            text += subprocess.run([objdump, "-d", f], check=True, capture_output=True, text=True).stdout
    return text


if __name__ == "__main__":
    # python -m ssdnerf_amd.asm_postpass scan FILE...   -- count packed fp32 instructions with crossed VGPR halves per kernel: FILE is a device listing (`hipcc -S
    # --cuda-device-only`), an `llvm-objdump -d` text, or a host object / shared library built by hipcc (its embedded gfx950 code objects are disassembled)
    import sys
    if len(sys.argv) >= 3 and sys.argv[1] == "scan":
        total = 0
        for path in sys.argv[2:]:
            with open(path, "rb") as fh:
                head = fh.read(4)
            text = disassemble_library(path) if head == b"\x7fELF" else open(path).read()
            found = scan_packed_cross_half(text)
            total += sum(found.values())
            print(f"{path}: {sum(found.values())} packed fp32 instructions read across the halves of a VGPR source pair" + ("" if not found else ":"))
            for k, v in sorted(found.items(), key=lambda kv: -kv[1])[:20]:
                print(f"    {v:6d}  {k}")
        sys.exit(1 if total else 0)
    print(__doc__)
