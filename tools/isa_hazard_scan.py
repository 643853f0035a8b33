#!/usr/bin/env python
"""tools/isa_hazard_scan.py listing.s kernel-substring [max_slots] -- static pass over a hipcc -S listing: for every v_mfma of the kernel, the
nearest preceding writer of each of its source operands (A, B, C) inside the same straight-line run, with the distance in ISSUE SLOTS
(an instruction = 1 slot, `s_nop N` = N + 1).  Lists every (writer -> v_mfma operand) pair closer than `max_slots` (default 6), a histogram of
the minimum distance per MFMA and per writer opcode, and -- for the LDS hand-offs -- every ds_read whose nearest preceding ds_write in the run has
no s_waitcnt / barrier in between (same-wave LDS traffic is in order, so this is informational).

Round 3 use: the guarded / unguarded builds of k_shade_mfma<float,2> (profiles/r03/hazard.txt)."""
import collections
import re
import sys

REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def split_ops(rest):
    ops, depth, cur = [], 0, ""
    for ch in rest:
        if ch == "[":
            depth += 1
        if ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops


NO_DST = ("ds_write", "global_store", "buffer_store", "flat_store", "s_", "v_cmp", "v_cmpx", "ds_add", "global_atomic", "buffer_atomic", "v_nop", "ds_nop")


def parse(line):
    line = line.split(";")[0].strip()
    if not line or line.startswith("."):
        return None
    parts = line.split(None, 1)
    op = parts[0]
    ops = split_ops(parts[1]) if len(parts) > 1 else []
    writes, reads = set(), set()
    if op.startswith("v_permlane32_swap") or op.startswith("v_permlane16_swap") or op.startswith("v_swap"):
        for o in ops[:2]:
            writes |= regs(o); reads |= regs(o)
    elif op.startswith(NO_DST) and not op.startswith(("ds_add_rtn", "global_atomic")):
        for o in ops:
            reads |= regs(o)
    else:
        if ops:
            writes |= regs(ops[0])
        for o in ops[1:]:
            reads |= regs(o)
        if op.startswith(("v_mac", "v_fmac", "v_pk_fmac", "v_dot2c", "v_writelane")):
            reads |= regs(ops[0])
    slots = 1
    if op == "s_nop":
        slots = int(ops[0], 0) + 1
    return dict(op=op, ops=ops, writes=writes, reads=reads, slots=slots, text=line)


def main():
    path, kern = sys.argv[1], sys.argv[2]
    max_slots = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    s = open(path).read()
    i = s.index(kern)
    i = s.index("\n", s.index(":", i))
    body = s[i:s.index(".end_amdhsa_kernel", i)].split("\n")
    run = []                                    # straight-line run: list of parsed instructions (reset at labels / branches)
    near = []                                   # (distance, writer, operand name, mfma text)
    hist = collections.Counter()
    by_writer = collections.Counter()
    n_mfma = 0
    lds_unfenced = 0
    for raw in body:
        t = raw.strip()
        if t.startswith(".LBB") or re.match(r"^[A-Za-z_.$][\w.$]*:", t):
            run = []
            continue
        ins = parse(raw)
        if ins is None:
            continue
        if ins["op"].startswith("v_mfma"):
            n_mfma += 1
            names = ("A", "B", "C")
            best = None
            for nm, o in zip(names, ins["ops"][1:4]):
                want = regs(o)
                if not want:
                    continue
                d = 0
                for prev in reversed(run):
                    d += prev["slots"]
                    if prev["writes"] & want:
                        if d < max_slots and not prev["op"].startswith("v_mfma"):
                            near.append((d, prev["text"], nm, ins["text"]))
                        if not prev["op"].startswith("v_mfma"):
                            by_writer[(prev["op"], nm, min(d, 9))] += 1
                            best = d if best is None else min(best, d)
                        break
            hist[min(best, 12) if best is not None else "none in run"] += 1
        if ins["op"].startswith("ds_read"):
            for prev in reversed(run):
                if prev["op"].startswith(("s_waitcnt", "s_barrier")):
                    break
                if prev["op"].startswith("ds_write"):
                    lds_unfenced += 1
                    break
        run.append(ins)
        if ins["op"].startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
            run = []
    print(f"kernel {kern}: {n_mfma} v_mfma")
    print("minimum distance (issue slots) from the nearest non-MFMA writer of any source operand, per v_mfma:")
    for k in sorted(hist, key=lambda v: (isinstance(v, str), v)):
        print(f"   {k:>12}: {hist[k]}")
    print("writer opcode -> operand, by distance (9 = 9 or more):")
    for (op, nm, d), n in sorted(by_writer.items(), key=lambda kv: (kv[0][2], kv[0][0])):
        if d < max_slots:
            print(f"   {d} slots  {op:28s} -> src{nm}: {n}")
    print(f"pairs closer than {max_slots} slots: {len(near)}")
    for d, w, nm, m in sorted(near)[:40]:
        print(f"   {d}: {w}   ->   src{nm} of {m}")
    print(f"ds_read with a ds_write earlier in the same run and no s_waitcnt / s_barrier between them: {lds_unfenced}")


if __name__ == "__main__":
    main()
