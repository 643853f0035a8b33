"""The build's assembly post-pass (ssdnerf_amd/asm_postpass.py).  Round 6: packed fp32 instructions whose halves read across a VGPR source pair are split into
two plain instructions (the instruction kind behind the run-to-run differences of rounds 2 - 5) and none may be left in the shipped library.  The padding rules of
rounds 3 / 5 (transcendental -> use, swap -> matrix operand) are off in the build and still tested here as functions: every pair gets the required issue slots, the
compiler's own pad is lengthened in place where there is one, nothing else moves, the pass is idempotent.  Hand-written listings and the compiler's real listing of
library sources (hipcc cross-compiles without a GPU)."""
import os
import subprocess

from ssdnerf_amd import build as B
from ssdnerf_amd.asm_postpass import closest_trans_use, pad_trans_use

WS = 4          # the r03 rule's distance, as a function argument (the build's default is the toolchain's own: WS == 1)

LISTING = """
	.text
_Z1kPf:                                 ; @_Z1kPf
	s_load_dwordx2 s[0:1], s[4:5], 0x0
	v_exp_f32_e32 v1, v0
	v_exp_f32_e32 v2, v3
	s_nop 0
	v_pk_add_f32 v[4:5], v[1:2], 1.0 op_sel_hi:[1,0]
	v_rcp_f32_e32 v6, v4
	v_mul_f32_e32 v7, v6, v0
	v_rcp_f32_e32 v8, v5
	v_add_f32_e32 v9, v0, v0
	v_add_f32_e32 v10, v0, v0
	v_add_f32_e32 v11, v0, v0
	v_add_f32_e32 v12, v0, v0
	v_mul_f32_e32 v13, v8, v0
	v_sqrt_f32_e32 v14, v0
.LBB0_1:
	v_mul_f32_e32 v15, v14, v0
.LBB0_2:
	v_mul_f32_e32 v20, v21, v0
	v_rcp_f32_e32 v21, v0
	s_cbranch_scc1 .LBB0_2
	global_store_dword v16, v15, s[0:1]
	s_endpgm
	.amdhsa_kernel _Z1kPf
	.end_amdhsa_kernel
"""


def test_pairs_are_padded_in_place_where_possible_and_the_pass_is_idempotent():
    out, st = pad_trans_use(LISTING, 4)
    lines = [l.strip() for l in out.split("\n")]
    # exp, exp, s_nop 0, pk_add: the existing pad grows to 4 slots (s_nop 3); rcp -> mul directly: s_nop 3 inserted; rcp .. 4 instructions .. mul: untouched;
    # sqrt -> (fall-through into the next block) -> its reader: s_nop 3; rcp at the end of a loop body -> (back-edge: the branch is one slot) -> its
    # reader at the top of the loop: s_nop 2
    assert lines[lines.index("v_pk_add_f32 v[4:5], v[1:2], 1.0 op_sel_hi:[1,0]") - 1] == "s_nop 3"
    assert lines[lines.index("v_mul_f32_e32 v7, v6, v0") - 1] == "s_nop 3"
    assert lines[lines.index("v_mul_f32_e32 v13, v8, v0") - 1] == "v_add_f32_e32 v12, v0, v0"
    assert lines[lines.index("v_mul_f32_e32 v15, v14, v0") - 1] == "s_nop 3"
    assert lines[lines.index("v_mul_f32_e32 v20, v21, v0") - 1] == "s_nop 2"
    assert st == dict(trans_instructions=6, pairs_closer_than_required=4, lengthened_in_place=1, inserted=3)
    assert closest_trans_use(LISTING) == 0 and closest_trans_use(out) >= 4
    again, st2 = pad_trans_use(out, 4)
    assert again == out and st2["pairs_closer_than_required"] == 0
    # everything that is not a pad is unchanged, in order
    assert [l for l in out.split("\n") if "s_nop" not in l] == [l for l in LISTING.split("\n") if "s_nop" not in l]


SWAP_LISTING = """
	.text
_Z1sPf:                                 ; @_Z1sPf
	v_perm_b32 v4, v1, v0, s2
	v_perm_b32 v5, v3, v2, s2
	s_nop 1
	v_permlane32_swap_b32_e32 v4, v5
	v_add_f32_e32 v30, v31, v31
	v_mfma_f32_32x32x16_bf16 v[8:23], v[40:43], v[4:7], v[8:23]
	v_permlane32_swap_b32_e32 v50, v51
	v_add_f32_e32 v30, v31, v31
	v_add_f32_e32 v30, v31, v31
	v_add_f32_e32 v30, v31, v31
	v_add_f32_e32 v30, v31, v31
	v_add_f32_e32 v30, v31, v31
	v_add_f32_e32 v30, v31, v31
	v_add_f32_e32 v30, v31, v31
	v_add_f32_e32 v30, v31, v31
	v_mfma_f32_32x32x16_bf16 v[8:23], v[50:53], v[44:47], v[8:23]
	v_permlane32_swap_b32_e32 v8, v9
	v_mfma_f32_32x32x16_bf16 v[8:23], v[60:63], v[64:67], v[8:23]
	s_endpgm
	.amdhsa_kernel _Z1sPf
	.end_amdhsa_kernel
"""


def test_swap_to_matrix_operand_pairs_get_their_slots():
    """the second rule (r05: profiles/r05/zz_soak_reproducibility.txt): >= N issue slots between a v_permlane32_swap and a matrix instruction that reads a swapped
    register as its A or B operand; a swap eight instructions ahead, or one whose result is only the ACCUMULATOR operand, is left alone; off by default in the module"""
    from ssdnerf_amd import asm_postpass as A
    saved = A.SWAP_MFMA_WAIT_STATES
    try:
        A.SWAP_MFMA_WAIT_STATES = 0
        assert pad_trans_use(SWAP_LISTING, 4)[0] == SWAP_LISTING
        A.SWAP_MFMA_WAIT_STATES = 8
        out, st = pad_trans_use(SWAP_LISTING, 4)
        lines = [l.strip() for l in out.split("\n")]
        first = lines.index("v_mfma_f32_32x32x16_bf16 v[8:23], v[40:43], v[4:7], v[8:23]")
        assert lines[first - 1] == "s_nop 6" and lines[first - 2] == "v_add_f32_e32 v30, v31, v31"          # swap, add (1 slot) -> 7 more
        second = lines.index("v_mfma_f32_32x32x16_bf16 v[8:23], v[50:53], v[44:47], v[8:23]")
        assert lines[second - 1] == "v_add_f32_e32 v30, v31, v31"                                            # eight instructions between: nothing to add
        third = lines.index("v_mfma_f32_32x32x16_bf16 v[8:23], v[60:63], v[64:67], v[8:23]")
        assert lines[third - 1] == "v_permlane32_swap_b32_e32 v8, v9"                                       # the accumulate operand is not this rule's business
        assert st["swap_mfma_pairs_padded"] == 1
        assert pad_trans_use(out, 4)[0] == out
        assert [l for l in out.split("\n") if "s_nop 6" not in l] == SWAP_LISTING.split("\n")
    finally:
        A.SWAP_MFMA_WAIT_STATES = saved


def test_real_listing_of_a_library_source_meets_the_invariant(tmp_path):
    src = os.path.join(B.CSRC, "raygen.hip")
    listing = tmp_path / "raygen.s"
    subprocess.check_call([B._hipcc()] + B.FLAGS + ["-S", "--cuda-device-only", src, "-o", str(listing)], stderr=subprocess.DEVNULL)
    text = listing.read_text()
    assert closest_trans_use(text) <= 1                      # the toolchain pads this hazard to one wait state (or leaves trans -> trans pairs adjacent)
    out, st = pad_trans_use(text, WS)
    assert st["trans_instructions"] > 0 and st["pairs_closer_than_required"] > 0
    assert closest_trans_use(out) >= WS
    assert [l for l in out.split("\n") if "s_nop" not in l] == [l for l in text.split("\n") if "s_nop" not in l]


def test_shipped_library_was_built_with_the_post_pass():
    import json
    report = json.load(open(os.path.join(B.LIB_DIR, "postpass_report.json")))
    assert report["toolchain"]["validated"] is True and set(report["sources"]) == set(B.SOURCES)
    assert report["settings"]["unpack_cross_half"] is True                      # r06: the one rule of the build
    shade = report["sources"]["shade_mfma.hip"]
    assert shade["packed_cross_half_split"] > 500 and shade["code_object_check"]["packed_cross_half"] == 0
    assert sum(v["packed_cross_half_split"] for v in report["sources"].values()) > 2000
    assert all(v["code_object_check"]["packed_cross_half"] == 0 for v in report["sources"].values())
    # the padding rules of r03 / r05 are off: the toolchain's own distance behind transcendentals, no swap -> matrix-operand rule
    assert report["settings"]["wait_states"] == B.TRANS_USE_WAIT_STATES == 1 and report["settings"]["swap_mfma_wait_states"] == B.SWAP_MFMA_WAIT_STATES == 0
    assert shade["trans_instructions"] > 1000 and shade["pairs_closer_than_required"] == 0


def test_no_crossed_packed_instruction_in_the_shipped_library():
    """the linked device code of lib/libssdnerf_hip.so, disassembled here: no v_pk_{fma,mul,add}_f32 reads across the halves of a VGPR source pair"""
    from ssdnerf_amd.asm_postpass import device_code_objects, disassemble_library, scan_packed_cross_half
    assert len(device_code_objects(B.LIB_PATH)) == len(B.SOURCES)                   # one code object per source
    text = disassemble_library(B.LIB_PATH, os.path.join(B.LLVM_BIN, "llvm-objdump"))
    assert text.count("v_mfma_f32_32x32x16_bf16") > 100, "the disassembly does not show the library's device code"
    assert "v_pk_fma_f32" in text                                                    # (uncrossed packed instructions stay)
    assert scan_packed_cross_half(text) == {}


PACKED = """
	v_pk_fma_f32 v[0:1], v[4:5], v[186:187], v[0:1] op_sel:[0,1,0]
	v_pk_mul_f32 v[0:1], v[0:1], v[186:187] op_sel_hi:[1,0]
	v_pk_add_f32 v[4:5], v[0:1], s[80:81] op_sel_hi:[0,1] neg_lo:[1,0] neg_hi:[1,0]
	v_pk_mul_f32 v[186:187], v[186:187], v[2:3] op_sel_hi:[0,1]
	v_pk_mul_f32 v[46:47], v[46:47], v[46:47] op_sel:[0,1] op_sel_hi:[0,1]
	v_pk_add_f32 v[136:137], v[136:137], 1.0 op_sel_hi:[1,0]
	v_pk_fma_f32 v[144:145], v[144:145], v[182:183], 0 op_sel_hi:[1,1,0]
	v_pk_fma_f32 v[8:9], v[2:3], v[4:5], v[6:7] clamp
	v_pk_mov_b32 v[56:57], v[56:57], v[58:59] op_sel:[1,0]
"""


def test_crossed_packed_instructions_are_split_into_the_same_arithmetic():
    """``unpack_cross_half`` on hand-written lines: crossed VGPR sources are split, constants and uncrossed instructions stay; and the two plain instructions compute,
    on random register contents, exactly what the packed instruction's definition says (D.lo = f(src_i[op_sel_i]), D.hi = f(src_i[op_sel_hi_i]))"""
    import re
    import numpy as np
    from ssdnerf_amd.asm_postpass import packed_cross_half, scan_packed_cross_half, split_packed_cross_half, unpack_cross_half
    lines = [l for l in PACKED.split("\n") if l.strip()]
    crossed = [packed_cross_half(l) is not None for l in lines]
    assert crossed == [True, True, True, True, True, False, False, False, False]
    out, st = unpack_cross_half(PACKED)
    assert st["packed_cross_half_split"] == 5 and scan_packed_cross_half(out) == {}
    assert all(l in out for l, c in zip(lines, crossed) if not c)                    # untouched
    rng = np.random.default_rng(0)

    def run_plain(ins, v, s):
        op, rest = ins.strip().split(None, 1)
        clamp = rest.endswith(" clamp")
        ops = [o.strip() for o in rest.replace(" clamp", "").split(",")]
        def val(o):
            neg = o.startswith("-"); o = o.lstrip("-")
            x = v[int(o[1:])] if o[0] == "v" else s[int(o[1:])] if o[0] == "s" else np.float32(float(o))
            return -x if neg else x
        if op == "v_mov_b32_e32":
            r = val(ops[1])
        else:
            a = [val(o) for o in ops[1:]]
            r = {"v_fma_f32": lambda: np.float32(np.float64(a[0]) * np.float64(a[1]) + np.float64(a[2])), "v_mul_f32_e64": lambda: np.float32(a[0] * a[1]),
                 "v_add_f32_e64": lambda: np.float32(a[0] + a[1])}[op]()
        v[int(ops[0][1:])] = np.clip(r, 0, 1).astype(np.float32) if clamp else r

    def run_packed(ins, v, s):
        op, rest = ins.strip().split(None, 1)
        bits = {m.group(1): [int(x) for x in m.group(2).split(",")] for m in re.finditer(r"(op_sel|op_sel_hi|neg_lo|neg_hi):\[([01,]+)\]", rest)}
        ops = [o.strip() for o in re.sub(r"(op_sel|op_sel_hi|neg_lo|neg_hi):\[[01,]+\]", "", rest).split(", ")]
        ops = [o.strip().rstrip(",") for o in ops if o.strip()]
        d = int(re.match(r"v\[(\d+):", ops[0]).group(1))
        n = len(ops) - 1
        res = []
        for half, key, dflt in ((0, "op_sel", 0), (1, "op_sel_hi", 1)):
            a = []
            for i, o in enumerate(ops[1:]):
                sel = (bits.get(key, []) + [dflt] * n)[i]
                neg = (bits.get("neg_lo" if half == 0 else "neg_hi", []) + [0] * n)[i]
                m = re.match(r"([vs])\[(\d+):", o)
                x = (v if m.group(1) == "v" else s)[int(m.group(2)) + sel]
                a.append(-x if neg else x)
            res.append({"v_pk_fma_f32": lambda: np.float32(np.float64(a[0]) * np.float64(a[1]) + np.float64(a[2])), "v_pk_mul_f32": lambda: np.float32(a[0] * a[1]),
                        "v_pk_add_f32": lambda: np.float32(a[0] + a[1])}[op]())
        v[d], v[d + 1] = res
    for l in [x for x, c in zip(lines, crossed) if c]:
        for _ in range(20):
            v0 = rng.standard_normal(256).astype(np.float32); s0 = rng.standard_normal(104).astype(np.float32)
            v1, v2 = v0.copy(), v0.copy()
            run_packed(l, v1, s0)
            for ins in split_packed_cross_half(l):
                run_plain(ins, v2, s0)
            assert np.array_equal(v1.view(np.uint32), v2.view(np.uint32)), l


def test_real_listings_split_completely(tmp_path):
    """the compiler's listing of a source that is full of crossed packed instructions (the triplane decode): all of them split, the result assembles and links"""
    from ssdnerf_amd.asm_postpass import scan_packed_cross_half, unpack_cross_half
    src = os.path.join(B.CSRC, "raymarching_ops.hip")
    listing = tmp_path / "r.s"
    subprocess.check_call([B._hipcc()] + B.FLAGS + ["-S", "--cuda-device-only", src, "-o", str(listing)], stderr=subprocess.DEVNULL)
    text = listing.read_text()
    before = scan_packed_cross_half(text)
    assert sum(before.values()) > 5
    out, st = unpack_cross_half(text)
    assert st["packed_cross_half_split"] == sum(before.values()) and scan_packed_cross_half(out) == {}
    fixed = tmp_path / "fixed.s"
    fixed.write_text(out)
    B.assemble_and_link(str(fixed), str(tmp_path / "f.o"), str(tmp_path / "f.out"))
    dis = subprocess.run([os.path.join(B.LLVM_BIN, "llvm-objdump"), "-d", str(tmp_path / "f.out")], check=True, capture_output=True, text=True).stdout
    assert scan_packed_cross_half(dis) == {}


HOLES = """
	.text
helper:                                 ; a device FUNCTION (not an .amdhsa_kernel): its caller may have left anything in flight
	v_mul_f32_e32 v3, v0, v0
	v_sqrt_f32_e32 v7, v0
	v_pk_mul_f32 v[8:9], v[7], v[7]
	v_exp_f32_e32 v12, v0
	s_setpc_b64 s[30:31]
_Z1kPf:
	v_mul_f32_e32 v30, v0, v0
	v_rcp_f32_e32 v5, v0
	global_load_lds_dwordx4 v[5:6], off
	v_mul_f32_e32 v6, v5, v0
	v_exp_f32_e32 v10, v0
	s_cbranch_scc1 .LBB0_1
	s_nop 0
.LBB0_1:
	s_cbranch_scc0 .LBB0_2
	s_nop 0
.LBB0_2:
	v_mul_f32_e32 v11, v10, v0
	v_log_f32_e32 v20, v0
	s_swappc_b64 s[30:31], s[4:5]
	v_mul_f32_e32 v21, v12, v0
	s_endpgm
	.amdhsa_kernel _Z1kPf
	.end_amdhsa_kernel
"""


def test_post_pass_handles_the_r03_advisor_holes():
    """(1) function entries / returns from calls leave every VGPR pending, a KERNEL entry none; (2) pending sets reach a block through a short
    intermediate block (fixpoint over labels); (3) `v[7]`; (4) the first operand of an LDS-DMA load is read, not written."""
    out, st = pad_trans_use(HOLES, 4)
    lines = [l.strip() for l in out.split("\n")]
    assert lines[lines.index("v_mul_f32_e32 v3, v0, v0") - 1] == "s_nop 3"                  # function entry: v0 may be a fresh transcendental result
    assert lines[lines.index("v_mul_f32_e32 v30, v0, v0") - 1] == "_Z1kPf:"                 # kernel entry: nothing in flight
    assert lines[lines.index("v_pk_mul_f32 v[8:9], v[7], v[7]") - 1] == "s_nop 3"           # v[7] is v7
    assert lines[lines.index("v_mul_f32_e32 v6, v5, v0") - 1] == "s_nop 2"                  # the LDS-DMA load did not overwrite v5 (it is one slot)
    # exp v10 -> branch (1 slot) -> .LBB0_1 -> branch (1 slot) -> .LBB0_2 -> reader: 2 slots on the all-taken path
    assert lines[lines.index("v_mul_f32_e32 v11, v10, v0") - 1] == "s_nop 1"
    assert lines[lines.index("v_mul_f32_e32 v21, v12, v0") - 1] == "s_nop 3"                # after the call: v12 may come from the callee's v_exp
    assert closest_trans_use(out) >= 4
    again, st2 = pad_trans_use(out, 4)
    assert again == out


def test_linked_code_object_is_verified_independently(tmp_path):
    """``verify_code_object`` disassembles the LINKED device code object and re-checks the rule with its own scanner: the compiler's own code
    object of a library source violates it (one wait state), the post-passed one passes, and both see the same transcendental count."""
    from ssdnerf_amd.asm_postpass import verify_code_object
    import pytest
    src = os.path.join(B.CSRC, "raygen.hip")
    raw_s, fixed_s = tmp_path / "raw.s", tmp_path / "fixed.s"
    subprocess.check_call([B._hipcc()] + B.FLAGS + ["-S", "--cuda-device-only", src, "-o", str(raw_s)], stderr=subprocess.DEVNULL)
    fixed_s.write_text(pad_trans_use(raw_s.read_text(), WS)[0])
    outs = {}
    for name, path in (("raw", raw_s), ("fixed", fixed_s)):
        obj, out = tmp_path / f"{name}.o", tmp_path / f"{name}.out"
        B.assemble_and_link(str(path), str(obj), str(out))
        outs[name] = str(out)
    with pytest.raises(RuntimeError, match="issue slots later"):
        verify_code_object(outs["raw"], WS)
    rep = verify_code_object(outs["fixed"], WS)
    assert rep["trans_instructions"] > 0 and (rep["closest_pair"] is None or rep["closest_pair"] >= WS)
