"""The build's assembly post-pass (ssdnerf_amd/asm_postpass.py): every transcendental -> use pair gets the required issue slots, the compiler's own
pad is lengthened in place where there is one, nothing else moves, and the pass is idempotent.  Checked on a hand-written listing and on the
compiler's real listing of one library source (hipcc cross-compiles without a GPU)."""
import os
import subprocess

from ssdnerf_amd import build as B
from ssdnerf_amd.asm_postpass import closest_trans_use, pad_trans_use

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
    assert A.SWAP_MFMA_WAIT_STATES == 0 or A.SWAP_MFMA_WAIT_STATES == B.SWAP_MFMA_WAIT_STATES
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
    out, st = pad_trans_use(text, B.TRANS_USE_WAIT_STATES)
    assert st["trans_instructions"] > 0 and st["pairs_closer_than_required"] > 0
    assert closest_trans_use(out) >= B.TRANS_USE_WAIT_STATES
    assert [l for l in out.split("\n") if "s_nop" not in l] == [l for l in text.split("\n") if "s_nop" not in l]


def test_shipped_library_was_built_with_the_post_pass():
    import json
    report = json.load(open(os.path.join(B.LIB_DIR, "postpass_report.json")))
    assert report["wait_states"] == B.TRANS_USE_WAIT_STATES >= 2 and report["toolchain"]["validated"] is True
    assert set(report["sources"]) == set(B.SOURCES)
    shade = report["sources"]["shade_mfma.hip"]
    assert shade["trans_instructions"] > 1000 and shade["closest_pair_after"] >= B.TRANS_USE_WAIT_STATES
    # the same rule re-checked on the linked code object by the independent scanner (asm_postpass.verify_code_object)
    assert shade["code_object_check"]["trans_instructions"] == shade["trans_instructions"] and shade["code_object_check"]["closest_pair"] >= B.TRANS_USE_WAIT_STATES
    # the swap -> matrix-operand rule is on in the shipped build, and the shading kernel is where it applies
    assert report["settings"]["swap_mfma_wait_states"] == B.SWAP_MFMA_WAIT_STATES >= 8 and shade["swap_mfma_pairs_padded"] > 0
    assert shade["code_object_check"]["swap_instructions"] > 100 and shade["code_object_check"]["closest_swap_mfma_pair"] >= B.SWAP_MFMA_WAIT_STATES   # (linked code object)
    assert all("swap_mfma_pairs_padded" not in v for k, v in report["sources"].items() if k != "shade_mfma.hip")


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
    fixed_s.write_text(pad_trans_use(raw_s.read_text(), B.TRANS_USE_WAIT_STATES)[0])
    outs = {}
    for name, path in (("raw", raw_s), ("fixed", fixed_s)):
        obj, out = tmp_path / f"{name}.o", tmp_path / f"{name}.out"
        B.assemble_and_link(str(path), str(obj), str(out))
        outs[name] = str(out)
    with pytest.raises(RuntimeError, match="issue slots later"):
        verify_code_object(outs["raw"], B.TRANS_USE_WAIT_STATES)
    rep = verify_code_object(outs["fixed"], B.TRANS_USE_WAIT_STATES)
    assert rep["trans_instructions"] > 0 and (rep["closest_pair"] is None or rep["closest_pair"] >= B.TRANS_USE_WAIT_STATES)
