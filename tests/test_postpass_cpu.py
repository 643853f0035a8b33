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
    assert report["wait_states"] == B.TRANS_USE_WAIT_STATES >= 2
    assert set(report["sources"]) == set(B.SOURCES)
    shade = report["sources"]["shade_mfma.hip"]
    assert shade["trans_instructions"] > 1000 and shade["closest_pair_after"] >= B.TRANS_USE_WAIT_STATES
