"""build()'s ISA check (__graft_entry__.scan_disassembly) on synthetic disassembly: it must accept the shape the
library has and reject the two regressions it exists for -- a write-through store separated from its `s_nop`, and a
stage kernel (specialised or catch-all) that spills to scratch."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

GOOD = """
0000000000001000 <_ZN12_GLOBAL__N_112stage_kernelIffLi1ELi0ELb0ELi1ELi1ELi1ELb0ELb0EEEvPKT_>:
	global_load_dwordx4 v[0:3], v[8:9], off nt                 // 000000001000: DC000000
	global_store_dwordx4 v[10:11], v[0:3], off sc0 sc1         // 000000001008: DC000000
	s_nop 1                                                    // 000000001010: BF800001
--
	global_store_dwordx4 v[10:11], v[4:7], off sc0 sc1         // 000000001018: DC000000
	s_nop 1
0000000000002000 <_ZN12_GLOBAL__N_119stage_thresh_kernelIffLin1ELin1ELb1ELi512ELi0EEEvPKT_>:
	global_store_dwordx4 v[10:11], v[0:3], off sc0 sc1
	s_nop 1
0000000000003000 <_ZN12_GLOBAL__N_116add_noise_kernelIfLb1EEEvPKT_>:
	global_store_dwordx4 v[10:11], v[0:3], off
""".strip("\n").splitlines()


def test_accepts_the_expected_shape():
    n_wt, n_kernels, spilled = G.scan_disassembly(GOOD)
    assert (n_wt, n_kernels) == (3, 2)
    assert not spilled


def test_rejects_a_store_without_its_nop():
    first_nop = next(i for i, ln in enumerate(GOOD) if ln.strip().startswith("s_nop"))
    bad = GOOD[:first_nop] + ["\tv_mov_b32 v0, v1"] + GOOD[first_nop + 1:]
    with pytest.raises(AssertionError, match="not by s_nop"):
        G.scan_disassembly(bad)
    with pytest.raises(AssertionError, match="without its s_nop"):      # the store is the kernel's last instruction
        G.scan_disassembly(GOOD[:first_nop] + GOOD[7:])


def test_rejects_scratch_in_a_specialised_kernel():
    bad = GOOD[:2] + ["\tscratch_load_dword v1, off, s32"] + GOOD[2:]
    with pytest.raises(AssertionError, match="spills to scratch"):
        G.scan_disassembly(bad)


def test_rejects_scratch_in_the_catch_all_thresholding_kernel_too():
    """round 4: no kernel of the library spills any more, the run-time dispatched thresholding kernel included"""
    i = next(i for i, ln in enumerate(GOOD) if "stage_thresh_kernel" in ln)
    bad = GOOD[:i + 1] + ["\tscratch_store_dword off, v40, s32"] + GOOD[i + 1:]
    with pytest.raises(AssertionError, match="spills to scratch"):
        G.scan_disassembly(bad)


def test_rejects_a_mixed_precision_fused_instruction():
    """round 6: v_fma_mixlo_f16 rounds an fp32 product and its fp16 conversion once -- the reference's half arithmetic is an
    fp32 operation, then a conversion (two roundings); the compiler folded the pair in the classifier-free blend of fp16
    networks until from_f32<__half> hid the fp32 value from it"""
    bad = GOOD[:2] + ["\tv_fma_mixlo_f16 v3, v5, s8, 0 op_sel_hi:[1,0,0]"] + GOOD[2:]
    with pytest.raises(AssertionError, match="rounds an fp32 operation"):
        G.scan_disassembly(bad)


NOTES = """
    .group_segment_fixed_size: 0
    .kernarg_segment_size: 240
    .name:           _ZN12_GLOBAL__N_112stage_kernelIffLi1ELi0ELb0ELi1ELi1ELi1ELb0ELb0EEEvPKT_S3_
    .private_segment_fixed_size: 0
    .group_segment_fixed_size: 3072
    .name:           _ZN12_GLOBAL__N_112stage_kernelIffLi3ELi2ELb1ELi100ELi1ELi5ELb0ELb1EEEvPKT_S3_
    .group_segment_fixed_size: 32
    .name:           _ZN12_GLOBAL__N_122adaptive_error_kernel2IfLb1EEEvPKT_
"""


def test_static_lds_rule():
    assert G.scan_kernel_notes(NOTES) == 2
    with pytest.raises(AssertionError, match="static LDS"):     # a non-DYN kernel that owns LDS: the KParams trap
        G.scan_kernel_notes(NOTES.replace("ELb0ELb1EEEv", "ELb0ELb0EEEv"))
    with pytest.raises(AssertionError, match="static LDS"):     # a DYN kernel of another form
        G.scan_kernel_notes(NOTES.replace("IffLi3ELi2E", "IffLi1ELi2E"))
