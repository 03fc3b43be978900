"""devtest/isa_lint.py: the Gramian kernels wait for their inline-asm LDS reads by hand, so the
compiler must not touch a read's destination before the wait that retires it.  The tool is
checked on two hand-written snippets, then run over the device assembly of the library."""
import importlib.util
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "you-can-not-recommend_amd", "csrc")


def load_lint():
    spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(CSRC, "devtest", "isa_lint.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


GOOD = """
_ZN4ycnr4goodEv: ; @good
	;;#ASMSTART
	ds_read2_b32 v[2:3], v10 offset1:32
	ds_read_b32 v4, v10 offset:256
	;;#ASMEND
	v_mfma_f32_16x16x4_f32 v[20:23], v30, v31, v[20:23]
	s_waitcnt lgkmcnt(1)
	v_add_f32_e32 v5, v2, v3
	s_waitcnt lgkmcnt(0)
	v_add_f32_e32 v5, v5, v4
	s_endpgm
"""

BAD = """
_ZN4ycnr3badEv: ; @bad
	;;#ASMSTART
	ds_read2_b32 v[2:3], v10 offset1:32
	ds_read_b32 v4, v10 offset:256
	;;#ASMEND
	v_mov_b32_e32 v8, v3
	s_waitcnt lgkmcnt(1)
	v_add_f32_e32 v5, v2, v4
	s_waitcnt lgkmcnt(0)
	s_endpgm
"""


DPP_GOOD = """
_ZN4ycnr7dppgoodEv: ; @dppgood
	v_mul_f32 v6, v6, v3
	s_nop 1
	v_fmac_f32_dpp v4, -v6, v2 row_newbcast:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
	v_mov_b32_e32 v9, v4
	v_add_f32_e32 v1, v1, v1
	v_add_f32_e32 v1, v1, v1
	v_add_f32_dpp v9, v9, v9 row_ror:8 row_mask:0xf bank_mask:0xf
	s_endpgm
"""

DPP_BAD = """
_ZN4ycnr6dppbadEv: ; @dppbad
	s_cbranch_scc1 .LBB0_2
	v_accvgpr_read_b32 v6, a3
	s_branch .LBB0_3
.LBB0_2:
	v_mov_b32_e32 v6, v7
	s_nop 0
.LBB0_3:
	v_fmac_f32_dpp v4, -v6, v2 row_newbcast:1 row_mask:0xf bank_mask:0xf bound_ctrl:1
	s_endpgm
"""


PK_BAD = """
_ZN4ycnr24als_gram_slab_x6d_kernelILi4ELb0EEEvNS_8StepArgsIfEE: ; @x6d
	v_pk_fma_f32 v[120:121], v[122:123], v[30:31], v[120:121] op_sel_hi:[1,0,1]
	s_endpgm
_ZN4ycnr21als_dual_solve_kernelILi2ELb1EEEvNS_8StepArgsIfEE: ; @no kernel may pack float32 multiplies (round 5)
	v_pk_mul_f32 v[2:3], v[4:5], v[6:7]
	s_endpgm
_ZN4ycnr15als_rmse_kernelIfEEvNS_8RmseArgsIT_EE: ; @packed moves and packed adds pass
	v_pk_mov_b32 v[2:3], v[4:5], v[6:7]
	v_pk_add_f32 v[2:3], v[4:5], v[6:7]
	v_fma_f32 v2, v3, v4, v2
	s_endpgm
"""


def test_packed_float32_lint_names_every_kernel_that_packs(tmp_path, capsys):
    lint = load_lint()
    f = tmp_path / "pk.s"
    f.write_text(PK_BAD)
    assert lint.lint_packed_fma(str(f)) == 2
    out = capsys.readouterr().out
    assert "x6d" in out and "als_dual_solve_kernelILi2" in out and "rmse" not in out.replace("kernels checked", "")


SCRATCH = """
	.amdhsa_kernel _ZN4ycnr21als_dual_solve_kernelILi7ELb1EEEvNS_8StepArgsIfEE
		.amdhsa_private_segment_fixed_size 0
		.amdhsa_next_free_vgpr 256
	.end_amdhsa_kernel
	.amdhsa_kernel _ZN4ycnr21als_dual_solve_kernelILi11ELb1EEEvNS_8StepArgsIfEE
		.amdhsa_private_segment_fixed_size 112
		.amdhsa_next_free_vgpr 512
	.end_amdhsa_kernel
	.amdhsa_kernel _ZN4ycnr21als_dual_solve_kernelILi4ELb1EEEvNS_8StepArgsIfEE
		.amdhsa_private_segment_fixed_size 0
		.amdhsa_next_free_vgpr 160
	.end_amdhsa_kernel
	.amdhsa_kernel _ZN4ycnr23als_reduce_solve_kernelIdLi8ELb0ELb0ELb0EEEvNS_8StepArgsIT_EE
		.amdhsa_private_segment_fixed_size 320
		.amdhsa_next_free_vgpr 512
	.end_amdhsa_kernel
	.amdhsa_kernel _ZN4ycnr24als_gram_slab_x6d_kernelILi7ELb1ELb1EEEvNS_8StepArgsIfEE
		.amdhsa_private_segment_fixed_size 16
		.amdhsa_next_free_vgpr 248
	.end_amdhsa_kernel
_ZN4ycnr24als_gram_slab_x6d_kernelILi7ELb1ELb1EEEvNS_8StepArgsIfEE: ; @counted waits + a spill
	buffer_load_dwordx4 v212, s[8:11], s83 offen lds
	scratch_store_dword off, v146, off offset:16 ; 4-byte Folded Spill
	;;#ASMSTART
	s_waitcnt vmcnt(14)
	;;#ASMEND
	scratch_load_dword v146, off, off offset:16 ; 4-byte Folded Reload
	s_endpgm
_ZN4ycnr21als_dual_solve_kernelILi11ELb1EEEvNS_8StepArgsIfEE: ; @scratch, but every wait is the compiler's own
	scratch_store_dword off, v1, off
	s_waitcnt vmcnt(0)
	s_endpgm
_ZN4ycnr21als_dual_solve_kernelILi7ELb1EEEvNS_8StepArgsIfEE: ; @two waves per SIMD AND a packed multiply-add
	v_pk_fma_f32 v[150:151], v[68:69], v[146:147], v[148:149] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]
	s_endpgm
"""


def test_scratch_lints_name_counted_waits_and_the_two_wave_dual_class(tmp_path, capsys):
    """(1) A kernel that counts its vector-memory waits by hand (inline-asm vmcnt, LDS-DMA) must have NO scratch: a spill is a
    vector-memory operation its count does not know.  (2) The dual class of 7 blocks built for two waves per SIMD solved rows
    wrong at C5 scale exactly when hipcc had paired two blocks' right-hand-side updates into v_pk_fma_f32
    (devtest/dual7/README.md): a class of 7+ blocks that fits two waves per SIMD must contain no packed float32 instruction.
    Other kernels below the general limit pass, whatever their scratch."""
    lint = load_lint()
    f = tmp_path / "scratch.s"
    f.write_text(SCRATCH)
    assert lint.lint_scratch(str(f)) == 0
    assert lint.lint_counted_waits_have_no_scratch(str(f)) == 1
    assert lint.lint_dual_occupancy(str(f)) == 1
    out = capsys.readouterr().out
    fails = [l for l in out.splitlines() if l.startswith("FAIL")]
    assert len(fails) == 2 and "x6d" in fails[0] and "dual class of 7 blocks" in fails[1]


ASM_LOAD_BAD = """
_ZN4ycnr7asmloadEv: ; @a value loaded by inline asm, carried around the loop and copied at the back edge while in flight
	s_branch .LBB0_2
.LBB0_1:
	buffer_load_dwordx4 v205, s[8:11], s80 offen lds
	s_cbranch_scc0 .LBB0_3
.LBB0_2:
	v_mov_b32_e32 v206, v194
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	;;#ASMSTART
	buffer_load_dword v194, v203, s[16:19], 0 offen
	;;#ASMEND
	buffer_load_dwordx4 v205, s[8:11], s79 offen lds
	s_branch .LBB0_1
.LBB0_3:
	s_endpgm
"""


def test_asm_vector_load_lint_follows_the_back_edge(tmp_path, capsys):
    """GramX6P loads the ids and ratings of a step into registers from inline asm and retires them with hand-counted waits.  Its
    first build let the compiler copy the loop-carried rating at the loop header, one operation before the count allowed: one
    row in 10^5 was wrong on the GPU.  The lint must see that copy (reached only around the loop's back edge), and pass the
    same loop with a wait that retires the load in front of the back edge."""
    lint = load_lint()
    bad, good = tmp_path / "bad.s", tmp_path / "good.s"
    bad.write_text(ASM_LOAD_BAD)
    good.write_text(ASM_LOAD_BAD.replace("\tbuffer_load_dwordx4 v205, s[8:11], s79 offen lds\n\ts_branch .LBB0_1",
                                         "\tbuffer_load_dwordx4 v205, s[8:11], s79 offen lds\n\ts_waitcnt vmcnt(1)\n\ts_branch .LBB0_1"))
    assert lint.lint_asm_vector_loads(str(bad), []) >= 1
    assert "v_mov_b32_e32 v206, v194" in capsys.readouterr().out
    assert lint.lint_asm_vector_loads(str(good), []) == 0


def test_dpp_lint_counts_wait_states_over_every_path(tmp_path, capsys):
    lint = load_lint()
    good, bad = tmp_path / "good.s", tmp_path / "bad.s"
    good.write_text(DPP_GOOD)
    bad.write_text(DPP_BAD)
    assert lint.lint_dpp(str(good), []) == 0
    assert lint.lint_dpp(str(bad), []) == 2  # the AGPR reload through the branch, the copy one wait state before the label
    assert "FAIL" in capsys.readouterr().out


def test_lint_accepts_waited_uses_and_flags_premature_ones(tmp_path, capsys):
    lint = load_lint()
    good, bad = tmp_path / "good.s", tmp_path / "bad.s"
    good.write_text(GOOD)
    bad.write_text(BAD)
    assert lint.lint(str(good), []) == 0
    assert lint.lint(str(bad), []) == 2  # the copy of v3 before any wait, and v4 behind lgkmcnt(1)
    assert "FAIL" in capsys.readouterr().out


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_device_assembly_has_no_premature_lds_uses(tmp_path):
    lint = load_lint()
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = tmp_path / "dev.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-ffp-contract=on",
                    "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", str(out),
                    os.path.join(CSRC, "ycnr_als.hip")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert lint.lint(str(out), []) == 0
    # vector loads issued from inline asm (GramX6P's ids and ratings): no use or copy before their counted wait, back edges included
    assert lint.lint_asm_vector_loads(str(out), []) == 0
    # the inline-asm v_fmac_f32_dpp pivot updates: no compiler-inserted copy may sit within two wait states
    # in front of one
    assert lint.lint_dpp(str(out), []) == 0
    # no instantiation may spill hundreds of bytes per lane to scratch (one did, unnoticed, at twice the run time)
    assert lint.lint_scratch(str(out)) == 0
    # a kernel with hand-counted vector-memory waits (GramX6D, GramX6P) carries no scratch at all; the dual classes of 7+
    # blocks are built for one wave per SIMD (devtest/dual7/README.md)
    assert lint.lint_counted_waits_have_no_scratch(str(out)) == 0
    assert lint.lint_dual_occupancy(str(out)) == 0
    # the fence of the stale-b and the dual7 hazard: no packed float32 arithmetic in ANY kernel (devtest/pkfma/README.md;
    # the library is built with -fno-slp-vectorize)
    assert lint.lint_packed_fma(str(out)) == 0
