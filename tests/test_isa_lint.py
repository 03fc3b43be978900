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
	ds_read2_b32 v[2:3], v10 offset1:32
	ds_read_b32 v4, v10 offset:256
	v_mfma_f32_16x16x4_f32 v[20:23], v30, v31, v[20:23]
	s_waitcnt lgkmcnt(1)
	v_add_f32_e32 v5, v2, v3
	s_waitcnt lgkmcnt(0)
	v_add_f32_e32 v5, v5, v4
	s_endpgm
"""

BAD = """
_ZN4ycnr3badEv: ; @bad
	ds_read2_b32 v[2:3], v10 offset1:32
	ds_read_b32 v4, v10 offset:256
	v_mov_b32_e32 v8, v3
	s_waitcnt lgkmcnt(1)
	v_add_f32_e32 v5, v2, v4
	s_waitcnt lgkmcnt(0)
	s_endpgm
"""


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
                    "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", str(out),
                    os.path.join(CSRC, "ycnr_als.hip")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert lint.lint(str(out), []) == 0
