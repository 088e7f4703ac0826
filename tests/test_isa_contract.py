"""CPU (hipcc cross-compiles): the compiled gfx950 code of the z-slide convolution holds the hand-counted
`s_waitcnt vmcnt(N)` contract of its helper waves (tools/isa_vmcnt_check.py; the failure of commit 2271921)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_vmcnt_check as chk  # noqa: E402

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SRC = os.path.join(ROOT, "calodiffusion_amd", "csrc", "kernels_conv_zs.hip")


@pytest.fixture(scope="module")
def zs_asm(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "zs.s"
    from calodiffusion_amd.build import FILE_FLAGS, FLAGS
    flags = [f for f in FLAGS if f not in ("-fPIC",)] + FILE_FLAGS.get("kernels_conv_zs.hip", [])
    subprocess.run([HIPCC, *flags, "-S", "--cuda-device-only", "-o", str(out), SRC], check=True, capture_output=True)
    return str(out)


def test_zslide_vmcnt_contract_holds(zs_asm):
    chk.check_file(zs_asm, "conv_zslide", verbose=False, require=2)


def test_checker_catches_a_merged_store(zs_asm, tmp_path):
    """Remove one of the eight row stores of the steady-state interval (what store merging did): the checker must object."""
    s = open(zs_asm).read()
    waits = [m.start() for m in re.finditer(r"vmcnt\(8\) ; zs_landed", s)]
    assert len(waits) >= 2
    j = s.index("zs_row_store", waits[1])
    bad = tmp_path / "bad.s"
    bad.write_text(s[:s.rfind("\n", 0, j)] + s[s.index("\n", j):])
    with pytest.raises(SystemExit, match="vmcnt contract violated"):
        chk.check_file(str(bad), "conv_zslide", verbose=False)
    # ... and an unmarked vector-memory instruction inside the window
    k = s.index("zs_row_store", waits[1])
    line_start = s.rfind("\n", 0, k) + 1
    bad.write_text(s[:line_start] + "\tglobal_load_dword v0, v[2:3], off\n" + s[line_start:])
    with pytest.raises(SystemExit, match="unmarked vector-memory"):
        chk.check_file(str(bad), "conv_zslide", verbose=False)


def test_lindiv_op_is_not_contracted(tmp_path):
    """CD_SOP_LINDIV promises the rounding of a chain of torch elementwise ops: in the compiled gfx950 code every term of
    lincomb_div_kernel is a v_mul_f32 followed by a v_add_f32 -- the only fused instructions are those of the IEEE division
    sequence (v_div_scale ... v_div_fixup)."""
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("hipcc not available")
    from calodiffusion_amd.build import FLAGS
    out = tmp_path / "misc.s"
    src = os.path.join(ROOT, "calodiffusion_amd", "csrc", "kernels_misc.hip")
    subprocess.run([HIPCC, *[f for f in FLAGS if f != "-fPIC"], "-S", "--cuda-device-only", "-o", str(out), src], check=True,
                   capture_output=True)
    s = open(out).read()
    m = re.search(r"^(_ZN2cd18lincomb_div_kernel\w+):[^\n]*\n(.*?)s_endpgm", s, re.S | re.M)
    assert m, "lincomb_div_kernel not found in the assembly"
    ops = re.findall(r"^\s+(v_\w+)", m.group(2), re.M)
    assert ops.count("v_mul_f32_e32") >= 6 and ops.count("v_add_f32_e32") == 5, ops
    fused = [o for o in ops if o.startswith(("v_fma", "v_fmac", "v_mac", "v_pk_fma"))]
    assert len(fused) <= 5 and "v_div_fixup_f32" in ops, fused  # the division's Newton steps only
