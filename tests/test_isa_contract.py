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
