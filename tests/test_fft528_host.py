"""CPU: the register-resident factors of the 528-point transform (rlao_amd/csrc/fft528.hpp) compiled for the host and checked
against the O(n^2) definition in float64 -- the 24- and 22-point prime-factor transforms, both directions, and the two-factor
528-point chain with the lane <-> index maps the Pyramid kernels use (tests/native/fft528_host.hip)."""
import os
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_fft528_factors_on_the_host(tmp_path):
    exe = tmp_path / "fft528_host"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O2", "-o", str(exe), os.path.join(REPO, "tests", "native", "fft528_host.hip")],
                   check=True, cwd=str(tmp_path), capture_output=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    print(out.stdout)
    assert out.returncode == 0 and "PASS" in out.stdout, out.stdout + out.stderr


def test_lds_layouts_of_the_528_passes_are_conflict_free():
    """The exchange layouts of pyr528_kernels.hip under the bank rules of MI355X_MICROARCH.md (scripts/lds_banks_528.py counts
    LDS cycles per wave instruction): the column pass (16 columns per workgroup, as shipped) is conflict-free in both directions."""
    out = subprocess.run(["python3", os.path.join(REPO, "scripts", "lds_banks_528.py")], capture_output=True, text=True, check=True).stdout
    rows = {l.split()[0] + " " + l.split()[1]: [int(x) for x in l.split()[2:]] for l in out.splitlines() if l.startswith("P2/16")}
    assert set(rows) == {"P2/16 fwd", "P2/16 inv"}
    for name, (w, w_ideal, r, r_ideal) in rows.items():
        assert w == w_ideal and r == r_ideal, (name, w, w_ideal, r, r_ideal)
