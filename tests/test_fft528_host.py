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
