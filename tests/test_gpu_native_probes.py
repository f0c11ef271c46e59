"""Device-level unit test of the 32 x 32 factor-and-invert building block of the big-front chain:
scripts/diag32_probe.hip includes the kernel header, runs diag32_factor_invert on SPD blocks of several
sizes in both precisions and compares L and L^-1 with a host Cholesky (exit code 0 = all within tolerance)."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_diag32_factor_invert_against_host_cholesky(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "diag32_probe"
    src = os.path.join(ROOT, "scripts", "diag32_probe.hip")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", src, "-o", str(exe)],
                   check=True, timeout=600)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout and "FAILED" not in r.stdout
