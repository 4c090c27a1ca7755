"""Builds and runs the C++ host-mirror tests (include/crgpu.hpp over the C ABI) on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "test_host_mirror")
    pkg = os.path.join(ROOT, "cellranger_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp"), "-L" + pkg, "-lcrgpu",
           "-Wl,-rpath," + pkg, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_cpp_host_mirror_compiles(tmp_path):
    """CPU: the header-only C++ layer compiles and links against libcrgpu.so."""
    from cellranger_amd import build

    build.build()
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_cpp_host_mirror_reference_unit_tests(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all tests passed" in r.stdout
