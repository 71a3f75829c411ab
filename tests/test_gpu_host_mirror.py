"""Builds and runs the C++ host-side mirror test (tests/cpp/test_host_mirror.cpp) on the GPU."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "tfhe-research_amd")
ORACLE = os.path.join(ROOT, "oracle")


def build_binary():
    exe = os.path.join(HERE, "cpp", "test_host_mirror")
    subprocess.run(["make", "-C", ORACLE, "-s", "all"], check=True)
    cmd = ["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", ORACLE,
           os.path.join(HERE, "cpp", "test_host_mirror.cpp"), "-o", exe,
           "-L", PKG, "-ltfhe_hip", "-L", os.path.join(ORACLE, "_build"), "-ltfhe_oracle",
           f"-Wl,-rpath,{PKG}", f"-Wl,-rpath,{os.path.join(ORACLE, '_build')}"]
    subprocess.run(cmd, check=True)
    return exe


def test_host_mirror_compiles_against_the_abi():
    """CPU: the header-only mirror compiles and links against the C ABI library."""
    assert os.path.exists(build_binary())


@pytest.mark.gpu
def test_host_mirror_on_gpu():
    out = subprocess.run([build_binary()], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host mirror OK" in out.stdout
