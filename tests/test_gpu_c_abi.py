"""Builds and runs a plain C99 consumer of the C ABI (tests/c/test_c_abi.c): the header is C, the library links from C,
and on the GPU the pool and the single context agree word for word."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "tfhe-research_amd")


def build_binary():
    exe = os.path.join(HERE, "c", "test_c_abi")
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(HERE, "c", "test_c_abi.c"), "-o", exe, "-L", PKG, "-ltfhe_hip", f"-Wl,-rpath,{PKG}"], check=True)
    return exe


def test_c_consumer_compiles_links_and_fails_loudly_without_a_gpu():
    """CPU: gcc -std=c99 -Werror against include/tfhe_hip.h; without a device the program reports the link check only"""
    out = subprocess.run([build_binary()], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "c abi OK" in out.stdout


@pytest.mark.gpu
def test_c_consumer_on_gpu():
    out = subprocess.run([build_binary()], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "pool of 2 == single context" in out.stdout
