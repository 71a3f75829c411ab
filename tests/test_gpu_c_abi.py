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


GOLDEN_SETS = ["ref_test", "misaligned", "n1024_full_word"]


@pytest.mark.gpu
def test_c_consumer_on_gpu():
    """results, not only linkage, with no Python on the caller's side: the C program reads the committed fixtures
    (tests/golden/<set>/{bsk,ksk,lwe_in,tv,lwe_out}.tfhe) through tfhe_file_read and memcmp's what tfhe_bootstrap_batch and
    tfhe_pool_bootstrap_batch return with lwe_out"""
    sets = [os.path.join(HERE, "golden", s) for s in GOLDEN_SETS]
    out = subprocess.run([build_binary()] + sets, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "pool of 2 == single context" in out.stdout
    assert out.stdout.count("golden OK") == len(GOLDEN_SETS), out.stdout


@pytest.mark.gpu
def test_library_loaded_on_the_gpu_box_is_the_shipped_build():
    """the library the -m gpu suite runs against identifies itself as the product: no dev / probe / subset tag in
    tfhe_version() (a build with a WRONG-BITS timing probe compiled in says so there, csrc/dev_switches.h)"""
    import ctypes as C
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    assert not os.environ.get("TFHE_HIP_LIB"), "the suite must run against the in-tree product library"
    lib = entry.load_package().lib()
    lib.tfhe_version.restype = C.c_char_p
    v = lib.tfhe_version().decode()
    assert "[" not in v and "DEV" not in v and "WRONG" not in v, v
