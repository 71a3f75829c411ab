"""CPU: the C-ABI library loads and exports every symbol include/tfhe_hip.h declares; host-only
entry points work; compute entry points fail loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from gpu_common import ROOT, pkg, to_pkg_params


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "tfhe_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tfhe_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    m = pkg()
    lib = m.lib()
    names = declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), f"{name} declared in tfhe_hip.h but not exported"


def test_host_side_test_vectors_match_oracle(oracle):
    m = pkg()
    for p in (oracle.REF_TEST, oracle.CFG2, oracle.CFG5):
        pp = to_pkg_params(p)
        assert np.array_equal(m.construct_identity_test_vector(pp), oracle.construct_identity_test_vector(p))
        for truth, f in ((m.GATE_AND, lambda l, r: l & r), (m.GATE_NAND, lambda l, r: 1 - (l & r)),
                         (m.GATE_OR, lambda l, r: l | r), (m.GATE_XOR, lambda l, r: l ^ r)):
            assert np.array_equal(m.construct_test_vector_boolean(pp, truth),
                                  oracle.construct_test_vector_boolean(p, f))
    with pytest.raises(m.TfheError):
        m.construct_test_from_lut(to_pkg_params(oracle.REF_TEST), [0, 1, 2])  # test_vector.rs:41


def test_params_validation_matches_oracle(oracle):
    m = pkg()
    cases = [oracle.REF_TEST, oracle.CFG1, oracle.CFG2, oracle.CFG5,
             oracle.Params(1, 10, 8, oracle.Decomposer(7, 5)), oracle.Params(1, 10, 8, oracle.Decomposer(8, 5)),
             oracle.Params(1, 10, 8, oracle.Decomposer(4, 6), log_p=11)]
    for p in cases:
        assert (m.params_validate(to_pkg_params(p)) == 0) == (oracle.validate(p) == 0)


def test_no_cpu_fallback():
    """Without a GPU the context cannot be created; with one, this test is vacuous."""
    import torch
    m = pkg()
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(m.TfheError) as e:
        m.Context(m.TfheParams(1, 10, 630, m.DecomposerParams(7, 3)))
    assert e.value.status == 6  # TFHE_ERR_NO_DEVICE


def test_pool_has_no_cpu_fallback_and_checks_its_arguments():
    """The multi-GPU pool of the C ABI (tfhe_pool_*): refuses bad arguments with a status, fails with
    TFHE_ERR_NO_DEVICE when no GPU is present (vacuous on a GPU box), and NULL handles are harmless."""
    import torch
    m = pkg()
    lib = m.lib()
    lib.tfhe_pool_size.restype = C.c_size_t
    lib.tfhe_pool_last_error.restype = C.c_char_p
    params = m.TfheParams(1, 10, 630, m.DecomposerParams(7, 3))._c()
    out = C.c_void_p()
    one = (C.c_int * 1)(0)
    assert lib.tfhe_pool_create(None, one, C.c_size_t(1), C.c_int(0), C.byref(out)) == m.TFHE_ERR_INVALID_ARGUMENT
    assert lib.tfhe_pool_create(C.byref(params), one, C.c_size_t(0), C.c_int(0), C.byref(out)) == m.TFHE_ERR_INVALID_ARGUMENT
    assert lib.tfhe_pool_create(C.byref(params), None, C.c_size_t(1), C.c_int(0), C.byref(out)) == m.TFHE_ERR_INVALID_ARGUMENT
    assert lib.tfhe_pool_size(None) == 0 and lib.tfhe_pool_last_error(None) == b""
    assert lib.tfhe_pool_synchronize(None) == m.TFHE_ERR_INVALID_ARGUMENT
    lib.tfhe_pool_destroy(None)
    bad = m.TfheParams(1, 10, 630, m.DecomposerParams(7, 5))._c()   # levels > floor(32/7): the reference would not terminate
    assert lib.tfhe_pool_create(C.byref(bad), one, C.c_size_t(1), C.c_int(0), C.byref(out)) == m.TFHE_ERR_INVALID_PARAMS
    if not torch.cuda.is_available():
        with pytest.raises(m.TfheError) as e:
            m.Pool(m.TfheParams(1, 10, 630, m.DecomposerParams(7, 3)), [0, 1])
        assert e.value.status == m.TFHE_ERR_NO_DEVICE


def test_product_does_not_import_the_oracle():
    """The shipped package must not reference oracle/ (only tests, smoke and the bench baseline may)."""
    pkg_dir = os.path.join(ROOT, "tfhe-research_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "tfhe_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f


def _version() -> str:
    lib = pkg().lib()
    lib.tfhe_version.restype = C.c_char_p
    return lib.tfhe_version().decode()


def test_shipped_library_is_not_a_dev_build():
    """tfhe_version() of the shipped library carries no build tag: a dev build (the only kind that may compile a WRONG-BITS
    timing probe in, csrc/dev_switches.h), the rounding-margin probe build and subset builds all say so in brackets"""
    v = _version()
    assert v.startswith("tfhe-research_amd ") and "[" not in v and "DEV" not in v and "WRONG" not in v, v


def test_wrong_bits_probes_need_a_dev_build():
    """csrc/dev_switches.h: a TFHE_PROBE_* switch without TFHE_DEV_BUILD does not compile; with it, it does; and the shipped
    recipe (tfhe-research_amd/build.py) refuses to pass either"""
    import subprocess
    from importlib import import_module
    csrc = os.path.join(ROOT, "tfhe-research_amd", "csrc")
    base = ["g++", "-std=c++17", "-fsyntax-only", "-x", "c++", "-I", csrc, os.path.join(csrc, "platform.h")]
    for probe in ("TFHE_PROBE_HOT_KEY", "TFHE_PROBE_NO_EXCHANGE_READS", "TFHE_PROBE_NO_TRANSPOSE", "TFHE_PROBE_NO_TEAM_SYNC"):
        bad = subprocess.run(base + [f"-D{probe}=1"], capture_output=True, text=True)
        assert bad.returncode != 0 and "TFHE_DEV_BUILD" in bad.stderr, probe
        ok = subprocess.run(base + [f"-D{probe}=1", "-DTFHE_DEV_BUILD"], capture_output=True, text=True)
        assert ok.returncode == 0, ok.stderr
    # no other header defines a probe default any more (one place: dev_switches.h)
    for f in os.listdir(csrc):
        if f != "dev_switches.h" and f.endswith((".h", ".hip", ".cpp")):
            assert "#define TFHE_PROBE_" not in open(os.path.join(csrc, f)).read(), f
    pkg()
    builder = import_module("tfhe_research_amd.build")
    for flag in ("-DTFHE_DEV_BUILD", "-DTFHE_PROBE_HOT_KEY=1"):
        with pytest.raises(ValueError):
            builder._compile("/tmp/never.so", [flag], False, "never")
