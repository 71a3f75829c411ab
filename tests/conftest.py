import ctypes as C
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (checker only)."""
    from oracle import oracle as orc
    orc.build()
    orc.set_poly_mul_mode(1)
    return orc


EMU_DIR = os.path.join(ROOT, "tests", "emu")
CSRC = os.path.join(ROOT, "tfhe-research_amd", "csrc")


_sanitizer_build = None


def sanitizer_binary():
    """tests/emu/sanitize (the emulator built with ASan + UBSan): two minutes of g++.  The build is started in the
    background the first time the emulator is asked for, so that it runs beside the emulator tests instead of after
    them; the sanitizer test (tests/test_emu_kernels.py) joins it here.  Returns the path of the up-to-date binary."""
    global _sanitizer_build
    exe = os.path.join(EMU_DIR, "sanitize")
    src = os.path.join(EMU_DIR, "sanitize_main.cpp")
    deps = [src, os.path.join(EMU_DIR, "emu.cpp")] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if _sanitizer_build is None:
        if os.path.exists(exe) and all(os.path.getmtime(d) <= os.path.getmtime(exe) for d in deps):
            return exe
        _sanitizer_build = subprocess.Popen(
            ["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-fsanitize=address,undefined",  # add -g to localise a report
             "-fno-sanitize-recover=all", "-pthread", "-I", CSRC, src, "-o", exe + ".tmp"])
        return None
    if _sanitizer_build.wait() != 0:
        raise RuntimeError("building tests/emu/sanitize failed")
    if os.path.exists(exe + ".tmp"):
        os.replace(exe + ".tmp", exe)
    return exe


@pytest.fixture(scope="session")
def emu():
    """The host SIMT emulator: the device headers (csrc/*.h) compiled by g++ (tests/emu/emu.cpp)."""
    so = os.path.join(EMU_DIR, "libtfhe_emu.so")
    srcs = [os.path.join(EMU_DIR, "emu.cpp")] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    sanitizer_binary()  # starts the sanitizer twin's build in the background if it is stale
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", "-I", CSRC,
                        "-o", so, os.path.join(EMU_DIR, "emu.cpp")], check=True)
    lib = C.CDLL(so)
    for f in ("emu_gl_mul", "emu_gl_add", "emu_gl_sub", "emu_gl_from_i32"):
        getattr(lib, f).restype = C.c_uint64
    lib.emu_gl_lift.restype = C.c_uint32
    lib.emu_fp_p.restype = C.c_double
    lib.emu_fp_from_key_word.restype = C.c_double
    return lib


