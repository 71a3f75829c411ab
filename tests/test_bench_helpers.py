"""CPU: the parts of bench.py that do not need a GPU -- the CPU legs (baseline + verification of the timed rows) on a
small parameter set with the 'GPU rows' played by the oracle itself (and by a corrupted copy: the run must fail), the
row selection, the usable-core count, and the kernel-source hash that gates the committed PMC traffic record."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_spaced_rows_and_usable_cores():
    assert bench.spaced_rows(4096, 3) == [0, 2048, 4095] or bench.spaced_rows(4096, 3) == [0, 2047, 4095]
    assert bench.spaced_rows(5, 16) == [0, 1, 2, 3, 4]
    assert bench.spaced_rows(1, 4) == [0] and bench.spaced_rows(10, 1) == [0]
    rows = bench.spaced_rows(4096, 16)
    assert rows[0] == 0 and rows[-1] == 4095 and len(rows) == 16 and rows == sorted(set(rows))
    c = bench.usable_cores()
    assert 1 <= c["usable"] <= c["affinity"] <= c["logical"]


def test_kernel_source_hash_ignores_host_sources_comments_and_stray_files(tmp_path, monkeypatch):
    h = bench.kernel_source_hash()
    assert len(h) == 16 and h == bench.kernel_source_hash()
    csrc = os.path.join(ROOT, "tfhe-research_amd", "csrc")
    stray = os.path.join(csrc, "kernels.hip.orig")
    strayd = os.path.join(csrc, "editor_backup_dir")
    try:
        open(stray, "w").write("junk")
        os.mkdir(strayd)
        assert bench.kernel_source_hash() == h          # an editor backup or a directory changes nothing
    finally:
        os.remove(stray)
        os.rmdir(strayd)
    # host-only sources are not part of a kernel measurement's identity
    import hashlib
    import re
    names = ["capi.cpp", "pool.cpp", "context.h"]
    seen = []
    real_open = open

    def spy(path, *a, **k):
        seen.append(os.path.basename(str(path)))
        return real_open(path, *a, **k)
    monkeypatch.setattr("builtins.open", spy)
    bench.kernel_source_hash()
    monkeypatch.undo()
    assert "kernels.hip" in seen and "pbs_wave.h" in seen and not any(n in seen for n in names)
    del hashlib, re


def test_cpu_legs_verify_the_timed_rows_and_fail_on_a_wrong_word(oracle, monkeypatch):
    """the reference's cfg(test) shape (n = 4) stands in for cfg2: rows 'written by the GPU' = the oracle's own outputs ->
    verified; one flipped word -> bit_exact false and the mismatch located"""
    p = oracle.REF_TEST
    monkeypatch.setitem(bench.WORKLOADS, "tiny", (p.k, p.glwe_poly_degree, p.n, (p.pbs.log_base, p.pbs.levels),
                                                  (p.ks.log_base, p.ks.levels), p.log_p, 8))
    lwe, bsk, ksk, tv = oracle.synthetic_inputs(p, 8, cfg_index=7)
    oracle.set_poly_mul_mode(1)
    gpu = np.stack([oracle.bootstrap(p, lwe[b], bsk, ksk, tv) for b in range(8)])
    host = {"bsk": bsk, "ksk": ksk, "tv": tv, "rows": list(range(8)), "single_rows": [0, 3, 7], "lwe_rows": lwe, "gpu_rows": gpu.copy()}
    base, ver = bench.cpu_legs("tiny", 30.0, dict(host))
    assert ver["bit_exact"] and ver["rows"] >= 3 and base["cores"] == 1 and base["kind"] == "port" and base["value"] > 0
    if bench.usable_cores()["usable"] > 1:
        assert base["cores_all"]["cores"] == min(bench.usable_cores()["usable"], 8)
    bad = dict(host)
    bad["gpu_rows"] = gpu.copy()
    bad["gpu_rows"][3, 2] ^= 1
    _, ver = bench.cpu_legs("tiny", 30.0, bad)
    assert not ver["bit_exact"] and any(m.get("row") == 3 for m in ver["mismatches"])
    # the aligned leg is checked with the oracle's aligned decomposer
    with oracle.decomposer_aligned(True):
        al = np.stack([oracle.bootstrap(p, lwe[b], bsk, ksk, tv) for b in range(8)])
    good = dict(host)
    good["aligned_gpu_rows"] = al
    _, ver = bench.cpu_legs("tiny", 30.0, good)
    assert ver["bit_exact"] and ver["aligned_decomposer"]["bit_exact"] and ver["aligned_decomposer"]["rows"] >= 3
    good["aligned_gpu_rows"] = al.copy()
    good["aligned_gpu_rows"][0, 0] ^= 1
    _, ver = bench.cpu_legs("tiny", 30.0, good)
    assert not ver["bit_exact"] and not ver["aligned_decomposer"]["bit_exact"]
    oracle.set_poly_mul_mode(1)
