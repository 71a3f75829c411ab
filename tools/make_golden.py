#!/usr/bin/env python3
"""Writes the golden fixtures under tests/golden/ from fixed seeds (run once; the output is committed).

    python tools/make_golden.py            # regenerate everything (about a minute on 8 cores)
    python tools/make_golden.py --check    # recompute and compare with what is committed

What the fixtures are for.  The reference (Janmajayamall/tfhe-research) holds no golden vectors
and cannot be built in this image (Rust, no cargo), so bit-level parity of this repository hangs
on the CPU oracle (oracle/tfhe_oracle.c).  The fixtures FREEZE the oracle's outputs:

  * `pytest -m "not gpu"` asserts the oracle of today still reproduces them (oracle drift fails CI),
  * `pytest -m gpu` asserts the HIP path reproduces them (NOT a fresh oracle run),
  * anyone with a Rust toolchain replays them through the reference crate itself
    (rust/reference_patch/golden_replay.rs), which pins the oracle to the
    crate in one `cargo test --release` -- the step this image cannot do.

Sets (every array is one file in the library's on-disk format, include/tfhe_hip.h: 104-byte header
+ little-endian u32 payload in the reference's row-major layout):

  ref_test/    the reference's cfg(test) parameters (lib.rs:77-99: N=512, k=2, n=4, PBS l=6 logB=4,
               KS l=5 logB=4) with REAL keys (oracle keygen, fixed seed): rows 0-3 encrypt the
               messages 0..3, rows 4-7 are edge cases (a~ = 0, b~ rounding to 2N, all words
               0x80000000, uniform words).  Identity test vector.
  n1024_full_word/  N=1024, k=1, n=3, PBS l=4 logB=8 (the whole word decomposed: every CMUX depends on the key)
  misaligned/  N=1024, k=1, n=3, PBS l=3 logB=7 (the BASELINE cfg2 decomposer, log_base does not
               divide 32: decomposer.rs:48-70 counts limbs from bit 0), uniform synthetic words,
               a non-identity LUT.

  per set: params (in every header), bsk, ksk, lwe_in, tv, approximate_lwe (switch_modulus,
  bootstrapping.rs:67-71), acc_init (X^-b~ * tv), acc_after_each (after CMUX 0..n-1), acc_final,
  extracted_lwe (sample_extract), lwe_out (key switch) -- the trace of bootstrapping.rs:58-120.

  full_size_digests.json: SHA-256 of lwe_out / extracted_lwe / acc_final for 8 rows of the full
  4096-row synthetic batch of BASELINE cfg1, cfg2, cfg3, cfg5 (inputs are regenerated from the
  SplitMix64 seed, oracle.synthetic_inputs; only digests are stored).

TEST INFRASTRUCTURE: uses the oracle (allowed for tests/ tooling), never shipped.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
from oracle import oracle as orc  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
FULL_ROWS = [0, 1, 2, 3, 1000, 2047, 4094, 4095]
FULL_BATCH = 4096


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u4").tobytes()).hexdigest()


def small_sets():
    """-> {name: (oracle.Params, dict of arrays)}; deterministic."""
    out = {}
    # ---- the reference's cfg(test) parameter set with real keys
    p = orc.REF_TEST
    rng = orc.Rng(0x676F6C64656E01)  # "golden" 01
    lwe_sk, glwe_sk, bsk, ksk = orc.keygen(p, rng)
    rows = [orc.encrypt_lwe(p, lwe_sk, m, rng) for m in range(4)]
    edge = orc.splitmix64_u32(orc.SYNTH_SEED + 0x601D, 4 * (p.n + 1)).reshape(4, p.n + 1).copy()
    edge[0, 0] = 0                 # a~_0 = 0: the CMUX whose difference is all zero
    edge[1, p.n] = 0xFFFFFFFF      # b~ rounds up to 2N and wraps to 0 (utils.rs:27-31)
    edge[2, :] = 0x80000000
    lwe_in = np.concatenate([np.stack(rows), edge]).astype(np.uint32)
    tv = orc.construct_identity_test_vector(p)
    out["ref_test"] = (p, dict(bsk=bsk, ksk=ksk, lwe_in=lwe_in, tv=tv, lwe_sk=lwe_sk, glwe_sk=glwe_sk))
    # ---- misaligned gadget base (BASELINE cfg2's decomposer at a short LWE key)
    p = orc.Params(1, 10, 3, orc.Decomposer(7, 3))
    lut = [2, 0, 3, 1]
    lwe_in, bsk, ksk, tv = orc.synthetic_inputs(p, 8, cfg_index=0x601E, lut=lut)
    lwe_in = lwe_in.copy()
    lwe_in[0, 0] = 0
    lwe_in[1, p.n] = 0xFFFFFFFF
    lwe_in[2, :] = 0x80000000
    lwe_in[3, :] = 0xF8F8F8F8      # every limb at B-1 with a carry chain (digit == B path)
    out["misaligned"] = (p, dict(bsk=bsk, ksk=ksk, lwe_in=lwe_in, tv=tv))
    # ---- N = 1024 with a decomposer that covers the whole word (l = 4, logB = 8): unlike the set above, whose
    # digits of a trivial accumulator are all zero (bits 28..31 are never decomposed), every CMUX here depends on
    # the key -- the frozen trace that pins the N = 1024 kernels (the complex-FFT backend's home) stage by stage
    p = orc.Params(1, 10, 3, orc.Decomposer(8, 4))
    lut = [1, 3, 0, 2]
    lwe_in, bsk, ksk, tv = orc.synthetic_inputs(p, 8, cfg_index=0x601F, lut=lut)
    lwe_in = lwe_in.copy()
    lwe_in[0, 0] = 0
    lwe_in[1, p.n] = 0xFFFFFFFF
    lwe_in[2, :] = 0x80000000
    lwe_in[3, :] = 0xFFFFFFFF      # every limb at B-1 with a carry chain
    out["n1024_full_word"] = (p, dict(bsk=bsk, ksk=ksk, lwe_in=lwe_in, tv=tv))
    return out


def trace_set(p, arrays):
    """run the oracle's bootstrap with the full trace over every row"""
    orc.set_poly_mul_mode(0)  # the literal Toeplitz product of utils.rs:155-160
    rows = arrays["lwe_in"].shape[0]
    tr = dict(approximate_lwe=[], acc_init=[], acc_after_each=[], acc_final=[], extracted_lwe=[], lwe_out=[])
    for b in range(rows):
        out, t = orc.bootstrap(p, arrays["lwe_in"][b], arrays["bsk"], arrays["ksk"], arrays["tv"],
                               trace=True, trace_each=True)
        tr["lwe_out"].append(out)
        for k in ("approximate_lwe", "acc_init", "acc_after_each", "acc_final", "extracted_lwe"):
            tr[k].append(t[k])
    orc.set_poly_mul_mode(1)
    return {k: np.stack(v).astype(np.uint32) for k, v in tr.items()}


FILE_KINDS = {  # name -> kind of include/tfhe_hip.h
    "bsk": "FILE_BSK", "ksk": "FILE_KSK", "lwe_in": "FILE_LWE", "lwe_out": "FILE_LWE", "extracted_lwe": "FILE_LWE",
    "acc_init": "FILE_GLWE", "acc_final": "FILE_GLWE",
    "acc_after_each": "FILE_WORDS",  # [rows * n][k+1][N] flattened to 4 dims below
    "tv": "FILE_WORDS", "approximate_lwe": "FILE_WORDS", "lwe_sk": "FILE_WORDS", "glwe_sk": "FILE_WORDS",
}


def full_size_digests(workers: int):
    res = {"batch": FULL_BATCH, "rows": FULL_ROWS,
           "inputs": "oracle.synthetic_inputs(params, 4096, cfg_index=<cfg number>, lut=<lut>): SplitMix64 high halves, "
                     "seed 0x7466686500000000 + cfg number, stream order lwe | bsk | ksk",
           "digest": "sha256 of the little-endian u32 words of one row", "configs": {}}
    for name, cfg_index in (("cfg1", 1), ("cfg2", 2), ("cfg3", 3), ("cfg5", 5)):
        p = orc.CONFIGS[name]
        lut = [int(x) for x in np.random.default_rng(5).integers(0, 16, size=16)] if name == "cfg5" else None
        lwe, bsk, ksk, tv = orc.synthetic_inputs(p, FULL_BATCH, cfg_index=cfg_index, lut=lut)

        def one(b):
            out, t = orc.bootstrap(p, lwe[b], bsk, ksk, tv, trace=True)
            return {"row": b, "lwe_in": sha(lwe[b]), "lwe_out": sha(out), "extracted_lwe": sha(t["extracted_lwe"]),
                    "acc_final": sha(t["acc_final"])}
        with ThreadPoolExecutor(workers) as pool:
            rows = list(pool.map(one, FULL_ROWS))
        res["configs"][name] = {
            "params": {"k": p.k, "log_n": p.glwe_poly_degree, "n": p.n, "pbs": [p.pbs.log_base, p.pbs.levels],
                       "ks": [p.ks.log_base, p.ks.levels], "log_p": p.log_p, "padding_bits": p.padding_bits},
            "cfg_index": cfg_index, "lut": lut, "tv": sha(tv), "bsk": sha(bsk), "ksk": sha(ksk), "rows": rows}
        print(f"  {name}: {len(rows)} rows", flush=True)
    return res


def generate(workers: int):
    """-> {relative path: bytes}"""
    import tempfile
    pkg = entry.load_package()
    orc.build()
    files = {}
    manifest = {"generator": "python tools/make_golden.py", "oracle": "oracle/tfhe_oracle.c, literal Toeplitz poly_mul",
                "format": "include/tfhe_hip.h on-disk format (magic TFHEAMD\\1, 104-byte header, LE u32 payload)",
                "sets": {}}
    with tempfile.TemporaryDirectory() as tmp:
        for name, (p, arrays) in small_sets().items():
            arrays = dict(arrays)
            arrays.update(trace_set(p, arrays))
            rows = arrays["lwe_in"].shape[0]
            arrays["acc_after_each"] = arrays["acc_after_each"].reshape(rows * p.n, p.k + 1, p.N)
            params = pkg.TfheParams(p.k, p.glwe_poly_degree, p.n, pkg.DecomposerParams(p.pbs.log_base, p.pbs.levels),
                                    pkg.DecomposerParams(p.ks.log_base, p.ks.levels), log_p=p.log_p,
                                    padding_bits=p.padding_bits)
            entry_m = {}
            for key, arr in sorted(arrays.items()):
                path = os.path.join(tmp, "f.tfhe")
                pkg.save_array(path, getattr(pkg, FILE_KINDS[key]), params, arr)
                with open(path, "rb") as f:
                    blob = f.read()
                rel = f"{name}/{key}.tfhe"
                files[rel] = blob
                entry_m[key] = {"shape": list(arr.shape), "sha256_file": hashlib.sha256(blob).hexdigest(),
                                "sha256_payload": sha(arr)}
            manifest["sets"][name] = {"params": {"k": p.k, "log_n": p.glwe_poly_degree, "n": p.n,
                                                 "pbs": [p.pbs.log_base, p.pbs.levels], "ks": [p.ks.log_base, p.ks.levels],
                                                 "log_p": p.log_p, "padding_bits": p.padding_bits},
                                      "rows": rows, "files": entry_m}
            print(f"  {name}: {len(entry_m)} files", flush=True)
    files["full_size_digests.json"] = (json.dumps(full_size_digests(workers), indent=1) + "\n").encode()
    files["MANIFEST.json"] = (json.dumps(manifest, indent=1, sort_keys=True) + "\n").encode()
    return files


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true", help="recompute and compare with the committed fixtures")
    ap.add_argument("--workers", type=int, default=min(8, os.cpu_count() or 1))
    args = ap.parse_args()
    files = generate(args.workers)
    if args.check:
        bad = []
        for rel, blob in files.items():
            path = os.path.join(GOLDEN, rel)
            if not os.path.exists(path) or open(path, "rb").read() != blob:
                bad.append(rel)
        if bad:
            raise SystemExit(f"golden fixtures differ from a fresh oracle run: {bad}")
        print(f"{len(files)} golden files reproduce")
        return
    for rel, blob in files.items():
        path = os.path.join(GOLDEN, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "wb") as f:
            f.write(blob)
    total = sum(len(b) for b in files.values())
    print(f"wrote {len(files)} files, {total / 1e6:.2f} MB, under {GOLDEN}")


if __name__ == "__main__":
    main()
