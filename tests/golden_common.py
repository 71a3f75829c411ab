"""Reader of the committed golden fixtures (tests/golden/, written by tools/make_golden.py).

The files are in the library's on-disk format (include/tfhe_hip.h); this is a second, independent
reader in plain numpy (the Rust replay harness under rust/ carries a third), so the fixtures can be
checked without the HIP library."""
import hashlib
import json
import os
import struct

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MAGIC = b"TFHEAMD\x01"


def fnv1a64(data: bytes) -> int:
    h = 0xCBF29CE484222325
    # vectorised FNV is awkward (serial dependency); the payloads are < 1 MB, chunked python is fine
    mask = (1 << 64) - 1
    for b in data:
        h = ((h ^ b) * 0x100000001B3) & mask
    return h


def read_tfhe_file(path: str, verify_checksum: bool = False):
    """-> (kind, params[12], flags, array).  The FNV-1a checksum is a pure-python loop, so it is only
    verified on request (test_golden_oracle does it once per file); the SHA-256 of MANIFEST.json is
    always cheap."""
    with open(path, "rb") as f:
        blob = f.read()
    assert blob[:8] == MAGIC, path
    kind, flags = struct.unpack_from("<II", blob, 8)
    params = struct.unpack_from("<12I", blob, 16)
    ndims, = struct.unpack_from("<I", blob, 64)
    dims = struct.unpack_from("<4I", blob, 68)
    words, checksum = struct.unpack_from("<QQ", blob, 88)
    assert 1 <= ndims <= 4 and int(np.prod(dims[:ndims], dtype=np.uint64)) == words, path
    assert len(blob) == 104 + 4 * words, path
    if verify_checksum:
        assert fnv1a64(blob[104:]) == checksum, path
    arr = np.frombuffer(blob, dtype="<u4", offset=104).reshape(dims[:ndims]).astype(np.uint32)
    return kind, params, flags, arr


def manifest():
    with open(os.path.join(GOLDEN, "MANIFEST.json")) as f:
        return json.load(f)


def full_size_digests():
    with open(os.path.join(GOLDEN, "full_size_digests.json")) as f:
        return json.load(f)


def load_set(name: str, oracle_module=None):
    """-> (params dict, {array name: ndarray}); acc_after_each comes back as [rows][n][k+1][N].
    Every file is checked against the SHA-256 the manifest recorded."""
    m = manifest()["sets"][name]
    arrays = {}
    for key, meta in m["files"].items():
        path = os.path.join(GOLDEN, name, key + ".tfhe")
        with open(path, "rb") as f:
            assert hashlib.sha256(f.read()).hexdigest() == meta["sha256_file"], f"{path} does not match MANIFEST.json"
        arrays[key] = read_tfhe_file(path)[3]
        assert list(arrays[key].shape) == meta["shape"], path
    p = m["params"]
    rows = m["rows"]
    arrays["acc_after_each"] = arrays["acc_after_each"].reshape(rows, p["n"], p["k"] + 1, 1 << p["log_n"])
    return p, arrays


def oracle_params(orc, p: dict):
    return orc.Params(p["k"], p["log_n"], p["n"], orc.Decomposer(*p["pbs"]), orc.Decomposer(*p["ks"]),
                      log_p=p["log_p"], padding_bits=p["padding_bits"])


def sha_row(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u4").tobytes()).hexdigest()
