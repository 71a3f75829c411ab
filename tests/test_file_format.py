"""CPU: the on-disk format for keys and ciphertexts (include/tfhe_hip.h): round trips in
the reference layouts, and refusal of foreign, truncated, padded or corrupted files."""
import numpy as np
import pytest

from gpu_common import pkg, rand_u32, to_pkg_params


def test_round_trip_of_every_kind(oracle, tmp_path):
    m = pkg()
    p = oracle.REF_TEST
    pp = to_pkg_params(p)
    rng = np.random.default_rng(1)
    arrays = {
        m.FILE_BSK: rand_u32(rng, p.bsk_shape()),
        m.FILE_KSK: rand_u32(rng, p.ksk_shape()),
        m.FILE_LWE: rand_u32(rng, (7, p.n + 1)),
        m.FILE_GLWE: rand_u32(rng, (3, p.k + 1, p.N)),
        m.FILE_GGSW: rand_u32(rng, (2, p.R, p.k + 1, p.N)),
    }
    for kind, arr in arrays.items():
        path = str(tmp_path / f"a{kind}.tfhe")
        m.save_array(path, kind, pp, arr, aligned=(kind == m.FILE_KSK))
        k2, p2, aligned, back = m.load_array(path)
        assert k2 == kind and p2 == pp and aligned == (kind == m.FILE_KSK)
        assert back.shape == arr.shape and np.array_equal(back, arr)
    single = rand_u32(rng, (p.n + 1,))
    m.save_array(str(tmp_path / "one.tfhe"), m.FILE_LWE, pp, single)
    assert np.array_equal(m.load_array(str(tmp_path / "one.tfhe"))[3], single)


def test_bootstrapping_key_files_check_parameters(oracle, tmp_path):
    m = pkg()
    p = oracle.REF_TEST
    pp = to_pkg_params(p)
    _, bsk, ksk, _ = oracle.synthetic_inputs(p, 1, cfg_index=3)
    prefix = str(tmp_path / "key")
    m.save_bootstrapping_key(prefix, pp, bsk, ksk)
    b2, k2 = m.load_bootstrapping_key(prefix, pp)
    assert np.array_equal(b2, bsk) and np.array_equal(k2, ksk)
    other = to_pkg_params(oracle.Params(2, 9, 4, oracle.Decomposer(4, 5)))
    with pytest.raises(m.TfheError) as e:
        m.load_bootstrapping_key(prefix, other)
    assert e.value.status == m.TFHE_ERR_INVALID_PARAMS
    with pytest.raises(m.TfheError):
        m.load_bootstrapping_key(prefix, pp, aligned=True)   # written for the literal decomposer


def test_damaged_files_are_refused(oracle, tmp_path):
    m = pkg()
    pp = to_pkg_params(oracle.REF_TEST)
    arr = rand_u32(np.random.default_rng(2), (5, 9))
    path = str(tmp_path / "x.tfhe")
    m.save_array(path, m.FILE_LWE, pp, arr)
    raw = open(path, "rb").read()
    assert len(raw) == 104 + arr.size * 4 and raw[:8] == b"TFHEAMD\x01"

    def expect_io(data):
        bad = str(tmp_path / "bad.tfhe")
        open(bad, "wb").write(data)
        with pytest.raises(m.TfheError) as e:
            m.load_array(bad)
        assert e.value.status == m.TFHE_ERR_IO

    expect_io(raw[:-4])                                   # truncated payload
    expect_io(raw + b"\0\0\0\0")                          # trailing bytes
    expect_io(b"NOTTFHE\x01" + raw[8:])                   # foreign magic
    expect_io(raw[:50])                                   # truncated header
    flipped = bytearray(raw)
    flipped[200] ^= 0x10
    expect_io(bytes(flipped))                             # one payload bit flipped: checksum
    dims = bytearray(raw)
    dims[68] = 6                                          # dims no longer multiply to the word count
    expect_io(bytes(dims))
    with pytest.raises(m.TfheError) as e:
        m.load_array(str(tmp_path / "missing.tfhe"))
    assert e.value.status == m.TFHE_ERR_IO
