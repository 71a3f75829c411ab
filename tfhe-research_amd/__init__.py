"""tfhe-research_amd -- MI355X-native TFHE programmable-bootstrapping engine.

This module is plumbing: a ctypes binding of the C ABI (include/tfhe_hip.h) that accepts numpy
arrays (host entry points) or torch CUDA/HIP tensors (device entry points, PyTorch is used only for
device memory, streams and torch.distributed).  All arithmetic happens in the HIP library
(csrc/*.hip).  There is NO CPU fallback: importing works anywhere, but every compute call needs
libtfhe_hip.so and a GPU and raises loudly otherwise.

The directory name contains a hyphen, so the package is registered under the importable name
`tfhe_research_amd` by __graft_entry__.load_package().
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TFHE_HIP_LIB") or os.path.join(_HERE, "libtfhe_hip.so")  # env: dev builds only

TFHE_OK = 0
STATUS_NAMES = {
    0: "TFHE_OK", 1: "TFHE_ERR_INVALID_PARAMS", 2: "TFHE_ERR_UNSUPPORTED", 3: "TFHE_ERR_NO_KEY",
    4: "TFHE_ERR_HIP", 5: "TFHE_ERR_INVALID_ARGUMENT", 6: "TFHE_ERR_NO_DEVICE", 7: "TFHE_ERR_EXACTNESS",
    8: "TFHE_ERR_IO",
}
(TFHE_ERR_INVALID_PARAMS, TFHE_ERR_UNSUPPORTED, TFHE_ERR_NO_KEY, TFHE_ERR_HIP, TFHE_ERR_INVALID_ARGUMENT,
 TFHE_ERR_NO_DEVICE, TFHE_ERR_EXACTNESS, TFHE_ERR_IO) = range(1, 9)
FILE_BSK, FILE_KSK, FILE_LWE, FILE_GLWE, FILE_GGSW, FILE_WORDS = 1, 2, 3, 4, 5, 6
DECOMPOSER_PBS, DECOMPOSER_KS = 0, 1
BACKEND_AUTO, BACKEND_GOLDILOCKS, BACKEND_FP64, BACKEND_GOLDILOCKS_SPLIT, BACKEND_FP64_P49, BACKEND_FP64_FFT = 0, 1, 2, 3, 4, 5
SHAPE_AUTO, SHAPE_WIDE, SHAPE_TEAM = 0, 1, 2   # tfhe_context_set_kernel_shape

# truth[(lhs << 1) | rhs]
GATE_AND = (0, 0, 0, 1)
GATE_OR = (0, 1, 1, 1)
GATE_NAND = (1, 1, 1, 0)
GATE_XOR = (0, 1, 1, 0)


class TfheError(RuntimeError):
    def __init__(self, status: int, message: str = ""):
        self.status = status
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")


class _CDecomposer(C.Structure):
    _fields_ = [("log_base", C.c_uint32), ("levels", C.c_uint32), ("log_q", C.c_uint32)]


class _CParams(C.Structure):
    _fields_ = [("glwe_dimension", C.c_uint32), ("glwe_poly_degree", C.c_uint32),
                ("lwe_dimension", C.c_uint32), ("padding_bits", C.c_uint32), ("log_p", C.c_uint32),
                ("log_q", C.c_uint32), ("ks_decomposer", _CDecomposer), ("pbs_decomposer", _CDecomposer)]


@dataclass(frozen=True)
class DecomposerParams:
    """decomposer.rs:2-6"""
    log_base: int
    levels: int
    log_q: int = 32


@dataclass(frozen=True)
class TfheParams:
    """lib.rs:23-34 (glwe_poly_degree = log2 N as in the reference)."""
    glwe_dimension: int
    glwe_poly_degree: int
    lwe_dimension: int
    pbs_decomposer: DecomposerParams
    ks_decomposer: DecomposerParams = field(default_factory=lambda: DecomposerParams(4, 5))
    log_p: int = 2
    padding_bits: int = 1
    log_q: int = 32
    # noise parameters (lib.rs:96-97,120-121); only key generation / encryption helpers read them
    lwe_std_dev: float = 0.000013071021089943935
    glwe_std_dev: float = 0.00000004990272175010415

    @property
    def N(self) -> int:
        return 1 << self.glwe_poly_degree

    @property
    def k(self) -> int:
        return self.glwe_dimension

    @property
    def n(self) -> int:
        return self.lwe_dimension

    @property
    def R(self) -> int:
        return (self.k + 1) * self.pbs_decomposer.levels

    @property
    def big_n(self) -> int:
        return self.N * self.k

    def bsk_shape(self):
        return (self.n, self.R, self.k + 1, self.N)

    def bsk_bmmp_shape(self):
        """the unrolled (BMMP) key: three GGSWs per pair of key bits, flattened [n/2*3][R][k+1][N]"""
        return (self.n // 2 * 3, self.R, self.k + 1, self.N)

    def ksk_shape(self):
        return (self.big_n * self.ks_decomposer.levels, self.n + 1)

    def external_product_bytes(self) -> int:
        """algorithmic bytes of one GGSW x GLWE external product in the reference's u32 layout,
        each operand moved once: 4*N*(k+1)*((k+1)*l + 2)  """
        return 4 * self.N * (self.k + 1) * (self.R + 2)

    def _c(self) -> _CParams:
        d = self.pbs_decomposer
        s = self.ks_decomposer
        return _CParams(self.glwe_dimension, self.glwe_poly_degree, self.lwe_dimension,
                        self.padding_bits, self.log_p, self.log_q,
                        _CDecomposer(s.log_base, s.levels, s.log_q),
                        _CDecomposer(d.log_base, d.levels, d.log_q))


_lib = None
_u32p = C.POINTER(C.c_uint32)


def library_available() -> bool:
    return os.path.exists(LIB_PATH)


def _preload_torch_hip_runtime():
    """PyTorch wheels bundle their own libamdhip64.so.  Two HIP runtimes in one process cannot both
    own the GPU (the second one reports "no ROCm-capable device"), so when torch is installed its
    copy is loaded first and our library binds to it by SONAME, whatever the import order."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """The HIP shared library.  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        _preload_torch_hip_runtime()
        _lib = C.CDLL(LIB_PATH)
        _lib.tfhe_last_error.restype = C.c_char_p
        _lib.tfhe_status_string.restype = C.c_char_p
        _lib.tfhe_version.restype = C.c_char_p
    return _lib


def _np(x) -> np.ndarray:
    return np.ascontiguousarray(x, dtype=np.uint32)


def _hp(a: np.ndarray):
    return a.ctypes.data_as(_u32p)


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _dp(t):
    """device pointer of a contiguous 4-byte torch tensor"""
    assert t.is_cuda and t.is_contiguous() and t.element_size() == 4, "need contiguous 32-bit CUDA tensor"
    return C.cast(C.c_void_p(t.data_ptr()), _u32p)


def construct_test_from_lut(params: TfheParams, lut) -> np.ndarray:
    """test_vector.rs:38-67 (host side of the C ABI)."""
    lut = _np(lut)
    out = np.zeros(params.N, dtype=np.uint32)
    cp = params._c()
    st = lib().tfhe_construct_test_from_lut(C.byref(cp), _hp(lut), C.c_size_t(lut.size), _hp(out))
    if st:
        raise TfheError(st, "construct_test_from_lut")
    return out


def construct_identity_test_vector(params: TfheParams) -> np.ndarray:
    """test_vector.rs:23-35"""
    return construct_test_from_lut(params, np.arange(1 << params.log_p, dtype=np.uint32))


def construct_test_vector_boolean(params: TfheParams, truth) -> np.ndarray:
    """test_vector.rs:5-20; truth[(lhs << 1) | rhs]"""
    out = np.zeros(params.N, dtype=np.uint32)
    cp = params._c()
    arr = (C.c_uint32 * 4)(*[int(v) for v in truth])
    st = lib().tfhe_construct_test_vector_boolean(C.byref(cp), arr, _hp(out))
    if st:
        raise TfheError(st, "construct_test_vector_boolean")
    return out


# -- on-disk format (host only; include/tfhe_hip.h "on-disk format") -----------------------------
def save_array(path: str, kind: int, params: TfheParams, array, aligned: bool = False) -> None:
    """One key or ciphertext array per file, in the layout the ABI takes it."""
    a = _np(array)
    assert 1 <= a.ndim <= 4
    dims = (C.c_uint32 * a.ndim)(*a.shape)
    cp = params._c()
    st = lib().tfhe_file_write(os.fsencode(path), C.c_uint32(kind), C.byref(cp), C.c_uint32(int(aligned)),
                               dims, C.c_uint32(a.ndim), _hp(a))
    if st:
        raise TfheError(st, f"writing {path}")


def load_array(path: str):
    """-> (kind, TfheParams, aligned, array); raises TfheError(TFHE_ERR_IO) on a foreign, truncated
    or corrupt file."""
    kind, flags, ndims, words = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64()
    dims = (C.c_uint32 * 4)()
    cp = _CParams()
    st = lib().tfhe_file_read_header(os.fsencode(path), C.byref(kind), C.byref(cp), C.byref(flags), dims,
                                     C.byref(ndims), C.byref(words))
    if st:
        raise TfheError(st, f"reading {path}")
    out = np.zeros(tuple(dims[i] for i in range(ndims.value)), dtype=np.uint32)
    st = lib().tfhe_file_read(os.fsencode(path), _hp(out), C.c_uint64(out.size))
    if st:
        raise TfheError(st, f"reading {path}")
    params = TfheParams(cp.glwe_dimension, cp.glwe_poly_degree, cp.lwe_dimension,
                        DecomposerParams(cp.pbs_decomposer.log_base, cp.pbs_decomposer.levels),
                        DecomposerParams(cp.ks_decomposer.log_base, cp.ks_decomposer.levels),
                        log_p=cp.log_p, padding_bits=cp.padding_bits)
    return kind.value, params, bool(flags.value & 1), out


def save_bootstrapping_key(prefix: str, params: TfheParams, bsk, ksk, aligned: bool = False) -> None:
    """BootstrappingKey (bootstrapping.rs:18-21) as <prefix>.bsk + <prefix>.ksk"""
    save_array(prefix + ".bsk", FILE_BSK, params, bsk, aligned)
    save_array(prefix + ".ksk", FILE_KSK, params, ksk, aligned)


def load_bootstrapping_key(prefix: str, params: TfheParams, aligned: bool = False):
    """-> (bsk, ksk); refuses files written for other parameters or the other decomposer alignment"""
    out = []
    for ext, want in ((".bsk", FILE_BSK), (".ksk", FILE_KSK)):
        kind, p, al, arr = load_array(prefix + ext)
        # the file stores the 12 integer fields only (not the noise std-devs): compare those
        if kind != want or bytes(p._c()) != bytes(params._c()) or al != aligned:
            raise TfheError(TFHE_ERR_INVALID_PARAMS, f"{prefix + ext} holds kind {kind} for {p} (aligned={al})")
        out.append(arr)
    return tuple(out)


def params_validate(params: TfheParams) -> int:
    cp = params._c()
    return lib().tfhe_params_validate(C.byref(cp))


class SystemRng:
    """Cryptographic randomness for key generation and encryption: every draw is os.urandom (the
    kernel CSPRNG), the counterpart of the reference's `R: CryptoRng + RngCore` / thread_rng
    (lwe.rs:55, glwe.rs:177, utils.rs:36-77).  Offers the three numpy-Generator methods the
    convenience helpers use, so a seeded numpy Generator can stand in for it IN TESTS ONLY."""

    @staticmethod
    def _u64(count: int) -> np.ndarray:
        return np.frombuffer(os.urandom(8 * count), dtype=np.uint64)

    def integers(self, low: int, high: int, size=None, dtype=np.int64) -> np.ndarray:
        """uniform integers in [low, high); high - low must be a power of two (2 and 2^32 are the
        only spans the callers need, so no rejection step and no modulo bias)"""
        span = int(high) - int(low)
        if span <= 0 or span & (span - 1):
            raise ValueError("SystemRng.integers: span must be a power of two")
        shape = () if size is None else (size if isinstance(size, tuple) else (size,))
        count = int(np.prod(shape, dtype=np.int64)) if shape else 1
        vals = (self._u64(count) & np.uint64(span - 1)).astype(np.int64) + int(low)
        return vals.astype(dtype).reshape(shape)

    def normal(self, loc: float, scale: float, size=None) -> np.ndarray:
        """Gaussian by Box-Muller over 53-bit uniforms from the CSPRNG"""
        shape = () if size is None else (size if isinstance(size, tuple) else (size,))
        count = int(np.prod(shape, dtype=np.int64)) if shape else 1
        half = (count + 1) // 2
        u = self._u64(2 * half)
        u1 = ((u[:half] >> np.uint64(11)).astype(np.float64) + 1.0) * 2.0 ** -53   # (0, 1]
        u2 = (u[half:] >> np.uint64(11)).astype(np.float64) * 2.0 ** -53            # [0, 1)
        r = np.sqrt(-2.0 * np.log(u1))
        z = np.concatenate([r * np.cos(2.0 * np.pi * u2), r * np.sin(2.0 * np.pi * u2)])[:count]
        return (loc + scale * z).reshape(shape)


class Context:
    """One GPU context = one device + one stream + one loaded BootstrappingKey."""

    def __init__(self, params: TfheParams, device: int = 0, backend: int = BACKEND_AUTO):
        self.params = params
        self._h = C.c_void_p()
        cp = params._c()
        lib().tfhe_context_backend.restype = C.c_char_p
        st = lib().tfhe_context_create_with_backend(C.byref(cp), C.c_int(device), C.c_int(backend),
                                                    C.byref(self._h))
        if st:
            self._h = C.c_void_p()
            raise TfheError(st, lib().tfhe_status_string(st).decode())

    # -- lifecycle ------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().tfhe_context_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, st: int):
        if st:
            raise TfheError(st, lib().tfhe_last_error(self._h).decode())

    @property
    def backend(self) -> str:
        return lib().tfhe_context_backend(self._h).decode()

    # -- extensions beyond the reference -------------------------------------------
    def set_decomposer_alignment(self, aligned: bool):
        """False: the reference's literal decomposer (bit-exact with the crate).  True: limbs and
        gadget factors counted down from bit 32, so bases with log_base not dividing 32 decrypt."""
        self._check(lib().tfhe_context_set_decomposer_alignment(self._h, C.c_int(int(aligned))))

    def set_kernel_shape(self, shape: int):
        """SHAPE_AUTO (default: by batch size), SHAPE_WIDE (2 (k+1) waves per sample: the latency shape, fp64-fft up to
        N = 1024) or SHAPE_TEAM (the throughput shape) for every blind rotation of this context; same bits either way"""
        self._check(lib().tfhe_context_set_kernel_shape(self._h, C.c_int(int(shape))))

    def set_bootstrap_order(self, ks_first: bool):
        """False: PBS then key switch (bootstrapping.rs:58-120), ciphertexts of n+1 words.  True:
        key switch then PBS (notes/TFHE.md:367-400), ciphertexts of k*N+1 words."""
        self._check(lib().tfhe_context_set_bootstrap_order(self._h, C.c_int(int(ks_first))))
        self._ks_first = bool(ks_first)

    @property
    def io_dim(self) -> int:
        """LWE dimension of ciphertexts at the bootstrap / gate boundary"""
        return self.params.big_n if getattr(self, "_ks_first", False) else self.params.n

    def prepared_ggsw_words(self) -> int:
        w = C.c_size_t()
        self._check(lib().tfhe_prepared_ggsw_words(self._h, C.byref(w)))
        return w.value

    def set_stream(self, hip_stream: int | None):
        """hip_stream: a hipStream_t handle (0 = HIP's default stream); None = back to a private stream."""
        self._bound_stream = None
        if hip_stream is None:
            self._check(lib().tfhe_context_use_own_stream(self._h))
        else:
            self._check(lib().tfhe_context_set_stream(self._h, C.c_void_p(hip_stream)))

    def use_torch_stream(self):
        """Bind to torch's current stream now.  The torch-tensor entry points do this by themselves on
        every call (_bind_torch); this is for callers that mix in host-pointer calls."""
        import torch
        self._bound_stream = torch.cuda.current_stream().cuda_stream
        self.set_stream(self._bound_stream)

    def _bind_torch(self):
        """Every torch-tensor entry point runs on torch's CURRENT stream, so it is ordered after the
        kernels that produced its inputs and before whatever consumes `out` on that stream, exactly
        like a torch op.  Re-binding only happens when the current stream changed since the last
        call (tfhe_context_set_stream then drains the stream it leaves: the workspace is shared)."""
        import torch
        s = torch.cuda.current_stream().cuda_stream
        if getattr(self, "_bound_stream", None) != s:
            self.set_stream(s)
            self._bound_stream = s

    def synchronize(self):
        self._check(lib().tfhe_context_synchronize(self._h))

    def reserve(self, max_batch: int):
        self._check(lib().tfhe_context_reserve(self._h, C.c_size_t(max_batch)))

    def set_timing(self, enable: bool):
        self._check(lib().tfhe_context_set_timing(self._h, C.c_int(int(enable))))

    def measure_hbm_copy(self, mib: int = 1024, reps: int = 10) -> float:
        """GB/s (read + write) of a 16-byte-per-lane stream copy: the HBM roofline of this device now."""
        out = C.c_double()
        self._check(lib().tfhe_measure_hbm_copy(self._h, C.c_size_t(mib << 20), C.c_int(reps), C.byref(out)))
        return out.value

    def fft_margin(self, reset: bool = True) -> float:
        """largest |value - nearest integer| the fp64-fft kernels have lifted since the last reset; only the probe
        build (libtfhe_hip_probe.so via TFHE_HIP_LIB, test instrumentation) records it -- the product library raises
        TFHE_ERR_UNSUPPORTED"""
        out = C.c_double()
        self._check(lib().tfhe_debug_fft_margin(self._h, C.byref(out), C.c_int(int(reset))))
        return out.value

    def blind_rotate_plan(self, batch: int) -> dict:
        """how a bootstrap of `batch` samples sends out its blind rotations (tfhe_debug_blind_rotate_plan)"""
        group, resident = C.c_size_t(), C.c_size_t()
        segments, streams = C.c_uint(), C.c_uint()
        self._check(lib().tfhe_debug_blind_rotate_plan(self._h, C.c_size_t(batch), C.byref(group), C.byref(segments),
                                                       C.byref(streams), C.byref(resident)))
        groups = -(-batch // max(1, group.value))
        waves, per_team = C.c_uint(), C.c_uint()
        self._check(lib().tfhe_debug_blind_rotate_shape(self._h, C.c_size_t(batch), C.byref(waves), C.byref(per_team)))
        k1 = self.params.k + 1
        return {"samples_per_group": group.value, "groups": groups, "segments": segments.value, "streams": streams.value,
                "launches": groups * segments.value * streams.value, "resident_samples": resident.value,
                "waves_per_team": waves.value, "samples_per_team": per_team.value,
                "kernel": "wide team (2 waves per polynomial: split by level and key part)"
                          if waves.value == 2 * k1 and per_team.value == 1 and segments.value == 1 and self.backend == "fp64-fft"
                          and self.params.glwe_poly_degree <= 10
                          else "pair (one wave per sample, both polynomials side by side)" if waves.value == 1 else "team"}

    def last_kernel_ms(self):
        br, ks = C.c_float(), C.c_float()
        self._check(lib().tfhe_last_kernel_ms(self._h, C.byref(br), C.byref(ks)))
        return br.value, ks.value

    def kernel_ms_ago(self, steps_ago: int):
        """(blind rotation ms, key switch ms) of the timed bootstrap `steps_ago` calls before the last one; the context
        keeps the last 64, so a timed loop needs no host synchronisation inside"""
        br, ks = C.c_float(), C.c_float()
        self._check(lib().tfhe_kernel_ms_ago(self._h, C.c_uint(steps_ago), C.byref(br), C.byref(ks)))
        return br.value, ks.value

    # -- keys -----------------------------------------------------------------------------------
    def load_bootstrapping_key(self, bsk, ksk):
        """bsk [n][R][k+1][N], ksk [k*N*l_ks][n+1]: reference layouts (numpy or torch device)."""
        p = self.params
        if _is_torch(bsk):
            self._bind_torch()
            assert tuple(bsk.shape) == p.bsk_shape() and tuple(ksk.shape) == p.ksk_shape()
            self._check(lib().tfhe_load_bootstrapping_key_device(self._h, _dp(bsk), _dp(ksk)))
        else:
            bsk, ksk = _np(bsk), _np(ksk)
            assert bsk.shape == p.bsk_shape() and ksk.shape == p.ksk_shape()
            self._check(lib().tfhe_load_bootstrapping_key(self._h, _hp(bsk), _hp(ksk)))

    def load_bootstrapping_key_bmmp(self, bsk_bmmp, ksk):
        """Key of the unrolled blind rotation (notes/BMMP Bootstrapping.md): bsk_bmmp [n/2*3][R][k+1][N]
        = GGSW(s s'), GGSW(s (1-s')), GGSW(s' (1-s)) per pair of key bits.  Bootstraps and gates then use
        it; needs N = 512 and even n.  Same plaintexts as the reference's bootstrap, different bits."""
        p = self.params
        if _is_torch(bsk_bmmp):
            self._bind_torch()
            assert tuple(bsk_bmmp.shape) == p.bsk_bmmp_shape() and tuple(ksk.shape) == p.ksk_shape()
            self._check(lib().tfhe_load_bootstrapping_key_bmmp_device(self._h, _dp(bsk_bmmp), _dp(ksk)))
        else:
            bsk_bmmp, ksk = _np(bsk_bmmp), _np(ksk)
            assert bsk_bmmp.shape == p.bsk_bmmp_shape() and ksk.shape == p.ksk_shape()
            self._check(lib().tfhe_load_bootstrapping_key_bmmp(self._h, _hp(bsk_bmmp), _hp(ksk)))

    @property
    def uses_bmmp(self) -> bool:
        return bool(lib().tfhe_context_uses_bmmp(self._h))

    # -- hot path -------------------------------------------------------------------------------
    def _tv_count(self, tv, batch):
        n_tv = 1 if tv.ndim == 1 else tv.shape[0]
        assert tv.shape[-1] == self.params.N and n_tv in (1, batch)
        return n_tv

    def bootstrap(self, lwe_in, test_vector_poly, out=None):
        """bootstrap(): bootstrapping.rs:58-120 over a batch [batch][n+1]."""
        p = self.params
        if _is_torch(lwe_in):
            self._bind_torch()
            import torch
            batch = lwe_in.shape[0]
            assert lwe_in.shape[1] == self.io_dim + 1
            if out is None:
                out = torch.empty_like(lwe_in)
            self._check(lib().tfhe_bootstrap_batch_device(
                self._h, _dp(lwe_in), C.c_size_t(batch), _dp(test_vector_poly),
                C.c_size_t(self._tv_count(test_vector_poly, batch)), _dp(out)))
            return out
        lwe_in, tv = _np(lwe_in), _np(test_vector_poly)
        single = lwe_in.ndim == 1
        lwe2 = lwe_in.reshape(-1, self.io_dim + 1)
        res = np.zeros_like(lwe2)
        self._check(lib().tfhe_bootstrap_batch(self._h, _hp(lwe2), C.c_size_t(lwe2.shape[0]), _hp(tv),
                                               C.c_size_t(self._tv_count(tv, lwe2.shape[0])), _hp(res)))
        return res[0] if single else res

    def blind_rotate(self, lwe_in, test_vector_poly, out=None):
        """bootstrapping.rs:67-105 -> GLWE accumulators [batch][k+1][N]."""
        p = self.params
        if _is_torch(lwe_in):
            self._bind_torch()
            import torch
            batch = lwe_in.shape[0]
            if out is None:
                out = torch.empty((batch, p.k + 1, p.N), dtype=lwe_in.dtype, device=lwe_in.device)
            self._check(lib().tfhe_blind_rotate_batch_device(
                self._h, _dp(lwe_in), C.c_size_t(batch), _dp(test_vector_poly),
                C.c_size_t(self._tv_count(test_vector_poly, batch)), _dp(out)))
            return out
        lwe2, tv = _np(lwe_in).reshape(-1, p.n + 1), _np(test_vector_poly)
        res = np.zeros((lwe2.shape[0], p.k + 1, p.N), dtype=np.uint32)
        self._check(lib().tfhe_blind_rotate_batch(self._h, _hp(lwe2), C.c_size_t(lwe2.shape[0]), _hp(tv),
                                                  C.c_size_t(self._tv_count(tv, lwe2.shape[0])), _hp(res)))
        return res

    def sample_extract(self, glwe, sample_index: int = 0) -> np.ndarray:
        p = self.params
        g = _np(glwe).reshape(-1, p.k + 1, p.N)
        res = np.zeros((g.shape[0], p.big_n + 1), dtype=np.uint32)
        self._check(lib().tfhe_sample_extract_batch(self._h, _hp(g), C.c_size_t(g.shape[0]),
                                                    C.c_size_t(sample_index), _hp(res)))
        return res

    def key_switch(self, lwe_big, out=None):
        """key_switch_lwe(): key_switching.rs:63-103 with the loaded KSK."""
        p = self.params
        if _is_torch(lwe_big):
            self._bind_torch()
            import torch
            batch = lwe_big.shape[0]
            if out is None:
                out = torch.empty((batch, p.n + 1), dtype=lwe_big.dtype, device=lwe_big.device)
            self._check(lib().tfhe_key_switch_batch_device(self._h, _dp(lwe_big), C.c_size_t(batch), _dp(out)))
            return out
        x = _np(lwe_big).reshape(-1, p.big_n + 1)
        res = np.zeros((x.shape[0], p.n + 1), dtype=np.uint32)
        self._check(lib().tfhe_key_switch_batch(self._h, _hp(x), C.c_size_t(x.shape[0]), _hp(res)))
        return res

    def external_product(self, ggsw, glwe) -> np.ndarray:
        """external_product(): ggsw.rs:132-161.  ggsw [R][k+1][N] (shared) or [batch][R][k+1][N]."""
        p = self.params
        g = _np(glwe).reshape(-1, p.k + 1, p.N)
        gg = _np(ggsw)
        count = 1 if gg.ndim == 3 else gg.shape[0]
        res = np.zeros_like(g)
        self._check(lib().tfhe_external_product_batch(self._h, _hp(gg), C.c_size_t(count), _hp(g),
                                                      C.c_size_t(g.shape[0]), _hp(res)))
        return res

    def prepare_ggsw_device(self, ggsw, out=None):
        """device u32 GGSW(s) -> device NTT-domain GGSW(s) (torch int64 tensor)."""
        import torch
        self._bind_torch()
        count = 1 if ggsw.dim() == 3 else ggsw.shape[0]
        if out is None:
            out = torch.empty((count, self.prepared_ggsw_words()), dtype=torch.int64, device=ggsw.device)
        self._check(lib().tfhe_prepare_ggsw_device(self._h, _dp(ggsw), C.c_size_t(count),
                                                   C.c_void_p(out.data_ptr())))
        return out

    def external_product_prepared(self, ggsw_prepared, glwe, out=None):
        import torch
        self._bind_torch()
        p = self.params
        batch = glwe.shape[0]
        count = ggsw_prepared.shape[0]
        if out is None:
            out = torch.empty_like(glwe)
        self._check(lib().tfhe_external_product_prepared_device(
            self._h, C.c_void_p(ggsw_prepared.data_ptr()), C.c_size_t(count), _dp(glwe),
            C.c_size_t(batch), _dp(out)))
        return out

    def cmux(self, ggsw, ct0, ct1):
        """cmux(): ggsw.rs:164-178 -> (result, ct1 clobbered with ct1 - ct0)."""
        p = self.params
        c0 = _np(ct0).reshape(-1, p.k + 1, p.N)
        c1 = _np(ct1).reshape(-1, p.k + 1, p.N).copy()
        gg = _np(ggsw)
        count = 1 if gg.ndim == 3 else gg.shape[0]
        res = np.zeros_like(c0)
        self._check(lib().tfhe_cmux_batch(self._h, _hp(gg), C.c_size_t(count), _hp(c0), _hp(c1),
                                          C.c_size_t(c0.shape[0]), _hp(res)))
        return res, c1

    # -- small ops ------------------------------------------------------------------------------
    def decompose(self, values, which: int = DECOMPOSER_PBS) -> np.ndarray:
        v = _np(values).ravel()
        d = self.params.pbs_decomposer if which == DECOMPOSER_PBS else self.params.ks_decomposer
        res = np.zeros((v.size, d.levels), dtype=np.uint32)
        self._check(lib().tfhe_decompose(self._h, C.c_int(which), _hp(v), C.c_size_t(v.size), _hp(res)))
        return res

    def decompose_glwe(self, glwe) -> np.ndarray:
        p = self.params
        g = _np(glwe).reshape(-1, p.k + 1, p.N)
        res = np.zeros((g.shape[0], p.R, p.N), dtype=np.uint32)
        self._check(lib().tfhe_decompose_glwe_batch(self._h, _hp(g), C.c_size_t(g.shape[0]), _hp(res)))
        return res

    def switch_modulus(self, values, log_from: int, log_to: int) -> np.ndarray:
        v = _np(values).ravel()
        res = np.zeros_like(v)
        self._check(lib().tfhe_switch_modulus(self._h, _hp(v), C.c_size_t(v.size), C.c_uint32(log_from),
                                              C.c_uint32(log_to), _hp(res)))
        return res

    def glwe_mul_monomial(self, glwe, monomial_index) -> np.ndarray:
        p = self.params
        g = _np(glwe).reshape(-1, p.k + 1, p.N)
        idx = np.ascontiguousarray(np.broadcast_to(np.asarray(monomial_index, dtype=np.int64), (g.shape[0],)))
        res = np.zeros_like(g)
        self._check(lib().tfhe_glwe_mul_monomial_batch(self._h, _hp(g), C.c_size_t(g.shape[0]),
                                                       idx.ctypes.data_as(C.POINTER(C.c_int64)), _hp(res)))
        return res

    def lwe_linear(self, c0: int, ct0, c1: int = 0, ct1=None, out=None):
        """c0*ct0 + c1*ct1 (wrapping): LweCiphertext Add / Mul<u32>, lwe.rs:9-23."""
        if _is_torch(ct0):
            self._bind_torch()
            import torch
            if out is None:
                out = torch.empty_like(ct0)
            width = ct0.shape[-1]
            self._check(lib().tfhe_lwe_linear_batch_device(
                self._h, C.c_uint32(c0 & 0xFFFFFFFF), _dp(ct0), C.c_uint32(c1 & 0xFFFFFFFF),
                _dp(ct1) if ct1 is not None else None, C.c_size_t(ct0.numel() // width), C.c_size_t(width), _dp(out)))
            return out
        a = _np(ct0)
        rows = a.reshape(-1, a.shape[-1])
        b = _np(ct1).reshape(rows.shape) if ct1 is not None else None
        res = np.zeros_like(rows)
        self._check(lib().tfhe_lwe_linear_batch(self._h, C.c_uint32(c0 & 0xFFFFFFFF), _hp(rows),
                                                C.c_uint32(c1 & 0xFFFFFFFF), _hp(b) if b is not None else None,
                                                C.c_size_t(rows.shape[0]), C.c_size_t(rows.shape[1]), _hp(res)))
        return res.reshape(a.shape)

    def lut_gate(self, truth, cts, out=None):
        """Gate of m = len(cts) inputs (notes/Boolean Gates.md:2-11): one PBS of sum_i 2^i * cts[i]
        (cts[0] = rightmost input) with lut[x] = truth[x mod 2^m]; needs log_p >= m."""
        p = self.params
        m = len(cts)
        assert len(truth) == 1 << m
        arr = (C.c_uint32 * (1 << m))(*[int(v) for v in truth])
        ptrs = (_u32p * m)()
        if _is_torch(cts[0]):
            self._bind_torch()
            import torch
            batch = cts[0].shape[0]
            for i, t in enumerate(cts):
                assert tuple(t.shape) == (batch, self.io_dim + 1)
                ptrs[i] = _dp(t)
            if out is None:
                out = torch.empty_like(cts[0])
            self._check(lib().tfhe_lut_gate_batch_device(self._h, arr, C.c_uint32(m), ptrs, C.c_size_t(batch), _dp(out)))
            return out
        host = [_np(t).reshape(-1, self.io_dim + 1) for t in cts]
        for i, t in enumerate(host):
            assert t.shape == host[0].shape
            ptrs[i] = _hp(t)
        res = np.zeros_like(host[0])
        self._check(lib().tfhe_lut_gate_batch(self._h, arr, C.c_uint32(m), ptrs, C.c_size_t(host[0].shape[0]), _hp(res)))
        return res

    def lwe_not(self, ct, out=None):
        """NOT without a bootstrap: (-a, enc(1) - b)."""
        p = self.params
        if _is_torch(ct):
            self._bind_torch()
            import torch
            if out is None:
                out = torch.empty_like(ct)
            self._check(lib().tfhe_lwe_not_batch_device(self._h, _dp(ct), C.c_size_t(ct.numel() // (self.io_dim + 1)), _dp(out)))
            return out
        a = _np(ct)
        rows = a.reshape(-1, self.io_dim + 1)
        res = np.zeros_like(rows)
        self._check(lib().tfhe_lwe_not_batch(self._h, _hp(rows), C.c_size_t(rows.shape[0]), _hp(res)))
        return res.reshape(a.shape)

    def gate(self, truth, ct0, ct1, out=None):
        """and()/or(): boolean.rs:9-53 generalised: bootstrap(2*ct1 + ct0) with the closure's TV."""
        p = self.params
        arr = (C.c_uint32 * 4)(*[int(v) for v in truth])
        if _is_torch(ct0):
            self._bind_torch()
            import torch
            batch = ct0.shape[0]
            if out is None:
                out = torch.empty_like(ct0)
            self._check(lib().tfhe_gate_batch_device(self._h, arr, _dp(ct0), _dp(ct1), C.c_size_t(batch), _dp(out)))
            return out
        a, b = _np(ct0).reshape(-1, self.io_dim + 1), _np(ct1).reshape(-1, self.io_dim + 1)
        res = np.zeros_like(a)
        self._check(lib().tfhe_gate_batch(self._h, arr, _hp(a), _hp(b), C.c_size_t(a.shape[0]), _hp(res)))
        return res

    # -- encryption side: the caller's random draws arrive in the buffers --------
    def glwe_encrypt_zero(self, glwe_sk, samples):
        """encrypt_glwe_zero glwe.rs:190-209: samples [count][k+1][N] (uniform masks, errors in the
        body) -> ciphertexts.  A torch device tensor is completed in place and returned."""
        p = self.params
        sk = _np(glwe_sk).reshape(p.k, p.N)
        if _is_torch(samples):
            self._bind_torch()
            assert tuple(samples.shape[-2:]) == (p.k + 1, p.N)
            self._check(lib().tfhe_glwe_encrypt_zero_batch_device(
                self._h, _hp(sk), _dp(samples), C.c_size_t(samples.numel() // ((p.k + 1) * p.N))))
            return samples
        out = _np(samples).copy().reshape(-1, p.k + 1, p.N)
        self._check(lib().tfhe_glwe_encrypt_zero_batch(self._h, _hp(sk), _hp(out), C.c_size_t(out.shape[0])))
        return out

    def glwe_decrypt(self, glwe_sk, glwe) -> np.ndarray:
        """decrypt_glwe_ciphertext glwe.rs:245-265 -> plaintext polynomials [count][N]."""
        p = self.params
        sk = _np(glwe_sk).reshape(p.k, p.N)
        g = _np(glwe).reshape(-1, p.k + 1, p.N)
        out = np.zeros((g.shape[0], p.N), dtype=np.uint32)
        self._check(lib().tfhe_glwe_decrypt_batch(self._h, _hp(sk), _hp(g), C.c_size_t(g.shape[0]), _hp(out)))
        return out

    def ggsw_encrypt(self, glwe_sk, messages, samples):
        """encrypt_ggsw_plaintext ggsw.rs:76-130 for messages [count]: samples
        [count][(k+1)l][k+1][N] pre-filled row by row."""
        p = self.params
        sk = _np(glwe_sk).reshape(p.k, p.N)
        msg = _np(messages).reshape(-1)
        if _is_torch(samples):
            self._bind_torch()
            assert samples.numel() == msg.size * p.R * (p.k + 1) * p.N
            self._check(lib().tfhe_ggsw_encrypt_batch_device(self._h, _hp(sk), _hp(msg), _dp(samples),
                                                             C.c_size_t(msg.size)))
            return samples
        out = _np(samples).copy().reshape(msg.size, p.R, p.k + 1, p.N)
        self._check(lib().tfhe_ggsw_encrypt_batch(self._h, _hp(sk), _hp(msg), _hp(out), C.c_size_t(msg.size)))
        return out

    def lwe_encrypt(self, lwe_sk, samples, plaintexts=None):
        """encrypt_lwe_plaintext lwe.rs:138-160: samples [batch][dim+1] (uniform masks, error in
        the b slot), plaintexts [batch] already encoded (None = encrypt_lwe_zero)."""
        sk = _np(lwe_sk).reshape(-1)
        dim = sk.size
        if _is_torch(samples):
            self._bind_torch()
            assert samples.shape[-1] == dim + 1
            self._check(lib().tfhe_lwe_encrypt_batch_device(
                self._h, _hp(sk), C.c_size_t(dim), _dp(plaintexts) if plaintexts is not None else None,
                _dp(samples), C.c_size_t(samples.numel() // (dim + 1))))
            return samples
        out = _np(samples).copy().reshape(-1, dim + 1)
        pt = _np(plaintexts).reshape(-1) if plaintexts is not None else None
        assert pt is None or pt.size == out.shape[0]
        self._check(lib().tfhe_lwe_encrypt_batch(self._h, _hp(sk), C.c_size_t(dim),
                                                 _hp(pt) if pt is not None else None, _hp(out),
                                                 C.c_size_t(out.shape[0])))
        return out

    def lwe_decrypt(self, lwe_sk, lwe, out=None):
        """decrypt_lwe lwe.rs:162-173 -> encoded plaintexts [batch] = b - <a, s>."""
        sk = _np(lwe_sk).reshape(-1)
        dim = sk.size
        if _is_torch(lwe):
            self._bind_torch()
            import torch
            batch = lwe.numel() // (dim + 1)
            if out is None:
                out = torch.empty(batch, dtype=lwe.dtype, device=lwe.device)
            self._check(lib().tfhe_lwe_decrypt_batch_device(self._h, _hp(sk), C.c_size_t(dim), _dp(lwe),
                                                            C.c_size_t(batch), _dp(out)))
            return out
        rows = _np(lwe).reshape(-1, dim + 1)
        res = np.zeros(rows.shape[0], dtype=np.uint32)
        self._check(lib().tfhe_lwe_decrypt_batch(self._h, _hp(sk), C.c_size_t(dim), _hp(rows),
                                                 C.c_size_t(rows.shape[0]), _hp(res)))
        return res

    def generate_ksk(self, from_sk, to_sk, samples) -> np.ndarray:
        """KeySwitchingKey::generate_ksk key_switching.rs:20-60 with the context's ks_decomposer:
        samples [from_dim*l_ks][to_dim+1] pre-filled."""
        f, t = _np(from_sk).reshape(-1), _np(to_sk).reshape(-1)
        out = _np(samples).copy()
        assert out.shape == (f.size * self.params.ks_decomposer.levels, t.size + 1)
        self._check(lib().tfhe_generate_ksk(self._h, _hp(f), C.c_size_t(f.size), _hp(t), C.c_size_t(t.size),
                                            _hp(out)))
        return out

    def bootstrapping_key_gen(self, lwe_sk, glwe_sk, bsk_samples, ksk_samples, load: bool = True):
        """bootstrapping_key_gen bootstrapping.rs:23-56 on pre-filled bsk / ksk buffers -> (bsk, ksk);
        `load` installs the key in this context.  Torch device tensors are completed in place."""
        p = self.params
        lsk, gsk = _np(lwe_sk).reshape(p.n), _np(glwe_sk).reshape(p.k, p.N)
        if _is_torch(bsk_samples):
            self._bind_torch()
            assert tuple(bsk_samples.shape) == p.bsk_shape() and tuple(ksk_samples.shape) == p.ksk_shape()
            self._check(lib().tfhe_bootstrapping_key_gen_device(self._h, _hp(lsk), _hp(gsk), _dp(bsk_samples),
                                                                _dp(ksk_samples), C.c_int(int(load))))
            return bsk_samples, ksk_samples
        bsk, ksk = _np(bsk_samples).copy(), _np(ksk_samples).copy()
        assert bsk.shape == p.bsk_shape() and ksk.shape == p.ksk_shape()
        self._check(lib().tfhe_bootstrapping_key_gen(self._h, _hp(lsk), _hp(gsk), _hp(bsk), _hp(ksk),
                                                     C.c_int(int(load))))
        return bsk, ksk

    def bootstrapping_key_gen_bmmp(self, lwe_sk, glwe_sk, bsk_samples, ksk_samples, load: bool = True):
        """The BMMP key on pre-filled buffers: bsk_samples [n/2*3][R][k+1][N], ksk_samples as for
        bootstrapping_key_gen -> (bsk_bmmp, ksk); `load` installs it.  Torch tensors in place."""
        p = self.params
        lsk, gsk = _np(lwe_sk).reshape(p.n), _np(glwe_sk).reshape(p.k, p.N)
        if _is_torch(bsk_samples):
            self._bind_torch()
            assert tuple(bsk_samples.shape) == p.bsk_bmmp_shape() and tuple(ksk_samples.shape) == p.ksk_shape()
            self._check(lib().tfhe_bootstrapping_key_gen_bmmp_device(self._h, _hp(lsk), _hp(gsk), _dp(bsk_samples),
                                                                     _dp(ksk_samples), C.c_int(int(load))))
            return bsk_samples, ksk_samples
        bsk, ksk = _np(bsk_samples).copy(), _np(ksk_samples).copy()
        assert bsk.shape == p.bsk_bmmp_shape() and ksk.shape == p.ksk_shape()
        self._check(lib().tfhe_bootstrapping_key_gen_bmmp(self._h, _hp(lsk), _hp(gsk), _hp(bsk), _hp(ksk),
                                                          C.c_int(int(load))))
        return bsk, ksk

    # -- convenience on top of the encryption-side entry points ---------------------------------
    @staticmethod
    def _noise(rng, std_dev: float, shape) -> np.ndarray:
        """two-sided rounded Gaussian on the 32-bit torus (utils.rs:36-54 without the saturation)"""
        return (np.rint(rng.normal(0.0, std_dev * 2.0 ** 32, size=shape)).astype(np.int64) & 0xFFFFFFFF).astype(np.uint32)

    def generate_keys(self, rng=None, load: bool = True, bmmp: bool = False):
        """LweSecretKey::random + GlweSecretKey::random + bootstrapping_key_gen (lwe.rs:53-60,
        glwe.rs:176-182, bootstrapping.rs:23-56): secrets, masks and errors are drawn here (the
        reference draws them with its `R: CryptoRng + RngCore`), the key material is completed on
        the GPU and, with `load`, installed.  -> (lwe_sk [n], glwe_sk [k][N], bsk, ksk).  bmmp=True
        makes the key of the unrolled blind rotation instead (bsk [n/2*3][R][k+1][N]).

        Randomness: by default every draw comes from the operating system's CSPRNG (SystemRng over
        os.urandom).  `rng=` is a TEST HOOK for reproducible runs: a numpy Generator (PCG64 etc.)
        is NOT acceptable for real keys -- the uniform masks published in the BSK/KSK would be raw
        consecutive outputs of the generator that produced the secret keys just before, and its
        state can be recovered from them."""
        p = self.params
        rng = rng if rng is not None else SystemRng()
        lwe_sk = rng.integers(0, 2, size=p.n).astype(np.uint32)
        glwe_sk = rng.integers(0, 2, size=(p.k, p.N)).astype(np.uint32)
        shape = p.bsk_bmmp_shape() if bmmp else p.bsk_shape()
        bsk = rng.integers(0, 1 << 32, size=shape, dtype=np.uint64).astype(np.uint32)
        bsk[:, :, p.k, :] = self._noise(rng, p.glwe_std_dev, (shape[0], p.R, p.N))
        ksk = rng.integers(0, 1 << 32, size=p.ksk_shape(), dtype=np.uint64).astype(np.uint32)
        ksk[:, p.n] = self._noise(rng, p.lwe_std_dev, ksk.shape[0])
        gen = self.bootstrapping_key_gen_bmmp if bmmp else self.bootstrapping_key_gen
        bsk, ksk = gen(lwe_sk, glwe_sk, bsk, ksk, load=load)
        return lwe_sk, glwe_sk, bsk, ksk

    def encrypt_bits(self, lwe_sk, messages, rng=None) -> np.ndarray:
        """LweCleartext::encode_message + encrypt_lwe_plaintext (lwe.rs:81-92,138-160) for a batch
        of messages < 2^log_p under `lwe_sk` (any dimension) -> [batch][dim+1].  Masks and errors
        come from the OS CSPRNG unless the test hook `rng=` is given (see generate_keys)."""
        p = self.params
        rng = rng if rng is not None else SystemRng()
        msg = np.asarray(messages, dtype=np.uint32).reshape(-1)
        if msg.size and int(msg.max()) >> p.log_p:
            raise TfheError(TFHE_ERR_INVALID_ARGUMENT, "assertion failed: m < 1 << log_p (lwe.rs:84)")
        sk = _np(lwe_sk).reshape(-1)
        samples = rng.integers(0, 1 << 32, size=(msg.size, sk.size + 1), dtype=np.uint64).astype(np.uint32)
        samples[:, sk.size] = self._noise(rng, p.lwe_std_dev, msg.size)
        return self.lwe_encrypt(sk, samples, (msg << (32 - p.log_p - p.padding_bits)).astype(np.uint32))

    def decrypt_bits(self, lwe_sk, lwe) -> np.ndarray:
        """decrypt_lwe + rounding to the nearest message slot (lwe.rs:162-173, :100-107)"""
        p = self.params
        shift = 32 - p.log_p - p.padding_bits
        raw = self.lwe_decrypt(lwe_sk, lwe).astype(np.uint64)
        return (((raw + (1 << (shift - 1))) >> shift) & ((1 << p.log_p) - 1)).astype(np.uint32)


class _BorrowedContext(Context):
    """A pool member's context handle: owned by the pool, never destroyed from here."""

    def __init__(self, params: TfheParams, handle):
        self.params = params
        self._h = C.c_void_p(handle)
        lib().tfhe_context_backend.restype = C.c_char_p

    def close(self):
        self._h = C.c_void_p()


class Pool:
    """Multi-GPU behind the C ABI (tfhe_pool_*): one context per listed device, ONE BootstrappingKey
    (bootstrapping.rs:18-21) prepared once and replicated device to device, batches of independent bootstrap() calls
    (bootstrapping.rs:58-65) cut into contiguous slices, no collective in the data path.  A device may be listed
    more than once (several members on one GPU)."""

    def __init__(self, params: TfheParams, devices, backend: int = BACKEND_AUTO):
        self.params = params
        self.devices = [int(d) for d in devices]
        self._h = C.c_void_p()
        cp = params._c()
        arr = (C.c_int * len(self.devices))(*self.devices)
        lib().tfhe_pool_last_error.restype = C.c_char_p
        lib().tfhe_pool_member.restype = C.c_void_p
        lib().tfhe_pool_size.restype = C.c_size_t
        st = lib().tfhe_pool_create(C.byref(cp), arr, C.c_size_t(len(self.devices)), C.c_int(backend), C.byref(self._h))
        if st:
            self._h = C.c_void_p()
            raise TfheError(st, lib().tfhe_status_string(st).decode())

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().tfhe_pool_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, st: int):
        if st:
            raise TfheError(st, lib().tfhe_pool_last_error(self._h).decode())

    def __len__(self) -> int:
        return lib().tfhe_pool_size(self._h)

    def member(self, i: int) -> Context:
        """borrowed context of member i (timing, backend name); owned by the pool"""
        h = lib().tfhe_pool_member(self._h, C.c_size_t(i))
        if not h:
            raise IndexError(i)
        return _BorrowedContext(self.params, h)

    @property
    def backend(self) -> str:
        return self.member(0).backend

    def shard(self, batch: int, member: int):
        """(first row, row count) of the slice member `member` processes"""
        first, count = C.c_size_t(), C.c_size_t()
        self._check(lib().tfhe_pool_shard(self._h, C.c_size_t(batch), C.c_size_t(member), C.byref(first), C.byref(count)))
        return first.value, count.value

    def set_decomposer_alignment(self, aligned: bool):
        self._check(lib().tfhe_pool_set_decomposer_alignment(self._h, C.c_int(int(aligned))))

    def set_kernel_shape(self, shape: int):
        self._check(lib().tfhe_pool_set_kernel_shape(self._h, C.c_int(int(shape))))

    def set_bootstrap_order(self, ks_first: bool):
        self._check(lib().tfhe_pool_set_bootstrap_order(self._h, C.c_int(int(ks_first))))
        self._ks_first = bool(ks_first)

    @property
    def io_dim(self) -> int:
        return self.params.big_n if getattr(self, "_ks_first", False) else self.params.n

    def reserve(self, max_batch: int):
        self._check(lib().tfhe_pool_reserve(self._h, C.c_size_t(max_batch)))

    def synchronize(self):
        self._check(lib().tfhe_pool_synchronize(self._h))

    def load_bootstrapping_key(self, bsk, ksk):
        """bsk [n][R][k+1][N], ksk [k*N*l_ks][n+1]: numpy (host) or torch tensors on member 0's device.  Uploaded and
        transformed once; the prepared key is copied device to device to the other members."""
        p = self.params
        if _is_torch(bsk):
            assert tuple(bsk.shape) == p.bsk_shape() and tuple(ksk.shape) == p.ksk_shape()
            import torch
            torch.cuda.synchronize(bsk.device)
            self._check(lib().tfhe_pool_load_bootstrapping_key_device(self._h, _dp(bsk), _dp(ksk)))
            return
        bsk, ksk = _np(bsk), _np(ksk)
        assert bsk.shape == p.bsk_shape() and ksk.shape == p.ksk_shape()
        self._check(lib().tfhe_pool_load_bootstrapping_key(self._h, _hp(bsk), _hp(ksk)))

    def replicate_key(self):
        """copy whatever key member 0 holds (prepared form + KSK) to every other member, device to device"""
        self._check(lib().tfhe_pool_replicate_key(self._h))

    def bootstrapping_key_gen(self, lwe_sk, glwe_sk, bsk_samples, ksk_samples, load: bool = True, bmmp: bool = False):
        """bootstrapping_key_gen (bootstrapping.rs:23-56) on pre-filled host buffers -> (bsk, ksk); generated on member
        0's device and, with `load`, installed on EVERY member (tfhe_pool_bootstrapping_key_gen[_bmmp])."""
        p = self.params
        lsk, gsk = _np(lwe_sk).reshape(p.n), _np(glwe_sk).reshape(p.k, p.N)
        bsk, ksk = _np(bsk_samples).copy(), _np(ksk_samples).copy()
        assert bsk.shape == (p.bsk_bmmp_shape() if bmmp else p.bsk_shape()) and ksk.shape == p.ksk_shape()
        fn = lib().tfhe_pool_bootstrapping_key_gen_bmmp if bmmp else lib().tfhe_pool_bootstrapping_key_gen
        self._check(fn(self._h, _hp(lsk), _hp(gsk), _hp(bsk), _hp(ksk), C.c_int(int(load))))
        return bsk, ksk

    def generate_keys(self, rng=None, load: bool = True, bmmp: bool = False):
        """Context.generate_keys for the pool: secrets and samples drawn on the host (OS CSPRNG unless the test hook
        `rng=` is given), key material completed on member 0's GPU and, with `load`, installed on every member.
        -> (lwe_sk, glwe_sk, bsk, ksk)"""
        lwe_sk, glwe_sk, bsk, ksk = self.member(0).generate_keys(rng=rng, load=load, bmmp=bmmp)
        if load:
            self.replicate_key()
        return lwe_sk, glwe_sk, bsk, ksk

    def bootstrap(self, lwe_in, test_vector_poly) -> np.ndarray:
        """bootstrap() over a host batch [batch][n+1], sharded over the members"""
        lwe, tv = _np(lwe_in).reshape(-1, self.io_dim + 1), _np(test_vector_poly)
        tv_count = 1 if tv.ndim == 1 else tv.shape[0]
        res = np.zeros_like(lwe)
        self._check(lib().tfhe_pool_bootstrap_batch(self._h, _hp(lwe), C.c_size_t(lwe.shape[0]), _hp(tv),
                                                    C.c_size_t(tv_count), _hp(res)))
        return res

    def gate(self, truth, ct0, ct1) -> np.ndarray:
        """boolean.rs:9-53 over a host batch, sharded over the members"""
        c0, c1 = _np(ct0).reshape(-1, self.io_dim + 1), _np(ct1).reshape(-1, self.io_dim + 1)
        t = (C.c_uint32 * 4)(*[int(v) for v in truth])
        res = np.zeros_like(c0)
        self._check(lib().tfhe_pool_gate_batch(self._h, t, _hp(c0), _hp(c1), C.c_size_t(c0.shape[0]), _hp(res)))
        return res

    def bootstrap_shards(self, lwe_shards, tv_shards, out_shards):
        """Device-resident shards (torch tensors, shard i on member i's device; None or an empty shard skips the
        member): enqueues on every member's stream and returns -- synchronize() waits.
        Ordering (tfhe_hip.h, tfhe_pool_bootstrap_shards_device): a member's work runs on ITS stream, not on the torch
        stream that filled the shard, so this first waits for the torch current stream of every shard's device (a no-op
        when it is idle); the OUTPUTS are ready -- and the shard tensors may be freed or refilled -- only after
        synchronize()."""
        n = len(self)
        import torch
        for dev in {t.device for t in list(lwe_shards) + list(tv_shards) if t is not None and t.is_cuda}:
            torch.cuda.current_stream(dev).synchronize()
        assert len(lwe_shards) == n and len(tv_shards) == n and len(out_shards) == n
        ptrs_in, ptrs_tv, ptrs_out = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)()
        counts, tv_counts = (C.c_size_t * n)(), (C.c_size_t * n)()
        for i in range(n):
            rows = 0 if lwe_shards[i] is None else int(lwe_shards[i].shape[0])
            counts[i] = rows
            if rows == 0:
                continue
            ptrs_in[i], ptrs_tv[i], ptrs_out[i] = lwe_shards[i].data_ptr(), tv_shards[i].data_ptr(), out_shards[i].data_ptr()
            tv_counts[i] = 1 if tv_shards[i].dim() == 1 else int(tv_shards[i].shape[0])
        self._check(lib().tfhe_pool_bootstrap_shards_device(self._h, ptrs_in, counts, ptrs_tv, tv_counts, ptrs_out))
