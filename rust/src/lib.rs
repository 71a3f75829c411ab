//! Rust-side binding a maintainer of Janmajayamall/tfhe-research would add to call the MI355X
//! engine from the crate, keeping the crate-internal signatures of the bootstrapping path.
//! SOURCE ONLY (never compiled here: no Rust toolchain in the image).  The C ABI is
//! `include/tfhe_hip.h`; everything below is plumbing between ndarray buffers and raw pointers.
//!
//! Reference signatures mirrored (file:line under src/):
//!   bootstrap            bootstrapping.rs:58-65
//!   key_switch_lwe       key_switching.rs:63-69
//!   external_product     ggsw.rs:132-136        cmux   ggsw.rs:164-169
//!   and / or             boolean.rs:9-14, 32-37
//!   construct_test_from_lut  test_vector.rs:38
//!   bootstrapping_key_gen    bootstrapping.rs:23-28   encrypt_lwe_plaintext / decrypt_lwe  lwe.rs:138-173
use ndarray::{Array1, Array2, Array3};
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
#[derive(Clone, Copy)]
pub struct CDecomposerParams { pub log_base: u32, pub levels: u32, pub log_q: u32 }

#[repr(C)]
#[derive(Clone, Copy)]
pub struct CTfheParams {
    pub glwe_dimension: u32, pub glwe_poly_degree: u32, pub lwe_dimension: u32,
    pub padding_bits: u32, pub log_p: u32, pub log_q: u32,
    pub ks_decomposer: CDecomposerParams, pub pbs_decomposer: CDecomposerParams,
}

#[repr(C)] pub struct TfheContext { _private: [u8; 0] }
#[repr(C)] pub struct TfhePool { _private: [u8; 0] }

extern "C" {
    // multi-GPU pool (include/tfhe_hip.h, "multi-GPU pool"): one context per listed HIP device, ONE key prepared
    // once and replicated device to device, batches cut into contiguous slices
    fn tfhe_pool_create(params: *const CTfheParams, devices: *const c_int, n_devices: usize, backend: c_int,
                        out: *mut *mut TfhePool) -> c_int;
    fn tfhe_pool_destroy(pool: *mut TfhePool);
    fn tfhe_pool_member(pool: *mut TfhePool, i: usize) -> *mut TfheContext;
    fn tfhe_pool_last_error(pool: *const TfhePool) -> *const c_char;
    fn tfhe_pool_load_bootstrapping_key(pool: *mut TfhePool, bsk: *const u32, ksk: *const u32) -> c_int;
    fn tfhe_pool_load_bootstrapping_key_bmmp(pool: *mut TfhePool, bsk_bmmp: *const u32, ksk: *const u32) -> c_int;
    fn tfhe_pool_bootstrap_batch(pool: *mut TfhePool, lwe_in: *const u32, batch: usize,
                                 test_vector_poly: *const u32, tv_count: usize, lwe_out: *mut u32) -> c_int;
    fn tfhe_pool_gate_batch(pool: *mut TfhePool, truth: *const u32, ct0: *const u32, ct1: *const u32,
                            batch: usize, lwe_out: *mut u32) -> c_int;
    fn tfhe_pool_set_kernel_shape(pool: *mut TfhePool, shape: c_int) -> c_int;
    fn tfhe_last_error(ctx: *const TfheContext) -> *const c_char;
    fn tfhe_bootstrap_batch(ctx: *mut TfheContext, lwe_in: *const u32, batch: usize,
                            test_vector_poly: *const u32, tv_count: usize, lwe_out: *mut u32) -> c_int;
    fn tfhe_key_switch_batch(ctx: *mut TfheContext, lwe_in: *const u32, batch: usize, lwe_out: *mut u32) -> c_int;
    fn tfhe_external_product_batch(ctx: *mut TfheContext, ggsw: *const u32, ggsw_count: usize,
                                   glwe_in: *const u32, batch: usize, glwe_out: *mut u32) -> c_int;
    fn tfhe_cmux_batch(ctx: *mut TfheContext, ggsw: *const u32, ggsw_count: usize, ct0: *const u32,
                       ct1: *mut u32, batch: usize, glwe_out: *mut u32) -> c_int;
    fn tfhe_gate_batch(ctx: *mut TfheContext, truth: *const u32, ct0: *const u32, ct1: *const u32,
                       batch: usize, lwe_out: *mut u32) -> c_int;
    fn tfhe_construct_test_from_lut(params: *const CTfheParams, lut: *const u32, lut_len: usize, out: *mut u32) -> c_int;
    #[allow(dead_code)]
    fn tfhe_context_set_stream(ctx: *mut TfheContext, hip_stream: *mut c_void) -> c_int;
    fn tfhe_bootstrapping_key_gen(ctx: *mut TfheContext, lwe_sk: *const u32, glwe_sk: *const u32,
                                  bsk: *mut u32, ksk: *mut u32, load: c_int) -> c_int;
    fn tfhe_lwe_encrypt_batch(ctx: *mut TfheContext, lwe_sk: *const u32, dimension: usize,
                              plaintexts: *const u32, lwe: *mut u32, batch: usize) -> c_int;
    fn tfhe_lwe_decrypt_batch(ctx: *mut TfheContext, lwe_sk: *const u32, dimension: usize,
                              lwe: *const u32, batch: usize, plaintext_out: *mut u32) -> c_int;
}

/// The reference panics on failure (assert!/unwrap); so does this shim -- but only on the Rust
/// side of the boundary: the C ABI itself returns status codes and never unwinds.
fn check(ctx: *const TfheContext, status: c_int, what: &str) {
    if status != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(tfhe_last_error(ctx)) }.to_string_lossy().into_owned();
        panic!("{what}: status {status}: {msg}");
    }
}

fn check_pool(pool: *const TfhePool, status: c_int, what: &str) {
    if status != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(tfhe_pool_last_error(pool)) }.to_string_lossy().into_owned();
        panic!("{what}: status {status}: {msg}");
    }
}

/// Owns the device copies of ONE `BootstrappingKey` (bootstrapping.rs:18-21) on one or several GPUs of a node.
/// `pool` spans the listed devices; `ctx` is its first member (borrowed) and serves the single-ciphertext calls.
pub struct GpuBootstrappingKey { pool: *mut TfhePool, ctx: *mut TfheContext, params: CTfheParams }

/// All GPUs of an 8-GPU node: `&ALL_8_GPUS` as the `devices` argument below; `&[0]` is one GPU.
pub const ALL_8_GPUS: [c_int; 8] = [0, 1, 2, 3, 4, 5, 6, 7];

impl GpuBootstrappingKey {
    fn create(params: &CTfheParams, devices: &[c_int]) -> (*mut TfhePool, *mut TfheContext) {
        let mut pool = std::ptr::null_mut();
        let st = unsafe { tfhe_pool_create(params, devices.as_ptr(), devices.len(), 0 /* TFHE_BACKEND_AUTO */, &mut pool) };
        assert!(st == 0, "tfhe_pool_create: status {st}");
        (pool, unsafe { tfhe_pool_member(pool, 0) })
    }

    /// `lwe_sk_ggsw_enc`: the n `GgswCiphertext.data` arrays ((k+1)l, k+1, N); `ksk`: `KeySwitchingKey.data`;
    /// `devices`: HIP device ordinals.  The key crosses PCIe once and is transformed once; the other devices get
    /// the prepared key over xGMI.
    pub fn upload(params: CTfheParams, devices: &[c_int], lwe_sk_ggsw_enc: &[Array3<u32>], ksk: &Array2<u32>) -> Self {
        let mut flat = Vec::new();
        for g in lwe_sk_ggsw_enc { flat.extend_from_slice(g.as_slice().unwrap()); }
        Self::upload_flat(&params, devices, &flat, ksk.as_slice().unwrap())
    }

    /// The same from already flattened buffers: `bsk` [n][(k+1)l][k+1][N], `ksk` [kN*l_ks][n+1]
    /// (the layout of the on-disk format and of tests/golden/).
    pub fn upload_flat(params: &CTfheParams, devices: &[c_int], bsk: &[u32], ksk: &[u32]) -> Self {
        let (pool, ctx) = Self::create(params, devices);
        check_pool(pool, unsafe { tfhe_pool_load_bootstrapping_key(pool, bsk.as_ptr(), ksk.as_ptr()) }, "load key");
        GpuBootstrappingKey { pool, ctx, params: *params }
    }

    /// Key of the unrolled blind rotation the crate sketches in notes/BMMP Bootstrapping.md:13-25:
    /// `bsk_bmmp` [n/2][3][(k+1)l][k+1][N] = GGSW(s s'), GGSW(s (1-s')), GGSW(s' (1-s)) per pair of key
    /// bits.  `bootstrap` / the gates then consume two key bits per step (N = 512, even n).
    pub fn upload_bmmp_flat(params: &CTfheParams, devices: &[c_int], bsk_bmmp: &[u32], ksk: &[u32]) -> Self {
        let (pool, ctx) = Self::create(params, devices);
        check_pool(pool, unsafe { tfhe_pool_load_bootstrapping_key_bmmp(pool, bsk_bmmp.as_ptr(), ksk.as_ptr()) }, "load BMMP key");
        GpuBootstrappingKey { pool, ctx, params: *params }
    }
}

/// Kernel shape of the blind rotation (`tfhe_context_set_kernel_shape`): `Auto` picks by batch -- a single `bootstrap()`
/// (the crate's own call, bootstrapping.rs:58-65) or a narrow gate level gets the wide team, large batches the throughput
/// kernels; the bits are the same.
#[derive(Clone, Copy)]
pub enum KernelShape { Auto = 0, Wide = 1, Team = 2 }

impl GpuBootstrappingKey {
    pub fn set_kernel_shape(&self, shape: KernelShape) {
        check_pool(self.pool, unsafe { tfhe_pool_set_kernel_shape(self.pool, shape as c_int) }, "set kernel shape");
    }
}

impl Drop for GpuBootstrappingKey {
    fn drop(&mut self) { unsafe { tfhe_pool_destroy(self.pool) } }  // destroys the member contexts too
}

/// bootstrapping.rs:58-65.  `_lwe_secret_key` / `_glwe_secret_key` keep the positional slots of the
/// reference signature; the reference never reads them (bootstrapping.rs:66-120) and neither do we.
pub fn bootstrap<SK1, SK2>(bk: &GpuBootstrappingKey, lwe_ciphertext: &Array1<u32>, _lwe_secret_key: &SK1,
                           _glwe_secret_key: &SK2, test_vector_poly: &Array1<u32>) -> Array1<u32> {
    let mut out = Array1::<u32>::zeros(lwe_ciphertext.len());
    check(bk.ctx, unsafe {
        tfhe_bootstrap_batch(bk.ctx, lwe_ciphertext.as_slice().unwrap().as_ptr(), 1,
                             test_vector_poly.as_slice().unwrap().as_ptr(), 1, out.as_slice_mut().unwrap().as_mut_ptr())
    }, "bootstrap");
    out
}

/// Batched form: `lwe_ciphertexts` is (batch, n+1) row-major.  Independent bootstraps: the batch is cut into
/// contiguous slices over the devices the key was uploaded to (8 GPUs: batch 2^20 -> 2^17 per GPU); no collective.
pub fn bootstrap_batch(bk: &GpuBootstrappingKey, lwe_ciphertexts: &Array2<u32>, test_vector_poly: &Array1<u32>) -> Array2<u32> {
    let mut out = Array2::<u32>::zeros(lwe_ciphertexts.raw_dim());
    check_pool(bk.pool, unsafe {
        tfhe_pool_bootstrap_batch(bk.pool, lwe_ciphertexts.as_slice().unwrap().as_ptr(), lwe_ciphertexts.nrows(),
                                  test_vector_poly.as_slice().unwrap().as_ptr(), 1, out.as_slice_mut().unwrap().as_mut_ptr())
    }, "bootstrap_batch");
    out
}

/// A stream of gates with one truth table `truth[(lhs << 1) | rhs]` (boolean.rs:9-53 per row), sharded like
/// bootstrap_batch: `ct0`, `ct1` are (batch, n+1).
pub fn gate_batch(bk: &GpuBootstrappingKey, truth: [u32; 4], ct0: &Array2<u32>, ct1: &Array2<u32>) -> Array2<u32> {
    let mut out = Array2::<u32>::zeros(ct0.raw_dim());
    check_pool(bk.pool, unsafe {
        tfhe_pool_gate_batch(bk.pool, truth.as_ptr(), ct0.as_slice().unwrap().as_ptr(), ct1.as_slice().unwrap().as_ptr(),
                             ct0.nrows(), out.as_slice_mut().unwrap().as_mut_ptr())
    }, "gate_batch");
    out
}

/// key_switching.rs:63-69 (parameters and key come from the uploaded BootstrappingKey)
pub fn key_switch_lwe(bk: &GpuBootstrappingKey, lwe_ciphertext: &Array1<u32>) -> Array1<u32> {
    let mut out = Array1::<u32>::zeros(bk.params.lwe_dimension as usize + 1);
    check(bk.ctx, unsafe { tfhe_key_switch_batch(bk.ctx, lwe_ciphertext.as_slice().unwrap().as_ptr(), 1, out.as_slice_mut().unwrap().as_mut_ptr()) }, "key_switch_lwe");
    out
}

/// ggsw.rs:132-136
pub fn external_product(bk: &GpuBootstrappingKey, ggsw_ciphertext: &Array3<u32>, glwe_ciphertext: &Array2<u32>) -> Array2<u32> {
    let mut out = Array2::<u32>::zeros(glwe_ciphertext.raw_dim());
    check(bk.ctx, unsafe {
        tfhe_external_product_batch(bk.ctx, ggsw_ciphertext.as_slice().unwrap().as_ptr(), 1,
                                    glwe_ciphertext.as_slice().unwrap().as_ptr(), 1, out.as_slice_mut().unwrap().as_mut_ptr())
    }, "external_product");
    out
}

/// ggsw.rs:164-169: `glwe_ciphertext1` is clobbered with ct1 - ct0, as in the reference.
pub fn cmux(bk: &GpuBootstrappingKey, ggsw_ciphertext: &Array3<u32>, glwe_ciphertext0: &Array2<u32>, glwe_ciphertext1: &mut Array2<u32>) -> Array2<u32> {
    let mut out = Array2::<u32>::zeros(glwe_ciphertext0.raw_dim());
    check(bk.ctx, unsafe {
        tfhe_cmux_batch(bk.ctx, ggsw_ciphertext.as_slice().unwrap().as_ptr(), 1, glwe_ciphertext0.as_slice().unwrap().as_ptr(),
                        glwe_ciphertext1.as_slice_mut().unwrap().as_mut_ptr(), 1, out.as_slice_mut().unwrap().as_mut_ptr())
    }, "cmux");
    out
}

fn gate(bk: &GpuBootstrappingKey, truth: [u32; 4], ct0: &Array1<u32>, ct1: &Array1<u32>) -> Array1<u32> {
    let mut out = Array1::<u32>::zeros(ct0.len());
    check(bk.ctx, unsafe { tfhe_gate_batch(bk.ctx, truth.as_ptr(), ct0.as_slice().unwrap().as_ptr(), ct1.as_slice().unwrap().as_ptr(), 1, out.as_slice_mut().unwrap().as_mut_ptr()) }, "gate");
    out
}
/// boolean.rs:9-14
pub fn and(bk: &GpuBootstrappingKey, ct0: &Array1<u32>, ct1: &Array1<u32>) -> Array1<u32> { gate(bk, [0, 0, 0, 1], ct0, ct1) }
/// boolean.rs:32-37
pub fn or(bk: &GpuBootstrappingKey, ct0: &Array1<u32>, ct1: &Array1<u32>) -> Array1<u32> { gate(bk, [0, 1, 1, 1], ct0, ct1) }
/// not in the reference; built through its closure hook construct_test_vector_boolean (test_vector.rs:5)
pub fn nand(bk: &GpuBootstrappingKey, ct0: &Array1<u32>, ct1: &Array1<u32>) -> Array1<u32> { gate(bk, [1, 1, 1, 0], ct0, ct1) }

/// test_vector.rs:38
pub fn construct_test_from_lut(params: &CTfheParams, lut: &[u32]) -> Array1<u32> {
    let mut out = Array1::<u32>::zeros(1usize << params.glwe_poly_degree);
    let st = unsafe { tfhe_construct_test_from_lut(params, lut.as_ptr(), lut.len(), out.as_slice_mut().unwrap().as_mut_ptr()) };
    assert!(st == 0, "lut must hold 2^log_p entries (test_vector.rs:41)");
    out
}

/// bootstrapping_key_gen (bootstrapping.rs:23-28) on the GPU.  The crate keeps drawing the
/// randomness with its own `rng` -- `bsk_samples` ((n, (k+1)l, k+1, N): `sample_uniform_array`
/// masks, `sample_gaussian_array` errors in every body row, glwe.rs:195,200) and `ksk_samples`
/// ((kN*l_ks, n+1): masks + the error in the b slot, lwe.rs:122-126) -- and the engine turns the
/// buffers into the key in place and installs it.
pub fn bootstrapping_key_gen(params: CTfheParams, lwe_secret_key: &Array1<u32>, glwe_secret_key: &Array2<u32>,
                             mut bsk_samples: ndarray::Array4<u32>, mut ksk_samples: Array2<u32>)
                             -> (GpuBootstrappingKey, ndarray::Array4<u32>, Array2<u32>) {
    // key generation runs on one GPU; upload_flat() the returned arrays to spread the key over more
    let (pool, ctx) = GpuBootstrappingKey::create(&params, &[0]);
    check(ctx, unsafe {
        tfhe_bootstrapping_key_gen(ctx, lwe_secret_key.as_slice().unwrap().as_ptr(),
                                   glwe_secret_key.as_slice().unwrap().as_ptr(),
                                   bsk_samples.as_slice_mut().unwrap().as_mut_ptr(),
                                   ksk_samples.as_slice_mut().unwrap().as_mut_ptr(), 1)
    }, "bootstrapping_key_gen");
    (GpuBootstrappingKey { pool, ctx, params }, bsk_samples, ksk_samples)
}

/// encrypt_lwe_plaintext (lwe.rs:138-160) over a batch: `samples` (batch, n+1) holds the uniform
/// masks and, in the b slot, the error; `plaintexts` are encoded (lwe.rs:81-92).
pub fn encrypt_lwe_batch(bk: &GpuBootstrappingKey, sk: &Array1<u32>, plaintexts: &Array1<u32>, mut samples: Array2<u32>) -> Array2<u32> {
    let batch = samples.nrows();
    check(bk.ctx, unsafe {
        tfhe_lwe_encrypt_batch(bk.ctx, sk.as_slice().unwrap().as_ptr(), sk.len(), plaintexts.as_slice().unwrap().as_ptr(),
                               samples.as_slice_mut().unwrap().as_mut_ptr(), batch)
    }, "encrypt_lwe_batch");
    samples
}

/// decrypt_lwe (lwe.rs:162-173) over a batch -> encoded plaintexts
pub fn decrypt_lwe_batch(bk: &GpuBootstrappingKey, sk: &Array1<u32>, cts: &Array2<u32>) -> Array1<u32> {
    let mut out = Array1::<u32>::zeros(cts.nrows());
    check(bk.ctx, unsafe {
        tfhe_lwe_decrypt_batch(bk.ctx, sk.as_slice().unwrap().as_ptr(), sk.len(), cts.as_slice().unwrap().as_ptr(),
                               cts.nrows(), out.as_slice_mut().unwrap().as_mut_ptr())
    }, "decrypt_lwe_batch");
    out
}
