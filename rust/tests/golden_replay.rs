//! Integration test of the shim crate (tfhe-hip-sys): the MI355X engine, called from Rust through
//! the C ABI, reproduces the committed golden fixtures (tests/golden/, tools/make_golden.py).
//! Needs a GPU and libtfhe_hip.so; SOURCE ONLY here (no Rust toolchain in the build image).
//!
//!     TFHE_GOLDEN_DIR=../tests/golden TFHE_HIP_LIB_DIR=../tfhe-research_amd cargo test --release
//!
//! The twin that pins the fixtures to the REFERENCE crate is rust/reference_patch/golden_replay.rs.
use ndarray::{Array1, Array2, Array4};
use std::{fs, path::PathBuf};
use tfhe_hip_sys::{CDecomposerParams, CTfheParams, GpuBootstrappingKey};

fn read(dir: &PathBuf, name: &str) -> ([u32; 12], Vec<usize>, Vec<u32>) {
    let b = fs::read(dir.join(format!("{name}.tfhe"))).unwrap();
    assert_eq!(&b[0..8], b"TFHEAMD\x01");
    let u32_at = |o: usize| u32::from_le_bytes(b[o..o + 4].try_into().unwrap());
    let mut params = [0u32; 12];
    for (i, p) in params.iter_mut().enumerate() {
        *p = u32_at(16 + 4 * i);
    }
    let ndims = u32_at(64) as usize;
    let dims: Vec<usize> = (0..ndims).map(|i| u32_at(68 + 4 * i) as usize).collect();
    let words: usize = dims.iter().product();
    assert_eq!(b.len(), 104 + 4 * words);
    (params, dims, (0..words).map(|i| u32_at(104 + 4 * i)).collect())
}

fn replay(set: &str) {
    let dir = PathBuf::from(std::env::var("TFHE_GOLDEN_DIR").expect("TFHE_GOLDEN_DIR")).join(set);
    let (p, bd, bsk) = read(&dir, "bsk");
    let (_, kd, ksk) = read(&dir, "ksk");
    let (_, ld, lwe_in) = read(&dir, "lwe_in");
    let (_, _, lwe_out) = read(&dir, "lwe_out");
    let (_, _, tv) = read(&dir, "tv");
    let params = CTfheParams {
        glwe_dimension: p[0], glwe_poly_degree: p[1], lwe_dimension: p[2], padding_bits: p[3], log_p: p[4], log_q: p[5],
        ks_decomposer: CDecomposerParams { log_base: p[6], levels: p[7], log_q: p[8] },
        pbs_decomposer: CDecomposerParams { log_base: p[9], levels: p[10], log_q: p[11] },
    };
    let bsk = Array4::from_shape_vec((bd[0], bd[1], bd[2], bd[3]), bsk).unwrap();
    let ksk = Array2::from_shape_vec((kd[0], kd[1]), ksk).unwrap();
    let key = GpuBootstrappingKey::upload_flat(&params, &[0], bsk.as_slice().unwrap(), ksk.as_slice().unwrap());
    let batch = Array2::from_shape_vec((ld[0], ld[1]), lwe_in).unwrap();
    let got = tfhe_hip_sys::bootstrap_batch(&key, &batch, &Array1::from_vec(tv));
    assert_eq!(got.as_slice().unwrap(), &lwe_out[..], "{set}");
}

#[test]
fn gpu_reproduces_ref_test() { replay("ref_test"); }

#[test]
fn gpu_reproduces_misaligned_base() { replay("misaligned"); }
