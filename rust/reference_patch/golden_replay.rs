//! Replays tests/golden/ (this repository's frozen oracle outputs) through the REFERENCE CRATE
//! ITSELF, so that anyone with a Rust toolchain pins the oracle -- and with it every parity claim of
//! the MI355X engine -- to Janmajayamall/tfhe-research in one command:
//!
//!     cp  <this repo>/rust/reference_patch/golden_replay.rs   <tfhe-research>/src/golden_replay.rs
//!     (apply the four visibility edits below)
//!     TFHE_GOLDEN_DIR=<this repo>/tests/golden cargo test --release golden_replay -- --nocapture
//!
//! SOURCE ONLY: the image this repository is built in has no cargo/rustc, so this file has never
//! been compiled.  It is deliberately plain (std + ndarray, both already dependencies of the crate).
//!
//! Why it must live INSIDE the crate: every module of the reference is private (`mod bootstrapping;`
//! src/lib.rs:12-21), so an external integration test cannot name `bootstrap`.  A child module of
//! the crate root can, and it also sees TfheParams' private fields (src/lib.rs:23-34).  Four items
//! are private to sibling modules and need `pub(crate)`:
//!
//!   src/lib.rs               + #[cfg(test)] mod golden_replay;
//!   src/bootstrapping.rs:19  lwe_sk_ggsw_enc: Vec<GgswCiphertext>   ->  pub(crate) lwe_sk_ggsw_enc: ...
//!   src/bootstrapping.rs:20  ksk: KeySwitchingKey                   ->  pub(crate) ksk: ...
//!   src/bootstrapping.rs:122 fn sample_extract(                     ->  pub(crate) fn sample_extract(
//!   src/ggsw.rs:40           data: Array3<u32>                      ->  pub(crate) data: Array3<u32>
//!
//! `--release` is required: ndarray's `scaled_add` / `dot` on u32 panic on overflow in debug builds
//! (key_switching.rs:88, utils.rs:158), the crate's arithmetic is only total with wrapping on.
//!
//! File format (include/tfhe_hip.h "on-disk format"): magic "TFHEAMD\1", u32 kind, u32 flags,
//! u32[12] params (k, log2 N, n, padding_bits, log_p, log_q, ks{log_base, levels, log_q},
//! pbs{log_base, levels, log_q}), u32 ndims, u32[4] dims, u32 0, u64 words, u64 FNV-1a-64 of the
//! payload, then `words` little-endian u32.
use crate::{
    bootstrapping::{bootstrap, sample_extract, BootstrappingKey},
    decomposer::{DecomposerParams, SignedDecomposer},
    ggsw::{cmux, GgswCiphertext},
    glwe::{trivial_encrypt_glwe_plaintext, GlweCiphertext, GlweCleartext, GlweSecretKey, Monomial},
    key_switching::{key_switch_lwe, KeySwitchingKey},
    lwe::{LweCiphertext, LweSecretKey},
    utils::switch_modulus,
    TfheParams,
};
use ndarray::{Array1, Array2, Array3};
use std::{fs, path::PathBuf};

struct Golden {
    params: [u32; 12],
    dims: Vec<usize>,
    data: Vec<u32>,
}

fn read_golden(dir: &PathBuf, name: &str) -> Golden {
    let path = dir.join(format!("{name}.tfhe"));
    let b = fs::read(&path).unwrap_or_else(|e| panic!("{}: {e}", path.display()));
    assert_eq!(&b[0..8], b"TFHEAMD\x01", "{}: bad magic", path.display());
    let u32_at = |o: usize| u32::from_le_bytes(b[o..o + 4].try_into().unwrap());
    let u64_at = |o: usize| u64::from_le_bytes(b[o..o + 8].try_into().unwrap());
    let mut params = [0u32; 12];
    for (i, p) in params.iter_mut().enumerate() {
        *p = u32_at(16 + 4 * i);
    }
    let ndims = u32_at(64) as usize;
    let dims: Vec<usize> = (0..ndims).map(|i| u32_at(68 + 4 * i) as usize).collect();
    let words = u64_at(88) as usize;
    assert_eq!(dims.iter().product::<usize>(), words);
    assert_eq!(b.len(), 104 + 4 * words, "{}: truncated or padded", path.display());
    let mut h: u64 = 0xcbf29ce484222325;
    for byte in &b[104..] {
        h = (h ^ *byte as u64).wrapping_mul(0x100000001b3);
    }
    assert_eq!(h, u64_at(96), "{}: checksum", path.display());
    let data = (0..words).map(|i| u32_at(104 + 4 * i)).collect();
    Golden { params, dims, data }
}

fn params_from(p: &[u32; 12]) -> TfheParams {
    TfheParams {
        glwe_dimension: p[0] as usize,
        glwe_poly_degree: p[1] as usize,
        lwe_dimension: p[2] as usize,
        padding_bits: p[3] as usize,
        log_p: p[4] as usize,
        log_q: p[5] as usize,
        ks_decomposer: DecomposerParams { log_base: p[6] as usize, levels: p[7] as usize, log_q: p[8] as usize },
        pbs_decomposer: DecomposerParams { log_base: p[9] as usize, levels: p[10] as usize, log_q: p[11] as usize },
        // noise parameters play no part in bootstrap(): lib.rs:96-97 defaults
        lwe_std_dev: 0.000013071021089943935,
        glwe_std_dev: 0.00000004990272175010415,
    }
}

fn replay_set(set: &str) {
    let root = PathBuf::from(std::env::var("TFHE_GOLDEN_DIR").expect("set TFHE_GOLDEN_DIR to <repo>/tests/golden"));
    let dir = root.join(set);
    let bsk = read_golden(&dir, "bsk"); // [n][(k+1)l][k+1][N]
    let ksk = read_golden(&dir, "ksk"); // [kN*l_ks][n+1]
    let lwe_in = read_golden(&dir, "lwe_in"); // [rows][n+1]
    let lwe_out = read_golden(&dir, "lwe_out");
    let tv = read_golden(&dir, "tv"); // [N]
    let approx = read_golden(&dir, "approximate_lwe"); // [rows][n+1]
    let acc_init = read_golden(&dir, "acc_init"); // [rows][k+1][N]
    let acc_each = read_golden(&dir, "acc_after_each"); // [rows*n][k+1][N]
    let acc_final = read_golden(&dir, "acc_final");
    let extracted = read_golden(&dir, "extracted_lwe"); // [rows][kN+1]

    let tfhe_params = params_from(&bsk.params);
    let (n, rows_ggsw, k1, big_n) = (bsk.dims[0], bsk.dims[1], bsk.dims[2], bsk.dims[3]);
    assert_eq!(n, tfhe_params.lwe_dimension);
    let ggsw_words = rows_ggsw * k1 * big_n;
    let bootstrapping_key = BootstrappingKey {
        lwe_sk_ggsw_enc: (0..n)
            .map(|i| GgswCiphertext {
                data: Array3::from_shape_vec((rows_ggsw, k1, big_n), bsk.data[i * ggsw_words..(i + 1) * ggsw_words].to_vec()).unwrap(),
            })
            .collect(),
        ksk: KeySwitchingKey { data: Array2::from_shape_vec((ksk.dims[0], ksk.dims[1]), ksk.data.clone()).unwrap() },
    };
    // bootstrap() takes the two secret keys and never reads them (bootstrapping.rs:61-62)
    let dummy_lwe_sk = LweSecretKey { data: Array1::zeros(n) };
    let dummy_glwe_sk = GlweSecretKey { data: Array2::zeros((k1 - 1, big_n)) };
    let test_vector_poly = Array1::from_vec(tv.data.clone());
    let glwe_params = tfhe_params.glwe_params();
    let ggsw_params = tfhe_params.ggsw_params();
    let rows = lwe_in.dims[0];
    let glwe_words = k1 * big_n;

    for r in 0..rows {
        let ct = LweCiphertext { data: Array1::from_vec(lwe_in.data[r * (n + 1)..(r + 1) * (n + 1)].to_vec()) };
        // (1) the whole path: bootstrapping.rs:58-120
        let out = bootstrap(&tfhe_params, &ct, &dummy_lwe_sk, &dummy_glwe_sk, &bootstrapping_key, &test_vector_poly);
        assert_eq!(out.data.as_slice().unwrap(), &lwe_out.data[r * (n + 1)..(r + 1) * (n + 1)], "{set}: lwe_out row {r}");

        // (2) stage by stage with the crate's own functions, to localise a mismatch
        let a = switch_modulus(ct.data.as_slice().unwrap(), tfhe_params.log_q, tfhe_params.glwe_poly_degree + 1);
        assert_eq!(&a[..], &approx.data[r * (n + 1)..(r + 1) * (n + 1)], "{set}: switch_modulus row {r}");
        let v_x = trivial_encrypt_glwe_plaintext(&glwe_params, &GlweCleartext::encode_message(test_vector_poly.as_slice().unwrap(), &glwe_params));
        let mut acc: GlweCiphertext = &v_x * &Monomial { index: -(a[n] as isize) };
        assert_eq!(acc.data.as_slice().unwrap(), &acc_init.data[r * glwe_words..(r + 1) * glwe_words], "{set}: acc_init row {r}");
        for i in 0..n {
            let mut c1 = &acc * &Monomial { index: a[i] as isize };
            acc = cmux(&ggsw_params, &bootstrapping_key.lwe_sk_ggsw_enc[i], &acc, &mut c1);
            let at = (r * n + i) * glwe_words;
            assert_eq!(acc.data.as_slice().unwrap(), &acc_each.data[at..at + glwe_words], "{set}: acc after CMUX {i} row {r}");
        }
        assert_eq!(acc.data.as_slice().unwrap(), &acc_final.data[r * glwe_words..(r + 1) * glwe_words], "{set}: acc_final row {r}");
        let ext = sample_extract(&acc, &glwe_params, 0);
        let ew = (k1 - 1) * big_n + 1;
        assert_eq!(ext.data.as_slice().unwrap(), &extracted.data[r * ew..(r + 1) * ew], "{set}: sample_extract row {r}");
        let ks = key_switch_lwe(&ext, &tfhe_params.lwe_params_post_pbs(), &tfhe_params.lwe_params(),
                                &SignedDecomposer::new(tfhe_params.ks_decomposer.clone()), &bootstrapping_key.ksk);
        assert_eq!(ks.data.as_slice().unwrap(), &lwe_out.data[r * (n + 1)..(r + 1) * (n + 1)], "{set}: key switch row {r}");
    }
    println!("golden set {set}: {rows} rows reproduce bit for bit through the reference crate");
}

#[test]
fn golden_replay_ref_test() {
    replay_set("ref_test");
    // rows 0..3 encrypt the messages 0..3 under the committed key; identity LUT
    let root = PathBuf::from(std::env::var("TFHE_GOLDEN_DIR").unwrap()).join("ref_test");
    let sk = read_golden(&root, "lwe_sk");
    let out = read_golden(&root, "lwe_out");
    let p = params_from(&out.params);
    let n = p.lwe_dimension;
    // phase = b - <a, s> in wrapping u32 (lwe.rs:162-173), then ROUND to the nearest message slot: the
    // fixtures' noise is two-sided, and LwePlaintext::decode (lwe.rs:102-107) truncates
    let shift = 32 - (p.log_p + p.padding_bits);
    for m in 0..4usize {
        let ct = &out.data[m * (n + 1)..(m + 1) * (n + 1)];
        let a_s = ct[..n].iter().zip(sk.data.iter()).fold(0u32, |acc, (a, s)| acc.wrapping_add(a.wrapping_mul(*s)));
        let phase = ct[n].wrapping_sub(a_s);
        let got = (phase.wrapping_add(1u32 << (shift - 1)) >> shift) & ((1u32 << p.log_p) - 1);
        assert_eq!(got, m as u32, "row {m} of ref_test decrypts to its message");
    }
}

#[test]
fn golden_replay_misaligned_base() {
    // log_base = 7 does not divide 32: pins decomposer.rs:48-70 (limbs counted from bit 0)
    replay_set("misaligned");
}

#[test]
fn golden_replay_n1024_full_word() {
    // N = 1024 with l = 4 levels of 2^8: the whole word is decomposed, so (unlike the set above, whose trivial
    // accumulator has only zero digits) every CMUX of this trace depends on the key
    replay_set("n1024_full_word");
}
