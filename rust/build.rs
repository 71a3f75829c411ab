// Links libtfhe_hip.so built by `python __graft_entry__.py` (hipcc, gfx950).
fn main() {
    let dir = std::env::var("TFHE_HIP_LIB_DIR").unwrap_or_else(|_| "../tfhe-research_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=tfhe_hip");
}
