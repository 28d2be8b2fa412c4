// Points rustc at libbiogpu.so (built by `make -C rust-bio_amd/csrc`).  BIOGPU_LIB_DIR overrides the in-tree location.
fn main() {
    let dir = std::env::var("BIOGPU_LIB_DIR").unwrap_or_else(|_| {
        let here = std::path::PathBuf::from(std::env::var("CARGO_MANIFEST_DIR").unwrap());
        here.join("../../rust-bio_amd").to_string_lossy().into_owned()
    });
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=biogpu");
    println!("cargo:rerun-if-env-changed=BIOGPU_LIB_DIR");
}
