// Points the linker at the built library: MPECDSA_HIP_LIB_DIR, or <repo>/multi_party_ecdsa_amd next to this crate.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("MPECDSA_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../../multi_party_ecdsa_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=mpecdsa_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=MPECDSA_HIP_LIB_DIR");
}
