// Points the linker at libeigentrust_hip.so.  EIGENTRUST_HIP_LIB_DIR overrides the in-tree
// location (protocol_amd/lib, where `make -C protocol_amd/csrc` puts the library).
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("EIGENTRUST_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../../protocol_amd/lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=eigentrust_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=EIGENTRUST_HIP_LIB_DIR");
}
