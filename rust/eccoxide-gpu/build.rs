// Link against libeccx.so (built by `make -C eccoxide_amd/csrc`, hipcc --offload-arch=gfx950).
// ECCX_LIB_DIR overrides the directory; the default is this repository's eccoxide_amd/.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("ECCX_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../eccoxide_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=eccx");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=ECCX_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/eccx.h");
}
