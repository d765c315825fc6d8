// UNVERIFIED (never compiled here).  Links libfrr_hip.so; FRR_LIB_DIR = directory holding it
// (default: ../../f_renderer_amd of this repository, where `python -c "import f_renderer_amd as fr; fr.build()"` puts it).
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("FRR_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../f_renderer_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=frr_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=FRR_LIB_DIR");
}
