// Links libltxhip.so.  LTXHIP_LIB_DIR = directory holding the library built by `make -C candle-video_amd`
// (defaults to ../../candle-video_amd relative to this crate, i.e. this repository's layout).
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("LTXHIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../candle-video_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=ltxhip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=LTXHIP_LIB_DIR");
}
