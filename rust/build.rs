// build.rs of a needletail checkout with `--features amd` (NOT compiled here: no rustc in the image).
// Links the C-ABI library built by `make -C needletail_amd/csrc` (gfx950 only; no other backend exists).
fn main() {
    if std::env::var("CARGO_FEATURE_AMD").is_ok() {
        if let Ok(dir) = std::env::var("NEEDLETAIL_AMD_LIB_DIR") {
            println!("cargo:rustc-link-search=native={dir}");
            println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
        }
        println!("cargo:rustc-link-lib=dylib=needletail_amd");
        println!("cargo:rerun-if-env-changed=NEEDLETAIL_AMD_LIB_DIR");
    }
}
