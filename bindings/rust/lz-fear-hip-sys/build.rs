// Points rustc at the directory holding liblzfear_hip.so (built by `python __graft_entry__.py`).
fn main() {
    let dir = std::env::var("LZFEAR_HIP_LIB_DIR").unwrap_or_else(|_| "../../../rust-lz-fear_amd".into());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=lzfear_hip");
}
