// Links libgpuq.so (built by `python arrow-ballista_amd/build.py`).  GPUQ_LIB_DIR points at the directory holding it.
fn main() {
    let dir = std::env::var("GPUQ_LIB_DIR").unwrap_or_else(|_| "../arrow-ballista_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=gpuq");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=GPUQ_LIB_DIR");
}
