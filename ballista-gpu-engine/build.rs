// 1. Links libgpuq.so (built by `python arrow-ballista_amd/build.py`).  GPUQ_LIB_DIR points at the directory holding it.
// 2. Generates the binary's option struct from the stock executor's spec (ballista/executor/build.rs does the same for its own
//    binary): without this step OUT_DIR/executor_configure_me_config.rs does not exist in THIS crate's OUT_DIR and
//    src/bin/gpu_executor.rs cannot include it.
extern crate configure_me_codegen;

fn main() -> Result<(), String> {
    let dir = std::env::var("GPUQ_LIB_DIR").unwrap_or_else(|_| "../arrow-ballista_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=gpuq");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=GPUQ_LIB_DIR");
    println!("cargo:rerun-if-changed=../ballista/executor/executor_config_spec.toml");
    configure_me_codegen::build_script_auto().map_err(|e| format!("configure_me code generation failed: {e}"))
}
