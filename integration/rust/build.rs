//! Links `libradiorust_amd.so` (the MI355X backend's C ABI, `include/radiorust_amd.h`).
//!
//! `RADIORUST_AMD_LIB_DIR` = the directory that holds the shared object (`radiorust_amd/lib` of the backend's
//! repository after `python -c "import __graft_entry__ as g; g.build()"`).  The library itself links the HIP
//! runtime; nothing else is needed at build time.
fn main() {
    println!("cargo:rerun-if-env-changed=RADIORUST_AMD_LIB_DIR");
    let dir = std::env::var("RADIORUST_AMD_LIB_DIR")
        .expect("set RADIORUST_AMD_LIB_DIR to the directory that holds libradiorust_amd.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=radiorust_amd");
    // so that the test binaries and examples find the library without LD_LIBRARY_PATH
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
}
