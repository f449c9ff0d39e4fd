// Links libprovekit_hip.so (built by `make -C provekit_amd/csrc`, hipcc --offload-arch=gfx950).
fn main() {
    println!("cargo:rerun-if-env-changed=PROVEKIT_HIP_LIB_DIR");
    let dir = std::env::var("PROVEKIT_HIP_LIB_DIR")
        .expect("set PROVEKIT_HIP_LIB_DIR to the directory that holds libprovekit_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=provekit_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
}
