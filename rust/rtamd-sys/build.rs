// Links against librtamd.so built by `make -C raytracer-rust_amd/csrc`.
// RTAMD_LIB_DIR must point at the directory that holds it.
fn main() {
    let dir = std::env::var("RTAMD_LIB_DIR").expect("set RTAMD_LIB_DIR to the directory of librtamd.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=rtamd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=RTAMD_LIB_DIR");
}
