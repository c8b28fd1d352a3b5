// Link against libvrod_hip.so; VROD_HIP_LIB_DIR = the directory holding it (vrod_amd/ in this repo).
fn main() {
    let dir = std::env::var("VROD_HIP_LIB_DIR").expect("set VROD_HIP_LIB_DIR to the directory of libvrod_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=vrod_hip");
    println!("cargo:rerun-if-env-changed=VROD_HIP_LIB_DIR");
}
