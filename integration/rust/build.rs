// build.rs -- link the reference crate against libadsb_hip.so.
// ADSB_HIP_DIR = directory holding libadsb_hip.so (…/dump1090_rs_amd in this repo).
fn main() {
    let dir = std::env::var("ADSB_HIP_DIR").expect("set ADSB_HIP_DIR to the directory of libadsb_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=adsb_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=ADSB_HIP_DIR");
}
