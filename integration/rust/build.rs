// build.rs -- link the reference crate against libadsb_hip.so, for the `hip` feature only.
// ADSB_HIP_DIR = directory holding libadsb_hip.so (…/dump1090_rs_amd in this repo).
fn main() {
    println!("cargo:rerun-if-env-changed=ADSB_HIP_DIR");
    // Without the feature this script does nothing: the reference's default `cargo build`,
    // `cargo test` and `cross test` (no GPU, no ADSB_HIP_DIR) keep working with the file in place.
    if std::env::var_os("CARGO_FEATURE_HIP").is_none() {
        return;
    }
    let dir = std::env::var("ADSB_HIP_DIR")
        .expect("--features hip: set ADSB_HIP_DIR to the directory of libadsb_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=adsb_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
}
