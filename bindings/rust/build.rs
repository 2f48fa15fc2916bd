// build.rs -- link liblpipm.so when the `hip` feature is on.
// LPIPM_LIB_DIR points at <repo>/lp_amd/lib (where `make -C lp_amd/csrc` leaves the library).
fn main() {
    if std::env::var_os("CARGO_FEATURE_HIP").is_some() {
        let dir = std::env::var("LPIPM_LIB_DIR").expect("set LPIPM_LIB_DIR to the directory holding liblpipm.so");
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-lib=dylib=lpipm");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
        println!("cargo:rerun-if-env-changed=LPIPM_LIB_DIR");
    }
}
