// Links libesim.so (built by `make -C epidemicsimulator_amd/csrc ARCH=gfx950`).  ESIM_LIB_DIR points at the directory holding it.
fn main() {
    let dir = std::env::var("ESIM_LIB_DIR").unwrap_or_else(|_| "../../epidemicsimulator_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=esim");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=ESIM_LIB_DIR");
}
