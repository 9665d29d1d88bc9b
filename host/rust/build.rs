// Link libdspfx.so (built by `make -C dsp-stuff_amd/csrc`).  DSPFX_LIB_DIR = the directory holding it.
fn main() {
    let dir = std::env::var("DSPFX_LIB_DIR").expect("set DSPFX_LIB_DIR to the directory of libdspfx.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=dspfx");
    println!("cargo:rerun-if-env-changed=DSPFX_LIB_DIR");
}
