// build.rs -- points rustc at libcute_nt_hip.so.  Set CUTE_NT_LIB_DIR to the directory that
// holds it (default: ../cute_nucleotides_amd relative to this crate).
fn main() {
    let dir = std::env::var("CUTE_NT_LIB_DIR").unwrap_or_else(|_| "../cute_nucleotides_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=cute_nt_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=CUTE_NT_LIB_DIR");
}
