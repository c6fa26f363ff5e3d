// build.rs -- builds (or finds) libcute_nt_hip.so and points rustc at it.  Only with the `hip` cargo feature: without it
// the crate compiles no binding and links nothing (the reference crate's own AVX2 / BMI2 functions are untouched).
//
//   CUTE_NT_LIB_DIR=/dir/with/the/.so   link a prebuilt library (e.g. the one `python -m cute_nucleotides_amd.build` or
//                                       `make -C hip` produced);
//   unset                               run `make -C <crate>/../hip OUT=$OUT_DIR product` (hip/Makefile: one hipcc command,
//                                       cross-compiles gfx950 without a GPU) and link the result.  CUTE_NT_HIP_DIR overrides
//                                       where the `hip/` sources live (default: next to this crate; inside the reference
//                                       crate after INTEGRATION.md 2: `hip/` in the crate root -- set it to "hip").
use std::env;
use std::path::PathBuf;
use std::process::Command;

fn main() {
    println!("cargo:rerun-if-env-changed=CUTE_NT_LIB_DIR");
    println!("cargo:rerun-if-env-changed=CUTE_NT_HIP_DIR");
    if env::var_os("CARGO_FEATURE_HIP").is_none() {
        return; // the back-end is opt-in: nothing to build, nothing to link
    }
    let dir = match env::var("CUTE_NT_LIB_DIR") {
        Ok(d) => PathBuf::from(d),
        Err(_) => {
            let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").expect("cargo sets CARGO_MANIFEST_DIR"));
            let hip = match env::var("CUTE_NT_HIP_DIR") {
                Ok(d) => PathBuf::from(d),
                Err(_) => manifest.join("..").join("hip"),
            };
            let out = PathBuf::from(env::var("OUT_DIR").expect("cargo sets OUT_DIR"));
            println!("cargo:rerun-if-changed={}", hip.display());
            let status = Command::new("make")
                .arg("-C")
                .arg(&hip)
                .arg(format!("OUT={}", out.display()))
                .arg("product")
                .status()
                .expect("the `hip` feature needs `make` and `hipcc` (ROCm) to build hip/, or CUTE_NT_LIB_DIR pointing at a prebuilt libcute_nt_hip.so");
            assert!(status.success(), "make -C {} product failed", hip.display());
            out
        }
    };
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=cute_nt_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
}
