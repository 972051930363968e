// Build script of the raytracer crate with the MI355X render path.
//
// Compiles the HIP sources of librt_amd (vendored into the crate under hip/, together with
// include/rt_abi.h) for gfx950 and links the crate against it.  `-ffp-contract=off` is part of
// the contract, not a tuning flag: the kernel must round every f64 operation separately to give
// the bits the crate's own arithmetic gives.  `-disable-machine-licm` only affects speed (register
// pressure).
use std::{env, path::PathBuf, process::Command};

fn main() {
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".to_string());
    let arch = env::var("RT_OFFLOAD_ARCH").unwrap_or_else(|_| "gfx950".to_string());
    let sources = ["hip/rt_amd.hip", "hip/rt_multi.hip"];
    let lib = out.join("librt_amd.so");
    let status = Command::new(&hipcc)
        .arg(format!("--offload-arch={arch}"))
        .args(["-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-mllvm", "-disable-machine-licm", "-Iinclude",
               "-Ihip", "-shared", "-pthread", "-o"])
        .arg(&lib)
        .args(sources)
        .status()
        .expect("hipcc not found: set HIPCC");
    assert!(status.success(), "hipcc failed");
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=rt_amd");
    println!("cargo:rustc-link-search=native=/opt/rocm/lib");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    println!("cargo:rustc-env=RT_AMD_LIB_DIR={}", out.display());
    for s in sources {
        println!("cargo:rerun-if-changed={s}");
    }
    println!("cargo:rerun-if-changed=hip/rt_download.h");
    println!("cargo:rerun-if-changed=include/rt_abi.h");
}
