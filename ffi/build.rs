// Builds librt_mi355.so with hipcc (gfx950 only) and links it.
// Same flags as ray-tracer_amd/csrc/Makefile: contraction off so kernels reproduce the CPU routines bit for bit.
use std::{env, path::PathBuf, process::Command};

fn main() {
    let root = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("..");
    let csrc = root.join("ray-tracer_amd/csrc");
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".into());
    let lib = out.join("librt_mi355.so");
    let status = Command::new(&hipcc)
        .args(["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"])
        .arg("-o").arg(&lib)
        .arg(csrc.join("rt_api.hip")).arg(csrc.join("rt_build.hip"))
        .status().expect("failed to run hipcc");
    assert!(status.success(), "hipcc failed");
    for f in ["rt_api.hip", "rt_build.hip", "rt_trace.hpp", "rt_math.hpp", "rt_internal.hpp"] {
        println!("cargo:rerun-if-changed={}", csrc.join(f).display());
    }
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=rt_mi355");
}
