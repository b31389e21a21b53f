fn main() {
    // libdfgpu.so is built by `make -C datafusion_amd/csrc` (hipcc --offload-arch=gfx950); point the linker at it
    if let Ok(dir) = std::env::var("DFGPU_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rerun-if-env-changed=DFGPU_LIB_DIR");
}
