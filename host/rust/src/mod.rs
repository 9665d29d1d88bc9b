//! `mod gpu;` -- libdspfx binding for the dsp-stuff host (see host/rust/README.md).
pub mod engine;
pub mod ffi;
pub mod gpu_bank;
pub mod gpu_chain;
