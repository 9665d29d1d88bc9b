//! Raw bindings of `include/dspfx.h` (ABI version 1).  Kept in step with the header by
//! `tests/test_rust_shim_sync.py`; NOT compiled in the build container (no rustc there).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

/// Opaque engine handle (`typedef struct dspfx_engine dspfx_engine`).
#[repr(C)]
pub struct dspfx_engine {
    _private: [u8; 0],
}

/// Opaque communicator handle (`typedef struct dspfx_comm dspfx_comm`).
#[repr(C)]
pub struct dspfx_comm {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct dspfx_engine_desc {
    pub abi_version: u32,
    pub device: i32,
    pub channels: u32,
    pub max_frames: u32,
    pub link_flags: u32,
    pub tile_channels: u32,
    pub channel_offset: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct dspfx_node_desc {
    pub kind: i32,
    pub mode: i32,
    pub params: [f32; 8],
    pub delay_len: u32,
    pub n_taps: u32,
    pub taps: *const f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct dspfx_ctl {
    pub node: i32,
    pub param: i32,
    pub signal: *const f32,
}

/// One link of a graph given to `dspfx_graph_set` (include/dspfx.h).
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct dspfx_graph_link {
    pub src: i32,
    pub dst: i32,
    pub port: i32,
}

/// Where a slider / mode store took effect (`dspfx_param_log`).
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct dspfx_param_event {
    pub seq: u64,
    pub frame: u64,
    pub node: i32,
    pub param: i32,
    pub value: f32,
    pub reserved: i32,
}

pub const DSPFX_ABI_VERSION: u32 = 2;
pub const DSPFX_BUF_SIZE: u32 = 128; // dsp-stuff/src/node.rs:257
pub const DSPFX_MAX_NODES: u32 = 32;
pub const DSPFX_COMM_ID_BYTES: usize = 128;

// dspfx_status
pub const DSPFX_OK: c_int = 0;
pub const DSPFX_ERR_INVALID: c_int = -1;
pub const DSPFX_ERR_NO_DEVICE: c_int = -2;
pub const DSPFX_ERR_HIP: c_int = -3;
pub const DSPFX_ERR_OOM: c_int = -4;
pub const DSPFX_ERR_UNSUPPORTED: c_int = -5;
pub const DSPFX_ERR_STATE: c_int = -6;

// link flags
pub const DSPFX_LINK_INTERNAL: u32 = 1;
pub const DSPFX_LINK_INPUT: u32 = 2;
pub const DSPFX_LINK_SIDE_RAW: u32 = 4;
pub const DSPFX_MAX_LINKS: u32 = 16;
pub const DSPFX_GRAPH_MAX_NODES: u32 = 16;
pub const DSPFX_GRAPH_INPUT: i32 = -1;
pub const DSPFX_GRAPH_ZERO: i32 = -2;
pub const DSPFX_GRAPH_INPUT2: i32 = -3;
pub const DSPFX_GRAPH_MAX_IO: u32 = 16;
/// Link source of input block k (`DSPFX_GRAPH_INPUT_N`).
pub const fn dspfx_graph_input_n(k: i32) -> i32 {
    if k == 0 { DSPFX_GRAPH_INPUT } else if k == 1 { DSPFX_GRAPH_INPUT2 } else { -(2 + k) }
}
pub const DSPFX_PORT_MAIN: i32 = 0;
pub const DSPFX_PORT_SIDE: i32 = 1;
pub const DSPFX_PORT_SLIDER: i32 = 2;
pub const DSPFX_PORT_RAW: i32 = 256;

// dspfx_kind
pub const DSPFX_GAIN: c_int = 0;
pub const DSPFX_BIQUAD: c_int = 1;
pub const DSPFX_LOW_PASS: c_int = 2;
pub const DSPFX_HIGH_PASS: c_int = 3;
pub const DSPFX_REVERB: c_int = 4;
pub const DSPFX_DISTORT: c_int = 5;
pub const DSPFX_OVERDRIVE: c_int = 6;
pub const DSPFX_CHEBYSHEV: c_int = 7;
pub const DSPFX_FIR: c_int = 8;
pub const DSPFX_ADD: c_int = 9;
pub const DSPFX_MIX: c_int = 10;
pub const DSPFX_SIGNAL_GEN: c_int = 11;
pub const DSPFX_ENVELOPE: c_int = 12;
pub const DSPFX_N_KINDS: c_int = 13;

// dspfx_distort_mode (nodes/distort.rs:18-28)
pub const DSPFX_DIST_HARD_CLIP: c_int = 0;
pub const DSPFX_DIST_SOFT_CLIP: c_int = 1;
pub const DSPFX_DIST_TANH: c_int = 2;
pub const DSPFX_DIST_RECIP_SOFT_CLIP: c_int = 3;
pub const DSPFX_DIST_FUZZ: c_int = 4;
pub const DSPFX_DIST_SIN: c_int = 5;
pub const DSPFX_DIST_ATAN: c_int = 6;
pub const DSPFX_DIST_SQUARE: c_int = 7;
pub const DSPFX_DIST_CHEBYSHEV4: c_int = 8;

// dspfx_signal_mode (nodes/signal_gen.rs:17-22)
pub const DSPFX_SIG_SINE: c_int = 0;
pub const DSPFX_SIG_TRIANGLE: c_int = 1;
pub const DSPFX_SIG_SQUARE: c_int = 2;
pub const DSPFX_SIG_CONSTANT: c_int = 3;

// dspfx_fir_mode (nodes/fir.rs)
pub const DSPFX_FIR_BALANCED: c_int = 0;
pub const DSPFX_FIR_AVERAGE: c_int = 1;
/// dspfx_fir_precision (dspfx.h): how a FIR node's steady-state sweep multiplies
pub const DSPFX_FIR_PRECISION_DEFAULT: c_int = 0;
pub const DSPFX_FIR_PRECISION_F32: c_int = 1;
pub const DSPFX_FIR_PRECISION_SPLIT: c_int = 2;
pub const DSPFX_FIR_PRECISION_HALF: c_int = 3;

#[link(name = "dspfx")]
extern "C" {
    pub fn dspfx_abi_version() -> u32;
    pub fn dspfx_strerror(status: c_int) -> *const c_char;
    pub fn dspfx_device_count() -> c_int;
    pub fn dspfx_node_defaults(kind: c_int, d: *mut dspfx_node_desc) -> c_int;
    pub fn dspfx_delay_len(seconds: f32, page_round: c_int) -> u32;
    pub fn dspfx_link_divisor(n_connected: u64) -> f32;

    pub fn dspfx_engine_create(desc: *const dspfx_engine_desc, out: *mut *mut dspfx_engine) -> c_int;
    pub fn dspfx_engine_destroy(e: *mut dspfx_engine);
    pub fn dspfx_last_error(e: *const dspfx_engine) -> *const c_char;

    pub fn dspfx_chain_set(e: *mut dspfx_engine, nodes: *const dspfx_node_desc, n_nodes: c_int) -> c_int;
    pub fn dspfx_chain_len(e: *const dspfx_engine) -> c_int;
    pub fn dspfx_kernels_ready(e: *mut dspfx_engine, wait_ms: c_int) -> c_int;
    pub fn dspfx_set_param(e: *mut dspfx_engine, node: c_int, param: c_int, value: f32) -> c_int;
    pub fn dspfx_set_param_seq(e: *mut dspfx_engine, node: c_int, param: c_int, value: f32, seq: *mut u64) -> c_int;
    pub fn dspfx_set_mode(e: *mut dspfx_engine, node: c_int, mode: c_int) -> c_int;
    pub fn dspfx_param_log(e: *mut dspfx_engine, dst: *mut dspfx_param_event, cap: c_int, after_seq: u64) -> c_int;
    pub fn dspfx_frames_submitted(e: *const dspfx_engine) -> u64;
    pub fn dspfx_set_delay_len(e: *mut dspfx_engine, node: c_int, delay_len: u32) -> c_int;
    pub fn dspfx_reserve_delay_len(e: *mut dspfx_engine, node: c_int, delay_len: u32) -> c_int;
    pub fn dspfx_ring_trim(e: *mut dspfx_engine) -> c_int;
    pub fn dspfx_set_taps(e: *mut dspfx_engine, node: c_int, taps_reversed: *const f64, n_taps: u32, mode: c_int) -> c_int;
    pub fn dspfx_set_fir_precision(e: *mut dspfx_engine, node: c_int, precision: c_int) -> c_int;
    pub fn dspfx_reset(e: *mut dspfx_engine) -> c_int;

    pub fn dspfx_tune_placement(e: *mut dspfx_engine, input: *const f32, side: *const f32, out: *mut f32, n_frames: u32, stream: *mut c_void) -> c_int;
    pub fn dspfx_process(e: *mut dspfx_engine, input: *const f32, side: *const f32, out: *mut f32, mix: *mut f32, n_frames: u32, stream: *mut c_void) -> c_int;
    pub fn dspfx_process_bus(e: *mut dspfx_engine, input: *const f32, side: *const f32, out: *mut f32, mix: *mut f32, n_frames: u32, n_connected: u64, stream: *mut c_void) -> c_int;
    pub fn dspfx_process_ctl(e: *mut dspfx_engine, input: *const f32, side: *const f32, out: *mut f32, mix: *mut f32, n_frames: u32, ctl: *const dspfx_ctl, n_ctl: c_int, stream: *mut c_void) -> c_int;
    pub fn dspfx_host_alloc(bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn dspfx_host_free(p: *mut c_void) -> c_int;
    pub fn dspfx_process_host(e: *mut dspfx_engine, input: *const f32, side: *const f32, out: *mut f32, mix: *mut f32, n_frames: u32) -> c_int;
    pub fn dspfx_process_partials(e: *mut dspfx_engine, input: *const f32, side: *const f32, out: *mut f32, n_frames: u32, stream: *mut c_void) -> c_int;
    pub fn dspfx_mix_collect(e: *mut dspfx_engine, mix: *mut f32, n_frames: u32, stream: *mut c_void) -> c_int;
    pub fn dspfx_mix_finish(e: *mut dspfx_engine, mix: *mut f32, n_frames: u32, n_connected: u64, stream: *mut c_void) -> c_int;

    // the mix bus across GPUs: one RCCL all-reduce of n_frames floats per block (include/dspfx.h)
    pub fn dspfx_comm_unique_id(id_out: *mut c_void) -> c_int;
    pub fn dspfx_comm_create(device: c_int, n_ranks: c_int, rank: c_int, id: *const c_void, out: *mut *mut dspfx_comm) -> c_int;
    pub fn dspfx_comm_destroy(c: *mut dspfx_comm);
    pub fn dspfx_comm_size(c: *const dspfx_comm) -> c_int;
    pub fn dspfx_comm_rank(c: *const dspfx_comm) -> c_int;
    pub fn dspfx_comm_last_error(c: *const dspfx_comm) -> *const c_char;
    pub fn dspfx_comm_backend(c: *const dspfx_comm) -> *const c_char;
    pub fn dspfx_mix_allreduce(e: *mut dspfx_engine, c: *mut dspfx_comm, mix: *mut f32, n_frames: u32, n_connected: u64, stream: *mut c_void) -> c_int;

    pub fn dspfx_process_mixpipe(e: *mut dspfx_engine, input: *const f32, side: *const f32, out: *mut f32, mix: *mut f32, n_frames: u32, n_connected: u64, stream: *mut c_void) -> c_int;
    pub fn dspfx_mixpipe_flush(e: *mut dspfx_engine, mix_older: *mut f32, mix_newer: *mut f32, n_connected: u64, stream: *mut c_void) -> c_int;
    pub fn dspfx_process_io(e: *mut dspfx_engine, ins: *const *const f32, n_ins: c_int, outs: *const *mut f32, n_outs: c_int, mix: *mut f32, n_frames: u32, stream: *mut c_void) -> c_int;
    pub fn dspfx_graph_set(e: *mut dspfx_engine, nodes: *const dspfx_node_desc, n_nodes: c_int, links: *const dspfx_graph_link, n_links: c_int) -> c_int;
    pub fn dspfx_graph_source(nodes: *const dspfx_node_desc, n_nodes: c_int, links: *const dspfx_graph_link, n_links: c_int, dst: *mut c_char, cap: usize) -> c_int;
    pub fn dspfx_link_average(e: *mut dspfx_engine, srcs: *const *const f32, n_srcs: c_int, dst: *mut f32, n_frames: u32, stream: *mut c_void) -> c_int;

    pub fn dspfx_state_size(e: *const dspfx_engine, node: c_int) -> i64;
    pub fn dspfx_state_export(e: *mut dspfx_engine, node: c_int, host_dst: *mut c_void, size: usize) -> c_int;
    pub fn dspfx_state_import(e: *mut dspfx_engine, node: c_int, host_src: *const c_void, size: usize) -> c_int;

    pub fn dspfx_fill_noise(e: *mut dspfx_engine, dst: *mut f32, n_frames: u32, n_abs0: u32, seed: u32, stream: *mut c_void) -> c_int;
    pub fn dspfx_sync(e: *mut dspfx_engine, stream: *mut c_void) -> c_int;
    pub fn dspfx_describe(e: *const dspfx_engine, dst: *mut c_char, cap: usize) -> c_int;
    pub fn dspfx_verify_fast_division(device: c_int, c: f32, mismatches: *mut u64) -> c_int;
    pub fn dspfx_verify_libm(device: c_int, func: c_int, mismatches: *mut u64, max_ulp: *mut u32) -> c_int;
    pub fn dspfx_profile_enable(e: *mut dspfx_engine, enable: c_int) -> c_int;
    pub fn dspfx_profile_read(e: *mut dspfx_engine, total_ms: *mut f64, launches: *mut u32, kernel_name: *mut c_char, cap: usize, reset: c_int) -> c_int;
    pub fn dspfx_algorithmic_bytes_per_sample(e: *const dspfx_engine, n_frames: u32) -> f64;
}
