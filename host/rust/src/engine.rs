//! Safe wrapper over `ffi.rs`: one `Engine` = N independent mono channels through one chain.
//! Mirrors `include/dspfx.hpp` / `dsp-stuff_amd/__init__.py` (same constructor names as the reference's
//! nodes, same defaults).  NOT compiled in the build container.
use super::ffi::*;
use std::ffi::CStr;
use std::os::raw::c_int;
use std::ptr;

#[derive(Debug)]
pub struct Error {
    pub status: c_int,
    pub message: String,
}

impl std::fmt::Display for Error {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "dspfx error {}: {}", self.status, self.message)
    }
}
impl std::error::Error for Error {}

/// One chain node: the descriptor plus the tap storage it may point into.
#[derive(Clone, Debug)]
pub struct NodeDesc {
    pub d: dspfx_node_desc,
    taps: Vec<f64>, // stored time-reversed, like fir.rs:163,168
}

// `dspfx_node_desc.taps` is a raw pointer, which makes the struct !Send / !Sync by default -- and with it GpuChain / GpuBank, which
// keep their chain as `Mutex<Vec<NodeDesc>>`, while the reference's `trait Node: Send + Sync` (node.rs:105) demands both (ADVICE
// r04).  The pointer is only ever null or aimed into this value's own `taps` Vec (set right before the FFI call in `raw()`), so
// moving or sharing a NodeDesc between threads moves / shares plain data.
unsafe impl Send for NodeDesc {}
unsafe impl Sync for NodeDesc {}

impl NodeDesc {
    fn defaults(kind: c_int) -> Self {
        let mut d = dspfx_node_desc { kind, mode: 0, params: [0.0; 8], delay_len: 0, n_taps: 0, taps: ptr::null() };
        let rc = unsafe { dspfx_node_defaults(kind, &mut d) };
        assert_eq!(rc, DSPFX_OK, "dspfx_node_defaults({kind})");
        NodeDesc { d, taps: Vec::new() }
    }
    fn with(kind: c_int, params: &[f32]) -> Self {
        let mut n = Self::defaults(kind);
        n.d.params[..params.len()].copy_from_slice(params);
        n
    }
    /// nodes/gain.rs: level 0..=10, default 1
    pub fn gain(level: f32) -> Self { Self::with(DSPFX_GAIN, &[level]) }
    /// nodes/biquad.rs:18-41: raw sliders a0,a1,a2,b0,b1,b2 (normalised by a0 in the engine)
    pub fn biquad(a0: f32, a1: f32, a2: f32, b0: f32, b1: f32, b2: f32) -> Self { Self::with(DSPFX_BIQUAD, &[a0, a1, a2, b0, b1, b2]) }
    pub fn low_pass(ratio: f32) -> Self { Self::with(DSPFX_LOW_PASS, &[ratio]) }
    pub fn high_pass(ratio: f32) -> Self { Self::with(DSPFX_HIGH_PASS, &[ratio]) }
    /// nodes/reverb.rs: feedback delay, a RESTORED node (`refresh_seconds` has run: the ring has reverb.rs:58's length).
    /// The seconds slider travels with the node (params[1]): any later `set_param` on it -- decay included -- swaps in a
    /// new zero ring of that length, like the reference's after_settings_change (reverb.rs:19, 55-71).
    pub fn reverb(seconds: f32, decay: f32) -> Self {
        let mut n = Self::with(DSPFX_REVERB, &[decay, seconds]);
        n.d.delay_len = unsafe { dspfx_delay_len(seconds, 0) };
        n
    }
    /// nodes/distort.rs: level 0..=30 (0 = bypass), mode = DSPFX_DIST_*
    pub fn distort(level: f32, mode: c_int) -> Self {
        let mut n = Self::with(DSPFX_DISTORT, &[level]);
        n.d.mode = mode;
        n
    }
    pub fn overdrive(boost: f32, drive: f32, level: f32) -> Self { Self::with(DSPFX_OVERDRIVE, &[boost, drive, level]) }
    pub fn chebyshev(level_pos: f32, level_neg: f32) -> Self { Self::with(DSPFX_CHEBYSHEV, &[level_pos, level_neg]) }
    /// nodes/fir.rs: `impulse_response` in natural order h[0..T)
    pub fn fir(impulse_response: &[f64], mode: c_int) -> Self {
        let mut n = Self::defaults(DSPFX_FIR);
        n.taps = impulse_response.iter().rev().copied().collect();
        n.d.mode = mode;
        n
    }
    pub fn add() -> Self { Self::defaults(DSPFX_ADD) }
    pub fn mix(ratio: f32) -> Self { Self::with(DSPFX_MIX, &[ratio]) }
    /// nodes/signal_gen.rs: a source (no "in" port); mode = DSPFX_SIG_*
    pub fn signal_gen(amplitude: f32, frequency: f32, mode: c_int) -> Self {
        let mut n = Self::with(DSPFX_SIGNAL_GEN, &[amplitude, frequency]);
        n.d.mode = mode;
        n
    }
    /// nodes/envelope.rs: attack / release in frames
    pub fn envelope(attack: f32, release: f32) -> Self { Self::with(DSPFX_ENVELOPE, &[attack, release]) }

    /// The node as JSON -- what `GpuChain` / `GpuBank` put under "chain" in the saved graph (runtime.rs:560-564 saves whatever
    /// `Node::save` returns): kind, mode, the eight slider slots, the delay length, the impulse response in natural order.
    pub fn to_json(&self) -> serde_json::Value {
        let ir: Vec<f64> = self.taps.iter().rev().copied().collect();
        serde_json::json!({ "kind": self.d.kind, "mode": self.d.mode, "params": self.d.params.to_vec(),
                            "delay_len": self.d.delay_len, "impulse_response": ir })
    }
    /// Inverse of `to_json`; `None` for anything that is not a node of this library (an unknown kind, a missing field).
    pub fn from_json(v: &serde_json::Value) -> Option<Self> {
        let kind = v.get("kind")?.as_i64()? as c_int;
        if !(0..DSPFX_N_KINDS).contains(&kind) {
            return None;
        }
        let mut n = Self::defaults(kind);
        n.d.mode = v.get("mode")?.as_i64()? as c_int;
        for (slot, p) in n.d.params.iter_mut().zip(v.get("params")?.as_array()?) {
            *slot = p.as_f64()? as f32;
        }
        n.d.delay_len = v.get("delay_len")?.as_u64()? as u32;
        n.taps = v.get("impulse_response")?.as_array()?.iter().rev().filter_map(|t| t.as_f64()).collect();
        Some(n)
    }
    /// A saved "chain" array; `None` if any entry is not a node.
    pub fn chain_from_json(v: &serde_json::Value) -> Option<Vec<Self>> {
        v.as_array()?.iter().map(Self::from_json).collect()
    }
}

/// The C engine, destroyed when the LAST holder goes: the `Engine` and every `ParamHandle` made from it share it, so a GUI
/// thread's handle can never point at freed memory (round 3's handle was a bare pointer with a comment).
struct Raw(*mut dspfx_engine);
unsafe impl Send for Raw {}
unsafe impl Sync for Raw {}      // every C entry point may be called from any thread (include/dspfx.h, "threads")
impl Drop for Raw {
    fn drop(&mut self) {
        unsafe { dspfx_engine_destroy(self.0) }
    }
}

pub struct Engine {
    raw: std::sync::Arc<Raw>,
    h: *mut dspfx_engine,        // == raw.0
    channels: u32,
}

// The engine may be driven by one thread at a time (include/dspfx.h); callers wrap it in a Mutex.
unsafe impl Send for Engine {}

impl Engine {
    pub fn new(channels: u32, max_frames: u32, link_flags: u32, device: i32) -> Result<Self, Error> {
        let desc = dspfx_engine_desc {
            abi_version: DSPFX_ABI_VERSION, device, channels, max_frames, link_flags, tile_channels: 0, channel_offset: 0,
        };
        let mut h = ptr::null_mut();
        let rc = unsafe { dspfx_engine_create(&desc, &mut h) };
        if rc != DSPFX_OK {
            // no CPU fallback: a missing device is an error the caller sees
            let msg = unsafe { CStr::from_ptr(dspfx_strerror(rc)) }.to_string_lossy().into_owned();
            return Err(Error { status: rc, message: msg });
        }
        Ok(Engine { raw: std::sync::Arc::new(Raw(h)), h, channels })
    }

    fn check(&self, rc: c_int) -> Result<(), Error> {
        if rc == DSPFX_OK {
            return Ok(());
        }
        let msg = unsafe { CStr::from_ptr(dspfx_last_error(self.h)) }.to_string_lossy().into_owned();
        Err(Error { status: rc, message: msg })
    }

    pub fn channels(&self) -> u32 { self.channels }

    pub fn set_chain(&mut self, chain: &[NodeDesc]) -> Result<(), Error> {
        let descs: Vec<dspfx_node_desc> = chain
            .iter()
            .map(|n| {
                let mut d = n.d;
                d.n_taps = n.taps.len() as u32;
                d.taps = if n.taps.is_empty() { ptr::null() } else { n.taps.as_ptr() };
                d
            })
            .collect();
        let rc = unsafe { dspfx_chain_set(self.h, descs.as_ptr(), descs.len() as c_int) };
        self.check(rc)
    }

    /// Kernels specialised for the chain's shape are compiled in the background and adopted at a block boundary (the engine
    /// serves on its interpreting kernel meanwhile, same samples): wait up to `wait_ms` for them.  `Ok(true)`: nothing pending.
    pub fn kernels_ready(&mut self, wait_ms: i32) -> Result<bool, Error> {
        let rc = unsafe { dspfx_kernels_ready(self.h, wait_ms as c_int) };
        if rc < 0 { self.check(rc)?; }
        Ok(rc == 1)
    }

    /// A whole saved graph (`DSPConfig`: runtime.rs:560-612) as ONE generated kernel: `nodes` in an order in which
    /// every link goes forward, `links` as (producer, consumer, port) with `DSPFX_GRAPH_INPUT` / `DSPFX_GRAPH_ZERO`
    /// producers and consumer == nodes.len() for the Output node.  `Err` with status `DSPFX_ERR_UNSUPPORTED` when
    /// the graph cannot be fused (FIR / Fuzz node, more than `DSPFX_GRAPH_MAX_NODES` nodes): evaluate it run by run.
    pub fn set_graph(&mut self, nodes: &[NodeDesc], links: &[dspfx_graph_link]) -> Result<(), Error> {
        let descs: Vec<dspfx_node_desc> = nodes.iter().map(|n| n.d).collect();
        let rc = unsafe {
            dspfx_graph_set(self.h, descs.as_ptr(), descs.len() as c_int, links.as_ptr(), links.len() as c_int)
        };
        self.check(rc)
    }

    /// A slider store (dsp-stuff-derive/src/lib.rs:487-492) including the reference's
    /// `after_settings_change` side effects: biquad.rs:15, 62-76 (coefficients renormalised, state reset); reverb.rs:19, 55-71
    /// (ANY slider of a Reverb node, decay included: a new zero-filled ring -- the echo tail is cut).
    pub fn set_param(&mut self, node: usize, param: usize, value: f32) -> Result<(), Error> {
        let rc = unsafe { dspfx_set_param(self.h, node as c_int, param as c_int, value) };
        self.check(rc)
    }
    pub fn set_mode(&mut self, node: usize, mode: c_int) -> Result<(), Error> {
        let rc = unsafe { dspfx_set_mode(self.h, node as c_int, mode) };
        self.check(rc)
    }
    /// A handle the GUI thread keeps for its slider stores while the node's task owns the `Engine` (behind its
    /// `Mutex`) and is inside `process`: the reference's widgets store into atomics from the GUI thread
    /// (dsp-stuff-derive/src/lib.rs:487-492).  The C side queues the store and applies it at the next block boundary.
    pub fn params(&self) -> ParamHandle {
        ParamHandle { raw: self.raw.clone() }
    }
    /// The stores applied so far with a sequence number above `after_seq`, oldest first.
    pub fn param_log(&mut self, after_seq: u64) -> Vec<dspfx_param_event> {
        let mut ev = vec![dspfx_param_event { seq: 0, frame: 0, node: 0, param: 0, value: 0.0, reserved: 0 }; 4096];
        let n = unsafe { dspfx_param_log(self.h, ev.as_mut_ptr(), ev.len() as c_int, after_seq) };
        ev.truncate(n.max(0) as usize);
        ev
    }
    pub fn frames_submitted(&self) -> u64 {
        unsafe { dspfx_frames_submitted(self.h) }
    }
    pub fn set_delay_seconds(&mut self, node: usize, seconds: f32) -> Result<(), Error> {
        let rc = unsafe { dspfx_set_delay_len(self.h, node as c_int, dspfx_delay_len(seconds, 0)) };
        self.check(rc)
    }
    pub fn reset(&mut self) -> Result<(), Error> {
        let rc = unsafe { dspfx_reset(self.h) };
        self.check(rc)
    }
    /// Capacity hint: the 128-row groups a delay of `seconds` at `node` would need are allocated now (no engine lock taken), so
    /// that no later seconds store allocates.  `dspfx_delay_len(1.0, 0)` covers the whole slider (reverb.rs:34-37).
    pub fn reserve_delay_seconds(&mut self, node: usize, seconds: f32) -> Result<(), Error> {
        let rc = unsafe { dspfx_reserve_delay_len(self.h, node as c_int, dspfx_delay_len(seconds, 0)) };
        self.check(rc)
    }
    /// Give back delay-ring capacity beyond the rings' current lengths (waits for the device).
    pub fn ring_trim(&mut self) -> Result<(), Error> {
        let rc = unsafe { dspfx_ring_trim(self.h) };
        self.check(rc)
    }

    /// One block from host slices, frame-major `[n_frames][channels]`; synchronous (H2D, chain, D2H).
    pub fn process_host(&mut self, input: &[f32], side: Option<&[f32]>, out: &mut [f32], mix: Option<&mut [f32]>,
                        n_frames: u32) -> Result<(), Error> {
        let want = n_frames as usize * self.channels as usize;
        assert!(input.len() == want && out.len() == want, "block must be [n_frames][channels]");
        if let Some(s) = side {
            assert_eq!(s.len(), want);
        }
        let mix_ptr = match mix {
            Some(m) => {
                assert_eq!(m.len(), n_frames as usize);
                m.as_mut_ptr()
            }
            None => ptr::null_mut(),
        };
        let rc = unsafe {
            dspfx_process_host(self.h, input.as_ptr(), side.map_or(ptr::null(), |s| s.as_ptr()), out.as_mut_ptr(), mix_ptr, n_frames)
        };
        self.check(rc)
    }

    /// Impulse-response reload (nodes/fir.rs:153-171): new taps, given in natural order h[0..T); the history is kept,
    /// exactly like the reference's `state` deque.
    pub fn set_taps(&mut self, node: usize, impulse_response: &[f64], mode: c_int) -> Result<(), Error> {
        let rev: Vec<f64> = impulse_response.iter().rev().copied().collect();
        let rc = unsafe { dspfx_set_taps(self.h, node as c_int, rev.as_ptr(), rev.len() as u32, mode) };
        self.check(rc)
    }

    /// How a FIR node's steady-state sweep multiplies: `DSPFX_FIR_PRECISION_DEFAULT`, `_F32` or `_SPLIT` (three bf16 parts per
    /// f32 operand on the bf16 matrix pipe: the same stated tolerance, 1.5 x faster).
    pub fn set_fir_precision(&mut self, node: usize, precision: c_int) -> Result<(), Error> {
        let rc = unsafe { dspfx_set_fir_precision(self.h, node as c_int, precision) };
        self.check(rc)
    }

    /// The mix bus across GPUs: sum this rank's un-normalised bus (device pointer, `n_frames` f32) over the
    /// communicator's ranks -- ONE RCCL all-reduce, in place, asynchronous on `stream` -- then the Output node's hop
    /// with the GLOBAL channel count (node.rs:189-191).
    ///
    /// # Safety
    /// `mix` must be a device pointer to at least `n_frames` floats that stays valid until the stream has run the call.
    pub unsafe fn mix_allreduce(&mut self, comm: &mut Comm, mix: *mut f32, n_frames: u32, n_channels_total: u64,
                                stream: *mut std::os::raw::c_void) -> Result<(), Error> {
        let rc = dspfx_mix_allreduce(self.h, comm.h, mix, n_frames, n_channels_total, stream);
        self.check(rc)
    }

    /// A graph engine with several input / output blocks (a REGION of a graph cut into several kernels): device pointers,
    /// `ins[k]` = input block k, `outs[m]` = output block m.
    ///
    /// # Safety
    /// Every non-null pointer must address a whole block in the engine's layout and stay valid until the stream has run the call.
    pub unsafe fn process_io(&mut self, ins: &[*const f32], outs: &[*mut f32], n_frames: u32,
                             stream: *mut std::os::raw::c_void) -> Result<(), Error> {
        let rc = dspfx_process_io(self.h, ins.as_ptr(), ins.len() as c_int, outs.as_ptr(), outs.len() as c_int, ptr::null_mut(), n_frames, stream);
        self.check(rc)
    }

    pub fn describe(&self) -> String {
        let mut buf = vec![0u8; 64 << 10];
        let rc = unsafe { dspfx_describe(self.h, buf.as_mut_ptr() as *mut _, buf.len()) };
        if rc != DSPFX_OK {
            return String::new();
        }
        let end = buf.iter().position(|&b| b == 0).unwrap_or(buf.len());
        String::from_utf8_lossy(&buf[..end]).into_owned()
    }
}

/// The mix bus' communicator: one per process / GPU (`dspfx_comm_create`).  Rank 0 calls `Comm::unique_id()` and hands the
/// bytes to every rank over the host's own control channel; `Comm::new` is collective.
pub struct Comm {
    h: *mut dspfx_comm,
}
unsafe impl Send for Comm {}

impl Comm {
    pub fn unique_id() -> Result<[u8; DSPFX_COMM_ID_BYTES], Error> {
        let mut id = [0u8; DSPFX_COMM_ID_BYTES];
        let rc = unsafe { dspfx_comm_unique_id(id.as_mut_ptr() as *mut _) };
        if rc != DSPFX_OK {
            return Err(Error { status: rc, message: "dspfx_comm_unique_id".into() });
        }
        Ok(id)
    }
    pub fn new(device: i32, n_ranks: i32, rank: i32, id: Option<&[u8; DSPFX_COMM_ID_BYTES]>) -> Result<Self, Error> {
        let mut h = ptr::null_mut();
        let rc = unsafe { dspfx_comm_create(device, n_ranks, rank, id.map_or(ptr::null(), |b| b.as_ptr() as *const _), &mut h) };
        if rc != DSPFX_OK {
            return Err(Error { status: rc, message: "dspfx_comm_create".into() });
        }
        Ok(Comm { h })
    }
    pub fn size(&self) -> i32 { unsafe { dspfx_comm_size(self.h) } }
    pub fn rank(&self) -> i32 { unsafe { dspfx_comm_rank(self.h) } }
}

impl Drop for Comm {
    fn drop(&mut self) {
        unsafe { dspfx_comm_destroy(self.h) }
    }
}

/// Slider / mode stores from a thread that does not own the `Engine` (see `Engine::params`).  Sound because
/// `dspfx_set_param` / `dspfx_set_mode` are thread-safe by contract (include/dspfx.h, "threads"): they only queue.
/// The handle keeps the C engine alive: it is destroyed when the `Engine` AND every handle are gone.
#[derive(Clone)]
pub struct ParamHandle {
    raw: std::sync::Arc<Raw>,
}
impl ParamHandle {
    /// Returns the store's sequence number.  `Err(DSPFX_ERR_STATE)`: the node's task replaced the chain (`dspfx_chain_set`) while this
    /// store was being made -- it was NOT stored (it had been checked against a node that is gone); `Err(DSPFX_ERR_OOM)`: a Reverb
    /// store whose longer ring does not fit -- ring and slider stay as they were.  Either way the widget keeps its old value.
    pub fn set_param(&self, node: usize, param: usize, value: f32) -> Result<u64, c_int> {
        let mut seq = 0u64;
        let rc = unsafe { dspfx_set_param_seq(self.raw.0, node as c_int, param as c_int, value, &mut seq) };
        if rc == 0 { Ok(seq) } else { Err(rc) }
    }
    pub fn set_mode(&self, node: usize, mode: c_int) -> Result<(), c_int> {
        let rc = unsafe { dspfx_set_mode(self.raw.0, node as c_int, mode) };
        if rc == 0 { Ok(()) } else { Err(rc) }
    }
}
