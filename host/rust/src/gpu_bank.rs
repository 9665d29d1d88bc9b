//! `GpuBank`: the N-channel engine AS A NODE of the reference graph -- what the scheduler (runtime.rs:646-732) can run.
//!
//! A node with N input ports `in0 .. in{N-1}`, N output ports `out0 .. out{N-1}` and one more output, `mix`.  Per 128-frame
//! block the blanket `Perform` wrapper (node.rs:267-352) hands `process` the N averaged input blocks; the node gathers them into
//! ONE page-locked `[128][N]` block, libdspfx evaluates the whole chain for all N channels in one launch
//! (`dspfx_process_host`: upload, kernel and download overlapped in channel parts), and the node scatters the N outputs back
//! into its output ports.  `mix` carries what an Output node fed by all N channels would hold: the per-frame sum divided by
//! f32(0.0001 + N) (node.rs:162-194, nodes/output.rs:215-249) -- the bus the kernel sums in its epilogue anyway.
//!
//! It implements `Node` by hand (the port lists depend on N, which `#[derive(DspNode)]`'s static `input = ".."` cannot say)
//! and `SimpleNode`; `Perform` comes from the blanket impl.  Register it like any node: a variant `GpuBank` in `enum Nodes`,
//! `("GPU bank", |id| Arc::new(Nodes::from(GpuBank::new(id))))` in `NODES`, `("gpu_bank", |v| Arc::new(Nodes::from(
//! GpuBank::restore(v))))` in `RESTORE` (nodes/mod.rs:38-123).
//!
//! The C calls this file makes, in order (tests/test_rust_shim_sync.py checks that each exists with this arity;
//! tests/cpp/test_host.cpp runs the same sequence from C++ against the oracle):
//!   new / restore : dspfx_engine_create(desc, out) -> dspfx_chain_set(e, nodes, n) -> dspfx_host_alloc(bytes, out) x 2
//!                   -> dspfx_link_divisor(n)
//!   process       : dspfx_process_host(e, in, side, out, mix, n_frames)
//!   sliders (GUI) : dspfx_set_param_seq(e, node, param, value, seq) / dspfx_set_mode(e, node, mode)    [ParamHandle]
//!   drop          : dspfx_host_free(p) x 2 -> dspfx_engine_destroy(e)
//! NOT compiled in the build container (no rustc).
use super::engine::{Engine, NodeDesc, ParamHandle};
use super::ffi::{dspfx_link_divisor, DSPFX_LINK_INTERNAL};
use super::gpu_chain::{default_chain, PinnedBlock};
use crate::{
    ids::{NodeId, PortId},
    node::*,
};
use std::collections::HashMap;
use std::sync::Mutex;

/// Channels of a bank made from the menu (a saved bank restores its own count).
pub const DEFAULT_BANK_CHANNELS: usize = 64;

struct BankState {
    engine: Engine,
    gather: PinnedBlock,  // [BUF_SIZE][N], page-locked: dspfx_process_host overlaps upload, kernel and download
    scatter: PinnedBlock, // [BUF_SIZE][N]
    mix: Vec<f32>,        // [BUF_SIZE] un-normalised sum over the N outputs
}

pub struct GpuBank {
    id: NodeId,
    inputs: PortStorage,
    outputs: PortStorage,
    channels: usize,
    in_names: Vec<String>,  // "in0", "in1", ...: looked up by name like every port (node.rs:224-250)
    out_names: Vec<String>,
    divisor: f32,           // f32(0.0001 + N): the Output node's collect_and_average over N pipes
    params: ParamHandle,    // the GUI thread's slider handle (queued stores: lib.rs:487-492 across the FFI)
    chain: Mutex<Vec<NodeDesc>>, // what every channel runs: saved under "chain", replaceable while the graph runs
    state: Mutex<BankState>,
}

impl GpuBank {
    /// `chain`: what every channel runs; the hops BETWEEN its nodes are the engine's (the hop INTO the bank is applied by the
    /// Perform wrapper per input port).
    fn build(id: NodeId, channels: usize, inputs: PortStorage, outputs: PortStorage, chain: Vec<NodeDesc>) -> Self {
        // a missing GPU is fatal, like every other failure on the reference's hot path (node.rs:173,271)
        let mut engine = Engine::new(channels as u32, BUF_SIZE as u32, DSPFX_LINK_INTERNAL, 0).expect("libdspfx engine");
        engine.set_chain(&chain).expect("chain");
        let n = channels * BUF_SIZE;
        let params = engine.params();
        let state = BankState {
            engine,
            gather: PinnedBlock::new(n).expect("dspfx_host_alloc"),
            scatter: PinnedBlock::new(n).expect("dspfx_host_alloc"),
            mix: vec![0.0; BUF_SIZE],
        };
        GpuBank {
            id,
            inputs,
            outputs,
            channels,
            in_names: (0..channels).map(|c| format!("in{c}")).collect(),
            out_names: (0..channels).map(|c| format!("out{c}")).collect(),
            divisor: unsafe { dspfx_link_divisor(channels as u64) },
            params,
            chain: Mutex::new(chain),
            state: Mutex::new(state),
        }
    }

    fn fresh_ports(channels: usize) -> (PortStorage, PortStorage) {
        let ins: HashMap<String, PortId> = (0..channels).map(|c| (format!("in{c}"), PortId::generate())).collect();
        let mut outs: HashMap<String, PortId> = (0..channels).map(|c| (format!("out{c}"), PortId::generate())).collect();
        outs.insert("mix".to_owned(), PortId::generate());
        (PortStorage::new(ins), PortStorage::new(outs))
    }

    /// A bank of `channels` channels of a chain of the host's choosing.
    pub fn with_chain(id: NodeId, channels: usize, chain: Vec<NodeDesc>) -> Self {
        let (ins, outs) = Self::fresh_ports(channels);
        Self::build(id, channels, ins, outs, chain)
    }
    /// Replace the chain while the graph runs (dspfx_chain_set never waits for the run-time compiler; state starts from zero).
    pub fn set_chain(&self, chain: Vec<NodeDesc>) -> Result<(), super::engine::Error> {
        self.state.lock().unwrap().engine.set_chain(&chain)?;
        *self.chain.lock().unwrap() = chain;
        Ok(())
    }
    /// Slider stores from the GUI thread: queued by libdspfx, applied at the next block boundary.
    pub fn params(&self) -> ParamHandle {
        self.params.clone()
    }
}

impl Node for GpuBank {
    fn title(&self) -> &'static str { "GPU bank" }
    fn cfg_name(&self) -> &'static str { "gpu_bank" }
    fn description(&self) -> &'static str {
        "N channels of one effect chain in one launch of libdspfx on an MI355X, plus their mix"
    }
    fn id(&self) -> NodeId { self.id }
    fn inputs(&self) -> &PortStorage { &self.inputs }
    fn outputs(&self) -> &PortStorage { &self.outputs }
    fn render(&self, ui: &mut eframe::egui::Ui) {
        ui.label(format!("{} channels on the GPU", self.channels));
    }
    fn save(&self) -> serde_json::Value {
        let chain: Vec<serde_json::Value> = self.chain.lock().unwrap().iter().map(NodeDesc::to_json).collect();
        serde_json::json!({ "id": self.id, "inputs": self.inputs, "outputs": self.outputs, "channels": self.channels, "chain": chain })
    }
}

impl NodeStatic for GpuBank {
    fn new(id: NodeId) -> Self {
        Self::with_chain(id, DEFAULT_BANK_CHANNELS, default_chain())
    }
    fn restore(value: serde_json::Value) -> Self {
        let id = serde_json::from_value(value["id"].clone()).unwrap();
        let channels = value["channels"].as_u64().unwrap() as usize;
        let ins: PortStorage = serde_json::from_value(value["inputs"].clone()).unwrap();
        let outs: PortStorage = serde_json::from_value(value["outputs"].clone()).unwrap();
        let chain = NodeDesc::chain_from_json(&value["chain"]).unwrap_or_else(default_chain);
        Self::build(id, channels, ins, outs, chain)
    }
}

impl SimpleNode for GpuBank {
    fn process(&self, inputs: ProcessInput, mut outputs: ProcessOutput) {
        let n = self.channels;
        let mut guard = self.state.lock().unwrap();
        let st = &mut *guard;
        let mut frames = BUF_SIZE;
        {
            let gather = st.gather.as_mut_slice();
            for (c, name) in self.in_names.iter().enumerate() {
                let block = inputs.get(name).unwrap(); // already averaged by the Perform wrapper (node.rs:297); zeros if unconnected
                frames = block.len();
                for (f, v) in block.iter().enumerate() {
                    gather[f * n + c] = *v;
                }
            }
        }
        {
            let g = &st.gather.as_slice()[..frames * n];
            let s = &mut st.scatter.as_mut_slice()[..frames * n];
            st.engine.process_host(g, None, s, Some(&mut st.mix[..frames]), frames as u32).expect("dspfx_process_host");
        }
        let scatter = st.scatter.as_slice();
        for (c, name) in self.out_names.iter().enumerate() {
            let block = outputs.get(name).unwrap();
            for (f, v) in block.iter_mut().enumerate() {
                *v = scatter[f * n + c];
            }
        }
        let mix = outputs.get("mix").unwrap();
        for (f, v) in mix.iter_mut().enumerate() {
            *v = st.mix[f] / self.divisor; // node.rs:189-191: what an Output node fed by the N channels holds
        }
    }
}
