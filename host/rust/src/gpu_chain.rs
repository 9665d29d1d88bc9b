//! `GpuChain`: a reference node (`Node` + `SimpleNode`, dsp-stuff/src/node.rs:104-146) whose `process`
//! hands the block to libdspfx.  It replaces the k effect nodes of a chain by one node; with
//! `link_flags = DSPFX_LINK_INTERNAL` the result is what those k nodes produce (the `Perform` wrapper
//! still applies the hop INTO this node, node.rs:290-299; the engine applies the k-1 hops between the
//! fused nodes).  NOT compiled in the build container.
use super::engine::{Engine, NodeDesc};
use super::ffi::{DSPFX_DIST_SOFT_CLIP, DSPFX_LINK_INTERNAL};
use crate::{ids::NodeId, node::*};
use std::sync::Mutex;

/// BASELINE's 5-node chain: what a node made from the menu starts with (a saved node restores its own chain).
pub fn default_chain() -> Vec<NodeDesc> {
    vec![
        NodeDesc::biquad(1.0, -1.8, 0.81, 0.0025, 0.005, 0.0025),
        NodeDesc::distort(3.0, DSPFX_DIST_SOFT_CLIP),
        NodeDesc::reverb(0.5, 0.5),
        NodeDesc::biquad(1.0, -1.98, 0.9801, 0.99, -1.98, 0.99),
        NodeDesc::gain(0.5),
    ]
}

/// One mono channel through ANY fusable chain.  `Node` is implemented by hand rather than derived: the chain is data of the
/// node (saved under "chain", restored, replaceable while the graph runs), which `#[derive(DspNode)]`'s field attributes
/// cannot express.
pub struct GpuChain {
    id: NodeId,
    inputs: PortStorage,
    outputs: PortStorage,
    chain: Mutex<Vec<NodeDesc>>,
    engine: Mutex<Engine>,
}

impl GpuChain {
    fn build(id: NodeId, inputs: PortStorage, outputs: PortStorage, chain: Vec<NodeDesc>) -> Self {
        // a missing GPU is fatal, like every other failure on the reference's hot path (node.rs:173,271)
        let mut e = Engine::new(1, BUF_SIZE as u32, DSPFX_LINK_INTERNAL, 0).expect("libdspfx engine");
        e.set_chain(&chain).expect("chain");
        GpuChain { id, inputs, outputs, chain: Mutex::new(chain), engine: Mutex::new(e) }
    }
    /// A node for a chain of the host's choosing (the k effect nodes it replaces, in order).
    pub fn with_chain(id: NodeId, chain: Vec<NodeDesc>) -> Self {
        let (inputs, outputs) = (PortStorage::default(), PortStorage::default());    // like the derived `new` (lib.rs:213-219)
        inputs.add("in".to_owned());
        outputs.add("out".to_owned());
        Self::build(id, inputs, outputs, chain)
    }
    /// Replace the chain while the graph runs: dspfx_chain_set never waits for the run-time compiler (ABI 2), the new chain's
    /// state starts from zero like freshly created reference nodes (runtime.rs:319-362 re-creates nodes on a graph edit).
    pub fn set_chain(&self, chain: Vec<NodeDesc>) -> Result<(), super::engine::Error> {
        self.engine.lock().unwrap().set_chain(&chain)?;
        *self.chain.lock().unwrap() = chain;
        Ok(())
    }
}

impl Node for GpuChain {
    fn title(&self) -> &'static str { "GPU chain" }
    fn cfg_name(&self) -> &'static str { "gpu_chain" }
    fn description(&self) -> &'static str { "a chain of effect nodes evaluated as one kernel by libdspfx on an MI355X" }
    fn id(&self) -> NodeId { self.id }
    fn inputs(&self) -> &PortStorage { &self.inputs }
    fn outputs(&self) -> &PortStorage { &self.outputs }
    fn render(&self, ui: &mut eframe::egui::Ui) {
        ui.label(format!("{} nodes on the GPU", self.chain.lock().unwrap().len()));
    }
    fn save(&self) -> serde_json::Value {
        let chain: Vec<serde_json::Value> = self.chain.lock().unwrap().iter().map(NodeDesc::to_json).collect();
        serde_json::json!({ "id": self.id, "inputs": self.inputs, "outputs": self.outputs, "chain": chain })
    }
}

impl NodeStatic for GpuChain {
    fn new(id: NodeId) -> Self {
        Self::with_chain(id, default_chain())
    }
    fn restore(value: serde_json::Value) -> Self {
        let id = serde_json::from_value(value["id"].clone()).unwrap();
        let ins: PortStorage = serde_json::from_value(value["inputs"].clone()).unwrap();
        let outs: PortStorage = serde_json::from_value(value["outputs"].clone()).unwrap();
        let chain = NodeDesc::chain_from_json(&value["chain"]).unwrap_or_else(default_chain);
        Self::build(id, ins, outs, chain)
    }
}

impl SimpleNode for GpuChain {
    fn process(&self, inputs: ProcessInput, mut outputs: ProcessOutput) {
        let input = inputs.get("in").unwrap(); // already averaged by the Perform wrapper (node.rs:297)
        let output = outputs.get("out").unwrap();
        let mut e = self.engine.lock().unwrap();
        e.process_host(input, None, output, None, input.len() as u32).expect("dspfx_process_host");
    }
}

/// The form that uses the GPU: N identical chains evaluated together.  The host gathers the N pipes'
/// 128-frame blocks into one frame-major `[128][N]` buffer, the engine returns the N outputs and,
/// optionally, the Output node's mix of all of them (nodes/output.rs:215-249).
pub struct Bank {
    engine: Engine,
    gather: PinnedBlock,   // page-locked: dspfx_process_host then overlaps upload, kernel and download
    scatter: PinnedBlock,
    mix: Vec<f32>,
}

/// `len` f32 of page-locked host memory from `dspfx_host_alloc`.
pub(crate) struct PinnedBlock {
    ptr: *mut f32,
    len: usize,
}
unsafe impl Send for PinnedBlock {}
impl PinnedBlock {
    pub(crate) fn new(len: usize) -> Result<Self, super::engine::Error> {
        let mut p: *mut std::os::raw::c_void = std::ptr::null_mut();
        let rc = unsafe { super::ffi::dspfx_host_alloc(len * std::mem::size_of::<f32>(), &mut p) };
        if rc != super::ffi::DSPFX_OK {
            return Err(super::engine::Error { status: rc, message: "dspfx_host_alloc".into() });
        }
        let block = PinnedBlock { ptr: p as *mut f32, len };
        unsafe { std::ptr::write_bytes(block.ptr, 0, len) };
        Ok(block)
    }
    pub(crate) fn as_slice(&self) -> &[f32] { unsafe { std::slice::from_raw_parts(self.ptr, self.len) } }
    pub(crate) fn as_mut_slice(&mut self) -> &mut [f32] { unsafe { std::slice::from_raw_parts_mut(self.ptr, self.len) } }
}
impl Drop for PinnedBlock {
    fn drop(&mut self) {
        unsafe { super::ffi::dspfx_host_free(self.ptr as *mut _) };
    }
}

impl Bank {
    /// The handle the GUI thread keeps for this bank's sliders: `Send + Sync`, stores are queued by libdspfx and applied at
    /// the next block boundary, so the widget code never contends with `process` for the bank (lib.rs:487-492's Relaxed
    /// atomic store, across the FFI).
    pub fn params(&self) -> super::engine::ParamHandle {
        self.engine.params()
    }

    pub fn new(channels: u32, chain: &[NodeDesc], link_flags: u32) -> Result<Self, super::engine::Error> {
        let mut engine = Engine::new(channels, BUF_SIZE as u32, link_flags, 0)?;
        engine.set_chain(chain)?;
        let n = channels as usize * BUF_SIZE;
        Ok(Bank { engine, gather: PinnedBlock::new(n)?, scatter: PinnedBlock::new(n)?, mix: vec![0.0; BUF_SIZE] })
    }

    /// `inputs[c]` / `outputs[c]`: channel c's block (what one pipe of the reference holds).
    /// Returns the un-normalised mix bus (sum over channels per frame); divide by
    /// `dspfx_link_divisor(N)` -- or call `dspfx_mix_finish` on the device path -- for the Output hop.
    pub fn process(&mut self, inputs: &[&[f32]], outputs: &mut [&mut [f32]]) -> Result<&[f32], super::engine::Error> {
        let n = self.engine.channels() as usize;
        assert!(inputs.len() == n && outputs.len() == n);
        let frames = inputs[0].len();
        assert!(frames <= BUF_SIZE);
        let gather = self.gather.as_mut_slice();
        for (c, ch) in inputs.iter().enumerate() {
            assert_eq!(ch.len(), frames);
            for (f, v) in ch.iter().enumerate() {
                gather[f * n + c] = *v;
            }
        }
        let g = &self.gather.as_slice()[..frames * n];
        let s = &mut self.scatter.as_mut_slice()[..frames * n];
        self.engine.process_host(g, None, s, Some(&mut self.mix[..frames]), frames as u32)?;
        let scatter = self.scatter.as_slice();
        for (c, ch) in outputs.iter_mut().enumerate() {
            for (f, v) in ch.iter_mut().enumerate() {
                *v = scatter[f * n + c];
            }
        }
        Ok(&self.mix[..frames])
    }
}
