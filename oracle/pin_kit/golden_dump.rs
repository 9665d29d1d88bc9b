//! golden_dump.rs -- the reference side of the parity pin (dsp-stuff_amd's `oracle/pin_kit/`).
//!
//! NOT part of the engine and never built by it: a module a maintainer WITH the Rust toolchain drops into the
//! reference tree (`dsp-stuff/src/golden_dump.rs`, declared as `mod golden_dump;` next to `mod nodes;` in main.rs,
//! plus the three lines at the top of `main()` that `oracle/pin_kit/README.md` shows).  It runs the REAL nodes --
//! built by the reference's own `RESTORE` / `NODES` tables (nodes/mod.rs:65-123), wired with real rivulet pipes
//! exactly like `LinkInstance::new` (runtime.rs:566-578) and `compute_inputs_for` (runtime.rs:159-196), driven
//! block by block through `Perform::perform` (node.rs:148-160; for every SimpleNode that is the blanket wrapper
//! node.rs:267-352, so `collect_and_average`, the `/ (0.0001 + n)` hop, the zeroed buffers and the fan-out copy are
//! the reference's own code) -- over the cases of `cases.json` (DSPConfig documents + input samples as f32 bit
//! patterns, exported from the repository's golden vectors by `oracle/pin_kit/export_cases.py`) and writes every
//! output sample as a bit pattern to `pin_out.json`.  `tools/compare_pin.py` then holds that file against the
//! golden vectors with the bars of the parity tests.  Two facts the CPU restatement could only state "as recalled"
//! are emitted as data as well: the granted view length of a fresh rivulet ring (reverb.rs:60-68) and the
//! operation order of `biquad::DirectForm1::run` (biquad.rs:87).
//!
//!     DSPFX_GOLDEN_DUMP=cases.json DSPFX_GOLDEN_OUT=pin_out.json cargo run --release
//!
//! The Input and Output nodes of a document are not instantiated (they open audio devices): the harness feeds the
//! Input node's links with the case's samples and reads the Output node's links with `collect_and_average` -- the
//! very call `Output::perform` makes (nodes/output.rs:215-223).  Nodes run one at a time in the document's
//! topological order on a current-thread tokio runtime: every pipe holds 8192 samples and every node moves exactly
//! BUF_SIZE per call, so no `grant` ever has to wait.

use std::collections::HashMap;
use std::sync::Arc;

use rivulet::{
    circular_buffer::{Sink, Source},
    splittable, SplittableView, View, ViewMut,
};
use serde::{Deserialize, Serialize};

use crate::ids::{NodeId, PortId};
use crate::node::{collect_and_average, Node, Perform, BUF_SIZE};
use crate::nodes::{Nodes, NODES, RESTORE};

type Src = splittable::View<Source<f32>>;

#[derive(Deserialize)]
struct CaseFile {
    schema: u32,
    cases: Vec<Case>,
}

#[derive(Deserialize)]
struct Case {
    name: String,
    /// the DSPConfig document as the reference's File > Save writes it (runtime.rs:44-48, 466-479)
    doc: Doc,
    /// node id -> menu title: build this node with `NodeStatic::new` (fresh from the menu) instead of `restore`
    #[serde(default)]
    fresh: HashMap<String, String>,
    /// the Input node's signal, one vector of f32 bit patterns per channel; every channel is an independent run
    x: Vec<Vec<u32>>,
}

#[derive(Deserialize)]
struct Doc {
    nodes: Vec<DocNode>,
    links: Vec<DocLink>,
}

#[derive(Deserialize)]
struct DocNode {
    id: usize,
    typename: String,
    cfg: serde_json::Value,
}

#[derive(Deserialize)]
struct DocLink {
    lhs: (usize, usize),
    rhs: (usize, usize),
}

#[derive(Serialize)]
struct CaseOut {
    name: String,
    /// the Output node's signal per channel, f32 bit patterns
    y: Vec<Vec<u32>>,
}

#[derive(Serialize)]
struct Probes {
    /// (requested length, `view().len()` after `try_grant(requested)` on a fresh ring): reverb.rs:60-68's num_zeros
    rivulet_view_len: Vec<(usize, usize)>,
    /// DirectForm1::<f32>::run on an 8-sample probe whose result differs between the candidate operation orders
    biquad_probe: BiquadProbe,
}

#[derive(Serialize)]
struct BiquadProbe {
    coeffs: [u32; 5], // a1 a2 b0 b1 b2
    x: Vec<u32>,
    y: Vec<u32>,
}

#[derive(Serialize)]
struct OutFile {
    schema: u32,
    buf_size: usize,
    results: Vec<CaseOut>,
    probes: Probes,
}

/// name of the port with this id in a saved cfg's "inputs" / "outputs" map
fn port_name(cfg: &serde_json::Value, which: &str, id: usize) -> Option<String> {
    cfg.get(which)?
        .as_object()?
        .iter()
        .find(|(_, v)| v.as_u64() == Some(id as u64))
        .map(|(k, _)| k.clone())
}

fn build_node(n: &DocNode, fresh: &HashMap<String, String>) -> Arc<Nodes> {
    if let Some(title) = fresh.get(&n.id.to_string()) {
        let (_, make) = NODES.iter().find(|(t, _)| *t == title.as_str()).expect("menu title");
        make(NodeId::new(n.id))
    } else {
        // what runtime.rs:620-632 (NodeInstance::restore) does with a saved node
        let (_, restorer) = RESTORE.iter().find(|(t, _)| *t == n.typename.as_str()).expect("typename");
        restorer(n.cfg.clone())
    }
}

/// document order is not execution order: sort so that every node comes after the nodes that feed it
fn topo_order(doc: &Doc) -> Vec<usize> {
    let mut done: Vec<usize> = Vec::new();
    let ids: Vec<usize> = doc.nodes.iter().map(|n| n.id).collect();
    while done.len() < ids.len() {
        let before = done.len();
        for id in &ids {
            if done.contains(id) {
                continue;
            }
            if doc.links.iter().filter(|l| l.rhs.0 == *id).all(|l| done.contains(&l.lhs.0)) {
                done.push(*id);
            }
        }
        assert!(done.len() > before, "the document has a cycle");
    }
    done
}

async fn run_channel(case: &Case, x: &[f32]) -> Vec<f32> {
    let doc = &case.doc;
    let input_id = doc.nodes.iter().find(|n| n.typename == "input").map(|n| n.id);
    let output_id = doc.nodes.iter().find(|n| n.typename == "output").expect("output node").id;
    let by_id: HashMap<usize, &DocNode> = doc.nodes.iter().map(|n| (n.id, n)).collect();
    let nodes: HashMap<usize, Arc<Nodes>> = doc
        .nodes
        .iter()
        .filter(|n| n.typename != "input" && n.typename != "output")
        .map(|n| (n.id, build_node(n, &case.fresh)))
        .collect();

    // one pipe per link, as LinkInstance::new makes them (runtime.rs:566-578)
    let mut sinks: Vec<Option<Sink<f32>>> = Vec::new();
    let mut sources: Vec<Option<Src>> = Vec::new();
    for _ in &doc.links {
        let (sink, source) = rivulet::circular_buffer::<f32>(8192);
        sinks.push(Some(sink));
        sources.push(Some(source.into_view()));
    }

    // per node: links of every input / output port, ports in local-index order (runtime.rs:159-196, 198-235)
    let mut in_ports: HashMap<usize, Vec<Vec<usize>>> = HashMap::new();
    let mut out_ports: HashMap<usize, Vec<Vec<usize>>> = HashMap::new();
    for (id, inst) in &nodes {
        let cfg = &by_id[id].cfg;
        let mut ins: Vec<(usize, PortId)> = inst.inputs().get_idxs().into_iter().map(|(p, i)| (i, p)).collect();
        ins.sort_by_key(|(i, _)| *i);
        let mut outs: Vec<(usize, PortId)> = inst.outputs().get_idxs().into_iter().map(|(p, i)| (i, p)).collect();
        outs.sort_by_key(|(i, _)| *i);
        let links_into = |pid: PortId| -> Vec<usize> {
            doc.links
                .iter()
                .enumerate()
                .filter(|(_, l)| {
                    l.rhs.0 == *id
                        && port_name(cfg, "inputs", l.rhs.1).and_then(|n| inst.inputs().get_id(&n)) == Some(pid)
                })
                .map(|(k, _)| k)
                .collect()
        };
        let links_from = |pid: PortId| -> Vec<usize> {
            doc.links
                .iter()
                .enumerate()
                .filter(|(_, l)| {
                    l.lhs.0 == *id
                        && port_name(cfg, "outputs", l.lhs.1).and_then(|n| inst.outputs().get_id(&n)) == Some(pid)
                })
                .map(|(k, _)| k)
                .collect()
        };
        in_ports.insert(*id, ins.iter().map(|(_, p)| links_into(*p)).collect());
        out_ports.insert(*id, outs.iter().map(|(_, p)| links_from(*p)).collect());
    }
    let fed_by_input: Vec<usize> = doc
        .links
        .iter()
        .enumerate()
        .filter(|(_, l)| Some(l.lhs.0) == input_id)
        .map(|(k, _)| k)
        .collect();
    let into_output: Vec<usize> = doc
        .links
        .iter()
        .enumerate()
        .filter(|(_, l)| l.rhs.0 == output_id)
        .map(|(k, _)| k)
        .collect();
    let order: Vec<usize> = topo_order(doc).into_iter().filter(|id| nodes.contains_key(id)).collect();

    let mut y = Vec::with_capacity(x.len());
    for block in x.chunks(BUF_SIZE) {
        assert_eq!(block.len(), BUF_SIZE, "cases hold whole blocks");
        // the Input node: one block into every link it feeds (input.rs does the same with the device's samples)
        for &k in &fed_by_input {
            let sink = sinks[k].as_mut().unwrap();
            sink.grant(BUF_SIZE).await.unwrap();
            sink.view_mut()[..BUF_SIZE].copy_from_slice(block);
            sink.release(BUF_SIZE);
        }
        for id in &order {
            let inst = &nodes[id];
            let mut in_pipes: Vec<Vec<(usize, Src)>> = in_ports[id]
                .iter()
                .map(|ls| ls.iter().map(|&k| (k, sources[k].take().unwrap())).collect())
                .collect();
            let mut out_pipes: Vec<Vec<(usize, Sink<f32>)>> = out_ports[id]
                .iter()
                .map(|ls| ls.iter().map(|&k| (k, sinks[k].take().unwrap())).collect())
                .collect();
            {
                let mut in_refs: Vec<Vec<&mut Src>> = in_pipes
                    .iter_mut()
                    .map(|p| p.iter_mut().map(|(_, v)| v).collect())
                    .collect();
                let mut out_refs: Vec<Vec<&mut Sink<f32>>> = out_pipes
                    .iter_mut()
                    .map(|p| p.iter_mut().map(|(_, v)| v).collect())
                    .collect();
                let mut in_slices: Vec<&mut [&mut Src]> = in_refs.iter_mut().map(|v| v.as_mut_slice()).collect();
                let mut out_slices: Vec<&mut [&mut Sink<f32>]> =
                    out_refs.iter_mut().map(|v| v.as_mut_slice()).collect();
                // runtime.rs:718-728, one turn of the loop
                inst.perform(&mut in_slices, &mut out_slices).await;
            }
            for port in in_pipes {
                for (k, v) in port {
                    sources[k] = Some(v);
                }
            }
            for port in out_pipes {
                for (k, v) in port {
                    sinks[k] = Some(v);
                }
            }
        }
        // the Output node: nodes/output.rs:215-223, then the release of :237-246
        let mut buf = [0.0f32; BUF_SIZE];
        let mut taken: Vec<(usize, Src)> = into_output.iter().map(|&k| (k, sources[k].take().unwrap())).collect();
        {
            let mut refs: Vec<&mut Src> = taken.iter_mut().map(|(_, v)| v).collect();
            collect_and_average(&mut buf, &mut refs).await;
            for v in refs.iter_mut() {
                if v.view().len() >= BUF_SIZE {
                    v.release(BUF_SIZE);
                }
            }
        }
        for (k, v) in taken {
            sources[k] = Some(v);
        }
        y.extend_from_slice(&buf);
    }
    y
}

fn probes() -> Probes {
    let mut lens = Vec::new();
    for n in [128usize, 1000, 1024, 4800, 24000, 24576, 48000] {
        // reverb.rs:60-68, verbatim
        let (mut new_sink, _new_source) = rivulet::circular_buffer::<f32>(n);
        let _ = new_sink.try_grant(n);
        new_sink.view_mut().fill(0.0);
        lens.push((n, new_sink.view().len()));
    }
    use biquad::{Biquad as _, DirectForm1};
    // coefficients and samples chosen (export_cases.py: biquad_probe) so that the candidate orders of the five products'
    // sum -- left to right, (b-terms) - (a-terms), fused -- round differently in f32
    let c = [-1.7990895f32, 0.81783146, 0.0046772375, 0.009354475, 0.0046772375];
    let x = [
        0.18588203191757202f32, -0.4798051118850708, 0.6797630190849304, -0.9063479900360107, 0.3373520076274872,
        0.7703438401222229, -0.12345679104328156, 0.5555555820465088,
    ];
    let mut f = DirectForm1::<f32>::new(biquad::Coefficients { a1: c[0], a2: c[1], b0: c[2], b1: c[3], b2: c[4] });
    let y: Vec<u32> = x.iter().map(|v| f.run(*v).to_bits()).collect();
    Probes {
        rivulet_view_len: lens,
        biquad_probe: BiquadProbe {
            coeffs: [c[0].to_bits(), c[1].to_bits(), c[2].to_bits(), c[3].to_bits(), c[4].to_bits()],
            x: x.iter().map(|v| v.to_bits()).collect(),
            y,
        },
    }
}

/// Called from the top of `main()` when DSPFX_GOLDEN_DUMP names the case file; returns true when it ran.
pub fn run_if_requested() -> bool {
    let Ok(path) = std::env::var("DSPFX_GOLDEN_DUMP") else {
        return false;
    };
    let out_path = std::env::var("DSPFX_GOLDEN_OUT").unwrap_or_else(|_| "pin_out.json".to_owned());
    let file: CaseFile = serde_json::from_str(&std::fs::read_to_string(&path).expect("case file")).expect("case json");
    assert_eq!(file.schema, 1, "case file schema");
    let rt = tokio::runtime::Builder::new_current_thread().build().unwrap();
    let mut results = Vec::new();
    for case in &file.cases {
        let mut y = Vec::new();
        for ch in &case.x {
            let x: Vec<f32> = ch.iter().map(|b| f32::from_bits(*b)).collect();
            let out = rt.block_on(run_channel(case, &x));
            y.push(out.iter().map(|v| v.to_bits()).collect());
        }
        eprintln!("golden_dump: {} ({} channel(s) x {} frames)", case.name, case.x.len(), case.x[0].len());
        results.push(CaseOut { name: case.name.clone(), y });
    }
    let out = OutFile { schema: 1, buf_size: BUF_SIZE, results, probes: probes() };
    std::fs::write(&out_path, serde_json::to_string(&out).unwrap()).expect("write the result file");
    eprintln!("golden_dump: {} cases -> {}", file.cases.len(), out_path);
    true
}
