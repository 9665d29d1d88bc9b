#!/usr/bin/env python3
"""Check oracle/pin_kit (golden_dump.rs + cases.json + README.md) against the REAL reference tree, textually (no rustc in this image).

golden_dump.rs is a module for a maintainer to drop into the reference and run once; it has never been compiled.  What CAN be checked
here, in the build container (reads /root/reference; stores nothing of it):

  1. every `crate::` item the module imports is a `pub` item of the reference with the shape the module relies on
     (ids::{NodeId, PortId}::new, node::{collect_and_average, Node, Perform, BUF_SIZE}, PortStorage::{get_idxs, get_id},
     nodes::{Nodes, NODES, RESTORE} with their exact element types);
  2. every rivulet / biquad / tokio / serde call the module makes is spelled the way the reference's own sources spell it
     (reverb.rs, node.rs, runtime.rs, biquad.rs) and the crates + features it needs are in dsp-stuff/Cargo.toml;
  3. the README's main.rs lines fit: `mod nodes;` is there to sit next to, `fn main() -> color_eyre::Result<()>` is the signature the
     early `return Ok(())` needs, and the patched file stays balanced;
  4. every node of every case restores: its typename is a RESTORE key, its cfg holds EVERY field the derive's <Node>Config struct
     requires (id, inputs, outputs + each field tagged `save`; dsp-stuff-derive/src/lib.rs:233-337 -- a missing key panics in
     restore()), its port maps name exactly the ports the node declares (input = / output = / as_input sliders), select fields hold
     a variant of their enum; every `fresh` title is a NODES key; the links name ports that exist.
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DSPFX_REFERENCE", "/root/reference")
KIT = os.path.join(ROOT, "oracle", "pin_kit")
sys.path.insert(0, os.path.join(ROOT, "tools"))
from check_rust_recipe import balanced, block_after, strip_comments  # noqa: E402

README_LINES = {"mod": "mod golden_dump;", "call": ["if golden_dump::run_if_requested() {", "return Ok(());", "}"]}


def dsp_fields(src):
    """[(field name, attribute text)] of the #[dsp(...)]-tagged fields of the node struct in `src`"""
    out, i = [], 0
    while True:
        i = src.find("#[dsp(", i)
        if i < 0:
            return out
        depth, j = 0, i + 5
        while True:
            if src[j] == "(":
                depth += 1
            elif src[j] == ")":
                depth -= 1
                if depth == 0:
                    break
            j += 1
        attr = src[i + 6:j]
        m = re.match(r"\)\]\s*(?:pub\s+)?(\w+)\s*:", src[j:])
        if m:
            out.append((m.group(1), attr))
        i = j


def node_schema(path):
    src = strip_comments(open(path).read())
    hdr = re.search(r"#\[dsp\(\s*((?:[^()]|\([^()]*\))*?)\)\]\s*pub struct (\w+)", src, re.S)
    assert hdr, path
    head = hdr.group(1)
    ins = re.findall(r'\binput\s*=\s*"(\w+)"', head)
    outs = re.findall(r'\boutput\s*=\s*"(\w+)"', head)
    cfg_name = re.search(r'cfg_name\s*=\s*"(\w+)"', head).group(1)
    saved, as_input, selects = [], [], {}
    body = src[hdr.end():]
    for name, attr in dsp_fields(body):
        flags = re.sub(r'"[^"]*"', '""', attr)
        if re.search(r"(^|,)\s*save\b", flags):
            saved.append(name)
        if "as_input" in flags:
            as_input.append(name)
        if re.search(r"(^|,)\s*select\b", flags):
            ty = re.search(r"\b%s\s*:\s*Atomic<(\w+)>" % name, body)
            if ty:
                ea, eb = block_after(src, r"(?:pub\s+)?enum %s\s*\{" % ty.group(1))
                selects[name] = re.findall(r"^\s*(\w+)\s*,?\s*$", src[ea:eb], re.M)
    return dict(struct=hdr.group(2), cfg_name=cfg_name, inputs=ins + as_input, outputs=outs, saved=saved, selects=selects)


def main():
    rd = lambda *p: open(os.path.join(REF, *p)).read()
    node_rs, ids_rs, main_rs = rd("dsp-stuff", "src", "node.rs"), rd("dsp-stuff", "src", "ids.rs"), rd("dsp-stuff", "src", "main.rs")
    mod_rs, runtime_rs, cargo = rd("dsp-stuff", "src", "nodes", "mod.rs"), rd("dsp-stuff", "src", "runtime.rs"), rd("dsp-stuff", "Cargo.toml")
    reverb_rs, biquad_rs, output_rs = (rd("dsp-stuff", "src", "nodes", f) for f in ("reverb.rs", "biquad.rs", "output.rs"))
    kit = open(os.path.join(KIT, "golden_dump.rs")).read()
    code = strip_comments(kit)
    report = []
    assert balanced(kit), "golden_dump.rs: unbalanced brackets"

    # ---- 1. crate:: items
    for line in ("use crate::ids::{NodeId, PortId};", "use crate::node::{collect_and_average, Node, Perform, BUF_SIZE};", "use crate::nodes::{Nodes, NODES, RESTORE};"):
        assert line in code, "golden_dump.rs no longer imports: " + line
    assert re.search(r"pub struct NodeId\(usize\)", ids_rs) and re.search(r"pub struct PortId\(usize\)", ids_rs)
    assert re.search(r"pub fn new\(val: usize\) -> Self", ids_rs), "ids: `pub fn new(val: usize)` (NodeId::new / PortId::new)"
    assert re.search(r"derive\([^)]*\bHash\b[^)]*\)\]\s*pub struct PortId", ids_rs) and re.search(r"derive\([^)]*\bPartialEq\b[^)]*\)\]\s*pub struct PortId", ids_rs)
    m = re.search(r"pub async fn collect_and_average\(\s*output: &mut \[f32\],\s*input: &mut \[&mut splittable::View<Source<f32>>\],\s*\) -> bool", node_rs)
    assert m, "node.rs: collect_and_average's signature changed"
    assert re.search(r"pub const BUF_SIZE: usize = 128;", node_rs)
    assert re.search(r"pub trait Node\b", node_rs) and re.search(r"pub trait Perform\s*:\s*Node", node_rs)
    assert re.search(r"fn inputs\(&self\) -> &crate::node::PortStorage;", node_rs) and re.search(r"fn outputs\(&self\) -> &crate::node::PortStorage;", node_rs)
    assert re.search(r"async fn perform\(\s*&self,\s*inputs: crate::node::NodeInputs<'_, '_, '_>,\s*outputs: crate::node::NodeOutputs<'_, '_, '_>,\s*\);", node_rs)
    assert re.search(r"pub type NodeInputs<'a, 'b, 'c> = &'a mut \[&'b mut \[&'c mut splittable::View<Source<f32>>\]\];", node_rs)
    assert re.search(r"pub type NodeOutputs<'a, 'b, 'c> = &'a mut \[&'b mut \[&'c mut Sink<f32>\]\];", node_rs)
    assert re.search(r"pub fn get_idxs\(&self\) -> HashMap<PortId, usize>", node_rs) and re.search(r"pub fn get_id\(&self, name: &str\) -> Option<PortId>", node_rs)
    assert re.search(r"pub enum Nodes\s*\{", mod_rs)
    assert "pub static NODES: &[(&str, fn(NodeId) -> Arc<Nodes>)] = &[" in mod_rs and "pub static RESTORE: &[(&str, fn(serde_json::Value) -> Arc<Nodes>)] = &[" in mod_rs
    assert re.search(r"#\[enum_dispatch::enum_dispatch\(Perform\)\]\s*#\[enum_dispatch::enum_dispatch\(Node\)\]\s*pub enum Nodes", mod_rs)
    report.append("crate::{ids, node, nodes} items the module imports exist with the shapes it relies on")

    # ---- 2. third-party calls, spelled as the reference spells them
    uses = re.search(r"use rivulet::\{(.*?)\};", code, re.S).group(1)
    ref_uses = re.search(r"use rivulet::\{(.*?)\};", reverb_rs, re.S).group(1)
    for item in re.findall(r"\b(Sink|Source|splittable|SplittableView|View|ViewMut)\b", uses):
        assert re.search(r"\b%s\b" % item, ref_uses), "rivulet::%s is not something reverb.rs imports" % item
    for call, where, text in (("rivulet::circular_buffer::<f32>(8192)", "runtime.rs", runtime_rs), ("source.into_view()", "runtime.rs", runtime_rs),
                              (".try_grant(", "reverb.rs", reverb_rs), (".view_mut().fill(0.0)", "reverb.rs", reverb_rs), (".view().len()", "reverb.rs", reverb_rs),
                              (".grant(BUF_SIZE).await.unwrap()", "node.rs", node_rs), (".view_mut()[..BUF_SIZE].copy_from_slice(", "node.rs", node_rs),
                              (".release(BUF_SIZE)", "node.rs", node_rs), ("collect_and_average(&mut buf, ", "output.rs", output_rs),
                              (".perform(&mut ", "runtime.rs", runtime_rs), (".as_mut_slice()", "runtime.rs", runtime_rs),
                              ("use biquad::{Biquad as _, DirectForm1};", "biquad.rs", biquad_rs), ("DirectForm1::<f32>::new(", "biquad.rs", biquad_rs),
                              ("biquad::Coefficients {", "biquad.rs", biquad_rs), (".run(*", "biquad.rs", biquad_rs)):
        assert call in code, "golden_dump.rs no longer contains `%s`" % call
        assert call in text, "`%s` is not how %s spells it" % (call, where)
    for dep, feat in (("tokio", '"rt"'), ("serde_json", None), ("serde", '"derive"'), ("rivulet", None), ("biquad", None)):
        m = re.search(r"^%s\s*=\s*(.*)$" % re.escape(dep), cargo, re.M)
        assert m and (feat is None or feat in m.group(1)), "Cargo.toml: %s %s" % (dep, feat or "")
    assert "tokio::runtime::Builder::new_current_thread()" in code
    report.append("rivulet / biquad / tokio / serde usage is spelled as in reverb.rs, node.rs, runtime.rs, output.rs, biquad.rs; crates + features present")

    # ---- 3. the README's main.rs lines
    readme = open(os.path.join(KIT, "README.md")).read()
    for l in [README_LINES["mod"]] + README_LINES["call"]:
        assert l in readme, "README.md does not show the line: " + l
    assert re.search(r"^mod nodes;", main_rs, re.M) and "mod golden_dump;" not in main_rs
    assert re.search(r"fn main\(\) -> color_eyre::Result<\(\)> \{", main_rs), "main() no longer returns color_eyre::Result<()>"
    patched = re.sub(r"^mod nodes;", "mod nodes;\n" + README_LINES["mod"], main_rs, count=1, flags=re.M)
    patched = re.sub(r"(fn main\(\) -> color_eyre::Result<\(\)> \{\n)", r"\1    " + "\n    ".join(README_LINES["call"]) + "\n", patched, count=1)
    assert balanced(patched) and "golden_dump::run_if_requested()" in patched
    assert "pub fn run_if_requested() -> bool" in code
    report.append("main.rs: `mod golden_dump;` + the early return fit next to `mod nodes;` / at the top of main(); balanced")

    # ---- 4. every case restores
    ra, rb = block_after(mod_rs, r"pub static RESTORE: &\[\(&str, fn\(serde_json::Value\) -> Arc<Nodes>\)\] = &\[")
    na, nb = block_after(mod_rs, r"pub static NODES: &\[\(&str, fn\(NodeId\) -> Arc<Nodes>\)\] = &\[")
    restore = dict(re.findall(r'\("(\w+)", \|v\| \{?\s*Arc::new\(Nodes::from\((\w+)::restore\(v\)\)\)', mod_rs[ra:rb]))
    titles = dict(re.findall(r'\("([^"]+)", \|id\| \{?\s*Arc::new\(Nodes::from\((\w+)::new\(id\)\)\)', mod_rs[na:nb]))
    files = {}
    for f in os.listdir(os.path.join(REF, "dsp-stuff", "src", "nodes")):
        if f.endswith(".rs") and f not in ("mod.rs", "input.rs", "output.rs"):
            if "dsp_stuff_derive::DspNode" not in open(os.path.join(REF, "dsp-stuff", "src", "nodes", f)).read():
                continue                     # (a node with a hand-written NodeStatic: none of the cases uses one)
            sc = node_schema(os.path.join(REF, "dsp-stuff", "src", "nodes", f))
            files[sc["struct"]] = sc
    cases = json.load(open(os.path.join(KIT, "cases.json")))
    assert cases["schema"] == 1
    seen = set()
    for c in cases["cases"]:
        ids = {n["id"]: n for n in c["doc"]["nodes"]}
        assert sum(n["typename"] == "output" for n in ids.values()) == 1 and sum(n["typename"] == "input" for n in ids.values()) <= 1, c["name"]
        for n in ids.values():
            if n["typename"] in ("input", "output"):
                continue
            assert n["typename"] in restore, "%s: typename %r is not a RESTORE key" % (c["name"], n["typename"])
            sc = files[restore[n["typename"]]]
            need = {"id", "inputs", "outputs"} | set(sc["saved"])
            assert need <= set(n["cfg"]), "%s: %s cfg lacks %s (restore() would panic)" % (c["name"], n["typename"], sorted(need - set(n["cfg"])))
            assert n["cfg"]["id"] == n["id"]
            assert set(n["cfg"]["inputs"]) == set(sc["inputs"]), "%s: %s input ports %s, the node declares %s" % (c["name"], n["typename"], sorted(n["cfg"]["inputs"]), sc["inputs"])
            assert set(n["cfg"]["outputs"]) == set(sc["outputs"]), (c["name"], n["typename"], n["cfg"]["outputs"], sc["outputs"])
            for f, variants in sc["selects"].items():
                assert n["cfg"][f] in variants, "%s: %s.%s = %r is not one of %s" % (c["name"], n["typename"], f, n["cfg"][f], variants)
            seen.add(n["typename"])
        for nid, title in c["fresh"].items():
            assert title in titles and int(nid) in ids, "%s: fresh title %r" % (c["name"], title)
            assert titles[title] == restore[ids[int(nid)]["typename"]], "%s: the fresh node's title and its document typename name different structs" % c["name"]
        for l in c["doc"]["links"]:
            (ln, lp), (rn, rp) = l["lhs"], l["rhs"]
            assert lp in ids[ln]["cfg"]["outputs"].values() and rp in ids[rn]["cfg"]["inputs"].values(), "%s: a link names a port its node does not have" % c["name"]
        assert all(len(ch) % 128 == 0 and len(ch) == len(c["x"][0]) for ch in c["x"]), "%s: whole blocks only" % c["name"]
    report.append("cases.json: %d cases, node types %s -- every cfg holds every field restore() unwraps, ports and select variants as declared" % (
        len(cases["cases"]), sorted(seen)))
    for r in report:
        print("ok  " + r)
    return 0


if __name__ == "__main__":
    sys.exit(main())
