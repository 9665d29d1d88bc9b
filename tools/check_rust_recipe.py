#!/usr/bin/env python3
"""Check host/rust's integration recipe against the REAL reference tree.

The Rust shim cannot be compiled in this image (no rustc; the reference needs nightly and ~400 crates), so nothing had ever
checked that the lines `host/rust/README.md` tells a maintainer to add would fit where it says they go, or that the traits and
types the shim implements exist in the reference with those signatures.  This script does, textually, in the build container
(it reads /root/reference, which does not exist on the GPU box; nothing of the reference is stored in this repository):

  1. every method of `impl Node / NodeStatic / SimpleNode for GpuChain / GpuBank` is a method of that trait in
     dsp-stuff/src/node.rs with the same parameter list and return type, and every REQUIRED trait method is implemented;
  2. every name the shim takes from `crate::` (`ids::NodeId`, `node::*`: PortStorage and its `add` / `default`, ProcessInput /
     ProcessOutput and their `get`, BUF_SIZE) is a `pub` item there;
  3. the anchors of the recipe exist in dsp-stuff/src/nodes/mod.rs and main.rs -- the `use self::{...}` list, the `pub mod` list,
     `pub enum Nodes {...}` under enum_dispatch(Perform) + enum_dispatch(Node), `pub static NODES`, `pub static RESTORE`, main.rs'
     `mod nodes;` -- and the inserted lines (RECIPE below = the README's table) have the shape of their neighbours, collide with
     no existing variant / menu title / cfg_name, leave every bracket balanced, and the RESTORE keys equal the shims' cfg_name().

Exit status 0 and one line per check, or an AssertionError naming what no longer fits.
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DSPFX_REFERENCE", "/root/reference")
SHIM = os.path.join(ROOT, "host", "rust", "src")

# what host/rust/README.md tells the maintainer to add (kept in step with it by tests/test_rust_recipe_cpu.py)
RECIPE = {
    "main_mod": "mod gpu;",
    "use": "use crate::gpu::{gpu_bank::GpuBank, gpu_chain::GpuChain};",
    "variants": ["GpuChain,", "GpuBank,"],
    "nodes": ['("GPU chain", |id| Arc::new(Nodes::from(GpuChain::new(id)))),', '("GPU bank", |id| Arc::new(Nodes::from(GpuBank::new(id)))),'],
    "restore": ['("gpu_chain", |v| Arc::new(Nodes::from(GpuChain::restore(v)))),', '("gpu_bank", |v| Arc::new(Nodes::from(GpuBank::restore(v)))),'],
}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def balanced(text):
    text = re.sub(r'"(?:\\.|[^"\\])*"', '""', strip_comments(text))
    text = re.sub(r"'(?:\\.|[^'\\])'", "''", text)
    stack = []
    pairs = {")": "(", "]": "[", "}": "{"}
    for ch in text:
        if ch in "([{":
            stack.append(ch)
        elif ch in ")]}":
            if not stack or stack.pop() != pairs[ch]:
                return False
    return not stack


def block_after(text, header_re):
    """The brace / bracket block that follows the first match of header_re: (start, end) of its contents."""
    m = re.search(header_re, text)
    assert m, "anchor not found: %s" % header_re
    i = m.end() - 1
    open_ch = text[i]
    close_ch = {"{": "}", "[": "]", "(": ")"}[open_ch]
    depth = 0
    for j in range(i, len(text)):
        if text[j] == open_ch:
            depth += 1
        elif text[j] == close_ch:
            depth -= 1
            if depth == 0:
                return i + 1, j
    raise AssertionError("unterminated block after %s" % header_re)


def norm(sig):
    return re.sub(r"\s+", " ", sig).replace("crate::ids::", "").replace("crate::node::", "").replace("egui::", "").replace("eframe::", "").strip()


def trait_methods(src, name):
    a, b = block_after(src, r"pub trait %s\b[^{]*\{" % name)
    body = strip_comments(src[a:b])
    out = {}
    for m in re.finditer(r"(?:async\s+)?fn\s+(\w+)\s*\(([^)]*)\)\s*(->\s*[^;{]+?)?\s*(where[^;{]*)?([;{])", body, re.S):
        out[m.group(1)] = {"params": norm(m.group(2)), "ret": norm(m.group(3) or ""), "required": m.group(5) == ";"}
    return out


def impl_methods(src, trait, ty):
    a, b = block_after(src, r"impl\s+%s\s+for\s+%s\s*\{" % (trait, ty))
    body = strip_comments(src[a:b])
    out = {}
    depth = 0
    for m in re.finditer(r"fn\s+(\w+)\s*\(([^)]*)\)\s*(->\s*[^{]+?)?\s*\{", body, re.S):
        if body[:m.start()].count("{") - body[:m.start()].count("}") == 0:      # a method of the impl, not a nested fn
            out[m.group(1)] = {"params": norm(m.group(2)), "ret": norm(m.group(3) or "")}
    return out


def main():
    node_rs = open(os.path.join(REF, "dsp-stuff", "src", "node.rs")).read()
    ids_rs = open(os.path.join(REF, "dsp-stuff", "src", "ids.rs")).read()
    mod_rs = open(os.path.join(REF, "dsp-stuff", "src", "nodes", "mod.rs")).read()
    main_rs = open(os.path.join(REF, "dsp-stuff", "src", "main.rs")).read()
    shim = {f: open(os.path.join(SHIM, f)).read() for f in ("gpu_chain.rs", "gpu_bank.rs", "engine.rs", "mod.rs", "ffi.rs")}
    report = []

    # ---- 1. trait methods
    assert re.search(r"pub trait Node\s*:\s*Send\s*\+\s*Sync", node_rs), "trait Node is no longer Send + Sync"
    for trait in ("Node", "NodeStatic", "SimpleNode"):
        want = trait_methods(node_rs, trait)
        assert want, "trait %s not found in node.rs" % trait
        for ty, f in (("GpuChain", "gpu_chain.rs"), ("GpuBank", "gpu_bank.rs")):
            got = impl_methods(shim[f], trait, ty)
            for name, sig in got.items():
                assert name in want, "%s implements %s::%s, which the reference's trait does not have" % (ty, trait, name)
                w = want[name]
                g_params = re.sub(r"\bmut\s+(\w+\s*:)", r"\1", sig["params"])          # `mut x: T` is the same parameter as `x: T`
                assert g_params == w["params"], "%s::%s(%s) != reference %s::%s(%s)" % (ty, name, sig["params"], trait, name, w["params"])
                assert sig["ret"] == w["ret"], "%s::%s returns %r, the reference %r" % (ty, name, sig["ret"], w["ret"])
            missing = [n for n, w in want.items() if w["required"] and n not in got]
            assert not missing, "%s lacks required %s methods %s" % (ty, trait, missing)
            report.append("%s: impl %s matches node.rs (%d methods)" % (ty, trait, len(got)))
    # the Send + Sync bound: every field type of the two nodes is Send + Sync by construction or by an explicit unsafe impl
    assert "unsafe impl Send for NodeDesc {}" in shim["engine.rs"] and "unsafe impl Sync for NodeDesc {}" in shim["engine.rs"], \
        "NodeDesc embeds a raw pointer: without these impls GpuChain / GpuBank are not Send + Sync (trait Node requires both)"
    report.append("NodeDesc is Send + Sync (the chain is kept in a Mutex<Vec<NodeDesc>>)")

    # ---- 2. crate:: names
    assert re.search(r"pub struct NodeId\b", ids_rs), "ids::NodeId"
    for item, pat in (("PortStorage", r"pub struct PortStorage\b"), ("PortStorage::add", r"pub fn add\(&self, name: String\)"),
                      ("PortStorage: Default", r"#\[derive\([^)]*Default[^)]*\)\]\s*pub struct PortStorage\b"),
                      ("PortStorage: Serialize", r"impl Serialize for PortStorage"), ("PortStorage: Deserialize", r"impl<'de> Deserialize<'de> for PortStorage"),
                      ("ProcessInput", r"pub struct ProcessInput\b"), ("ProcessOutput", r"pub struct ProcessOutput\b"),
                      ("BUF_SIZE", r"pub const BUF_SIZE: usize = 128;"), ("blanket Perform", r"impl<T: SimpleNode[^>]*>\s*Perform for T")):
        assert re.search(pat, node_rs), "node.rs no longer has %s (%s)" % (item, pat)
    for ty in ("ProcessInput", "ProcessOutput"):
        a, b = block_after(node_rs, r"impl[^{]*\b%s\b[^{]*\{" % ty)
        assert re.search(r"pub fn get\b", node_rs[a:b]), "%s::get" % ty
    report.append("crate::ids::NodeId, node::{PortStorage(+add, Default, serde), ProcessInput/Output::get, BUF_SIZE, blanket Perform}: present")

    # ---- 3. anchors and inserted lines
    assert re.search(r"^mod nodes;", main_rs, re.M), "main.rs: `mod nodes;`"
    patched_main = re.sub(r"^mod nodes;", "mod nodes;\n" + RECIPE["main_mod"], main_rs, count=1, flags=re.M)
    assert balanced(patched_main) and "mod gpu;" not in main_rs
    assert re.search(r"use self::\{", mod_rs) and re.search(r"^pub mod \w+;", mod_rs, re.M)
    assert re.search(r"#\[enum_dispatch::enum_dispatch\(Perform\)\]\s*#\[enum_dispatch::enum_dispatch\(Node\)\]\s*pub enum Nodes\s*\{", mod_rs), \
        "enum Nodes is no longer dispatched over Perform and Node"
    ea, eb = block_after(mod_rs, r"pub enum Nodes\s*\{")
    variants = re.findall(r"^\s*(\w+),\s*$", strip_comments(mod_rs[ea:eb]), re.M)
    assert len(variants) >= 15, variants
    for v in RECIPE["variants"]:
        assert re.fullmatch(r"\w+,", v) and v[:-1] not in variants, "variant %s collides or is malformed" % v
    na, nb = block_after(mod_rs, r"pub static NODES: &\[\(&str, fn\(NodeId\) -> Arc<Nodes>\)\] = &\[")
    ra, rb = block_after(mod_rs, r"pub static RESTORE: &\[\(&str, fn\(serde_json::Value\) -> Arc<Nodes>\)\] = &\[")
    entry = r'\("([^"]+)", \|(\w+)\| \{?\s*Arc::new\(Nodes::from\((\w+)::(new|restore)\(\2\)\)\)\s*\}?\),'
    titles = [m.group(1) for m in re.finditer(entry, mod_rs[na:nb])]
    keys = [m.group(1) for m in re.finditer(entry, mod_rs[ra:rb])]
    assert len(titles) == len(keys) >= 15 and len(titles) == len(variants), (len(titles), len(keys), len(variants))
    cfg_names = {}
    for ty, f in (("GpuChain", "gpu_chain.rs"), ("GpuBank", "gpu_bank.rs")):
        m = re.search(r'fn cfg_name\(&self\) -> &\'static str \{ "(\w+)" \}', shim[f])
        t = re.search(r'fn title\(&self\) -> &\'static str \{ "([^"]+)" \}', shim[f])
        assert m and t, "%s: cfg_name / title" % ty
        cfg_names[ty] = (m.group(1), t.group(1))
    for line, kind, taken in [(l, "new", titles) for l in RECIPE["nodes"]] + [(l, "restore", keys) for l in RECIPE["restore"]]:
        m = re.fullmatch(entry, line)
        assert m and m.group(4) == kind, "inserted line does not have the shape of its neighbours: %s" % line
        assert m.group(1) not in taken, "%s collides with an existing entry" % m.group(1)
        ty = m.group(3)
        assert ty + "," in RECIPE["variants"]
        assert m.group(1) == cfg_names[ty][0 if kind == "restore" else 1], "%s: the table says %r, the shim's %s() %r" % (
            ty, m.group(1), "cfg_name" if kind == "restore" else "title", cfg_names[ty][0 if kind == "restore" else 1])
    # RESTORE is looked up by the saved node's "typename" = cfg_name(): find that lookup so the key is what restore() is given
    runtime_rs = open(os.path.join(REF, "dsp-stuff", "src", "runtime.rs")).read()
    assert re.search(r"RESTORE", runtime_rs) and re.search(r"cfg_name\(\)", runtime_rs), "runtime.rs no longer restores by cfg_name"
    patched = (mod_rs[:eb] + "    " + "\n    ".join(RECIPE["variants"]) + "\n" + mod_rs[eb:nb] + "    " + "\n    ".join(RECIPE["nodes"]) + "\n" +
               mod_rs[nb:rb] + "    " + "\n    ".join(RECIPE["restore"]) + "\n" + mod_rs[rb:])
    patched = patched.replace("use self::{", RECIPE["use"] + "\nuse self::{", 1)
    assert balanced(patched), "the patched nodes/mod.rs is not balanced"
    assert mod_rs[eb - 1] == "\n" and mod_rs[nb - 1] == "\n" and mod_rs[rb - 1] == "\n", "the lists no longer end on a line of their own"
    report.append("nodes/mod.rs: %d variants / NODES / RESTORE entries; the 2 + 2 + 2 inserted lines fit, collide with nothing, balanced" % len(variants))
    # the shim's own module file declares what the `use` line names
    for modname in ("gpu_bank", "gpu_chain", "engine", "ffi"):
        assert re.search(r"pub mod %s;" % modname, shim["mod.rs"]), "host/rust/src/mod.rs lacks `pub mod %s;`" % modname
    # the README carries exactly these lines
    readme = open(os.path.join(ROOT, "host", "rust", "README.md")).read()
    for l in [RECIPE["main_mod"], RECIPE["use"]] + RECIPE["variants"] + RECIPE["nodes"] + RECIPE["restore"]:
        assert l in readme, "host/rust/README.md does not show the line: %s" % l
    report.append("host/rust/README.md shows the same lines")
    for r in report:
        print("ok  " + r)
    return 0


if __name__ == "__main__":
    sys.exit(main())
