"""host/rust/src/ffi.rs cannot be compiled here (no rustc); this keeps it in step with include/dspfx.h:
same entry points with the same argument counts, same struct fields in the same order, same constants."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = open(os.path.join(ROOT, "include", "dspfx.h")).read()
FFI = open(os.path.join(ROOT, "host", "rust", "src", "ffi.rs")).read()


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def _c_protos():
    body = _strip_comments(HDR)
    out = {}
    for m in re.finditer(r"\b(dspfx_\w+)\s*\(([^;{}]*?)\)\s*;", body):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return out


def _rust_protos():
    body = _strip_comments(FFI)
    out = {}
    for m in re.finditer(r"pub fn (dspfx_\w+)\s*\(([^)]*)\)", body):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if not args else len([a for a in args.split(",") if a.strip()])
    return out


def test_every_entry_point_is_bound_with_the_same_arity(dspfx):
    c, r = _c_protos(), _rust_protos()
    assert set(c) == set(dspfx.EXPORTS)                  # the header parse is sound
    assert set(r) == set(c), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    assert {k: r[k] for k in c} == c


def _c_struct_fields(name):
    m = re.search(r"typedef struct %s\s*\{(.*?)\}\s*%s;" % (name, name), _strip_comments(HDR), re.S)
    return [re.sub(r"\[.*\]", "", d.strip().split()[-1].lstrip("*")) for d in m.group(1).split(";") if d.strip()]


def _rust_struct_fields(name):
    m = re.search(r"pub struct %s\s*\{(.*?)\}" % name, _strip_comments(FFI), re.S)
    return [f.strip().split(":")[0].replace("pub ", "").strip() for f in m.group(1).split(",") if f.strip()]


def test_repr_c_structs_match():
    for name in ("dspfx_engine_desc", "dspfx_node_desc", "dspfx_ctl", "dspfx_graph_link"):
        assert _rust_struct_fields(name) == _c_struct_fields(name), name


def test_constants_match():
    c = {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(DSPFX_[A-Z0-9_]+)\s*=\s*(-?\d+)", _strip_comments(HDR))}
    c.update({m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(DSPFX_[A-Z0-9_]+)\s+\(?(-?\d+)\)?u?\b", HDR)})
    r = {m.group(1): int(m.group(2)) for m in re.finditer(r"pub const (DSPFX_[A-Z0-9_]+): \w+ = (-?\d+);", FFI)}
    assert r, "no constants parsed"
    for k, v in r.items():
        assert k in c, k
        assert c[k] == v, (k, c[k], v)
    for k in c:                                       # every enum value / flag of the header is mirrored
        assert k in r, k


def test_every_c_call_the_rust_sources_make_exists_with_that_arity():
    """host/rust/src/{engine,gpu_chain,gpu_bank}.rs call the C ABI through `unsafe { dspfx_xxx(...) }`: every such call names an
    entry point of include/dspfx.h and passes as many arguments as it takes (a typo or a stale signature would only show up in
    a compiler this image does not have)."""
    c = _c_protos()
    src_dir = os.path.join(ROOT, "host", "rust", "src")
    seen = set()
    for fn in ("engine.rs", "gpu_chain.rs", "gpu_bank.rs"):
        text = re.sub(r'"(?:[^"\\\\]|\\\\.)*"', '""', _strip_comments(open(os.path.join(src_dir, fn)).read()))     # (no string literals)
        for m in re.finditer(r"\b(?:super::ffi::)?(dspfx_[a-z_]+)\s*\(", text):
            name = m.group(1)
            if name in ("dspfx_engine", "dspfx_comm"):
                continue
            # the argument list: up to the matching parenthesis
            depth, k = 1, m.end()
            while depth and k < len(text):
                depth += {"(": 1, ")": -1}.get(text[k], 0)
                k += 1
            args = text[m.end():k - 1].strip()
            n = 0
            if args:
                d = 0
                n = 1
                for ch in args:
                    d += {"(": 1, "[": 1, "{": 1, ")": -1, "]": -1, "}": -1}.get(ch, 0)
                    n += ch == "," and d == 0
                if args.endswith(","):
                    n -= 1
            assert name in c, (fn, name)
            assert n == c[name], (fn, name, n, c[name])
            seen.add(name)
    # GpuBank's documented sequence is really in its source
    bank = open(os.path.join(src_dir, "gpu_bank.rs")).read()
    for call in ("Engine::new", "set_chain", "PinnedBlock::new", "dspfx_link_divisor", "process_host", "engine.params()"):
        assert call in bank, call
    for trait in ("impl Node for GpuBank", "impl NodeStatic for GpuBank", "impl SimpleNode for GpuBank"):
        assert trait in bank, trait
    assert {"dspfx_engine_create", "dspfx_chain_set", "dspfx_process_host", "dspfx_host_alloc", "dspfx_host_free", "dspfx_engine_destroy",
            "dspfx_set_param_seq", "dspfx_link_divisor"} <= seen, seen


def test_rust_sources_are_lexically_well_formed():
    """No rustc here: the least a file must be is balanced -- every bracket closed by its own kind, outside comments, strings
    and character literals.  (It caught nothing so far; it is there for the edit that forgets a brace.)"""
    import glob
    src_dir = os.path.join(ROOT, "host", "rust")
    files = sorted(glob.glob(os.path.join(src_dir, "src", "*.rs")) + glob.glob(os.path.join(src_dir, "*.rs")))
    assert len(files) >= 5
    pairs = {")": "(", "]": "[", "}": "{"}
    for f in files:
        s = open(f).read()
        s = re.sub(r"//[^\n]*", "", s)
        s = re.sub(r"/\*.*?\*/", "", s, flags=re.S)
        s = re.sub(r'"(\\.|[^"\\])*"', '""', s)
        s = re.sub(r"'(\\.|[^'\\])'", "''", s)
        stack, line = [], 1
        for ch in s:
            if ch == "\n":
                line += 1
            elif ch in "([{":
                stack.append((ch, line))
            elif ch in ")]}":
                assert stack and stack[-1][0] == pairs[ch], "%s:%d: unexpected %r" % (f, line, ch)
                stack.pop()
        assert not stack, "%s: unclosed %r" % (f, stack[-1])
