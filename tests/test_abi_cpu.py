"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every
symbol include/dspfx.h declares, and fails loudly (no CPU fallback) without a GPU.
No compute entry point is called here."""
import os
import re

import numpy as np
import pytest

import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_all_exported(dspfx):
    hdr = open(os.path.join(ROOT, "include", "dspfx.h")).read()
    declared = set(re.findall(r"\b(dspfx_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"dspfx_kind", "dspfx_status"}
    assert declared, "no declarations parsed"
    L = dspfx.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(dspfx.EXPORTS), declared ^ set(dspfx.EXPORTS)
    assert L.dspfx_abi_version() == dspfx.ABI_VERSION


def test_header_constants_match_reference_and_oracle(dspfx):
    hdr = open(os.path.join(ROOT, "include", "dspfx.h")).read()
    assert "#define DSPFX_BUF_SIZE 128" in hdr          # node.rs:257
    assert dspfx.BUF_SIZE == O.BUF_SIZE == 128
    kinds = ["GAIN", "BIQUAD", "LOW_PASS", "HIGH_PASS", "REVERB", "DISTORT", "OVERDRIVE", "CHEBYSHEV", "FIR", "ADD", "MIX"]
    for i, k in enumerate(kinds):
        assert getattr(dspfx, k) == getattr(O, k) == i
        assert re.search(rf"DSPFX_{k} = {i},", hdr), k
    # distort.rs:18-28 declaration order
    assert dspfx.DISTORT_MODES == ["HardClip", "SoftClip", "Tanh", "RecipSoftClip", "Fuzz", "Sin", "Atan", "Square", "Chebyshev4"]


def test_host_helpers_match_oracle(dspfx):
    for s in (0.0, 0.001, 0.0026, 0.0027, 0.25, 0.5, 0.75, 1.0):
        for pr in (False, True):
            assert dspfx.delay_len(s, pr) == O.delay_len(s, pr)
    for n in (0, 1, 2, 3, 100, 2047, 2048, 65536, 1 << 20):
        assert dspfx.link_divisor(n) == O.link_divisor(n), n
    # saturation shortcut in the library must equal the literal loop for huge N
    assert dspfx.link_divisor(1 << 25) == np.float32(1 << 24)


def test_node_defaults_match_reference(dspfx):
    import ctypes as C
    L = dspfx.lib()
    from dsp_stuff_amd import _NodeDesc
    d = _NodeDesc()
    assert L.dspfx_node_defaults(dspfx.BIQUAD, C.byref(d)) == 0
    assert [round(float(x), 6) for x in d.params[:6]] == [1.0, -0.24, 0.0, 0.758, 0.0, 0.0]   # biquad.rs:18-41
    assert L.dspfx_node_defaults(dspfx.REVERB, C.byref(d)) == 0
    assert d.delay_len == 128 and d.params[0] == 0.5                                             # reverb.rs:37,44-52
    assert L.dspfx_node_defaults(dspfx.DISTORT, C.byref(d)) == 0
    assert d.mode == dspfx.SOFT_CLIP and d.params[0] == 0.0                                      # distort.rs:46-50
    assert L.dspfx_node_defaults(dspfx.GAIN, C.byref(d)) == 0 and d.params[0] == 1.0
    assert L.dspfx_node_defaults(99, C.byref(d)) == -1


def test_no_cpu_fallback(dspfx):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert dspfx.device_count() == 0
    with pytest.raises(dspfx.DspfxError) as ei:
        dspfx.Engine(64)
    assert ei.value.status == -2   # DSPFX_ERR_NO_DEVICE


def test_product_never_imports_oracle():
    """The product path must not route through oracle/ (it is the checker only)."""
    pkg = os.path.join(ROOT, "dsp-stuff_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", ".c")) or f == "Makefile":
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert "liboracle" not in src and "numpy_model" not in src, f
                assert not re.search(r"#\s*include[^\n]*oracle", src), f
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, re.M), f
                assert "load_oracle" not in src, f


def test_no_getenv_on_the_per_block_path():
    """every DSPFX_* switch is read at a setup call (library load, dspfx_engine_create, dspfx_chain_set /
    dspfx_graph_set -> EnvSwitches / FirEnv, dspfx_comm_create), never by a process call, a slider store, dspfx_mix_allreduce
    or the FIR launch path -- a host may be changing its environment at that moment.  Structural check: in the translation
    units of libdspfx.so, `getenv(` may appear only inside the functions named here."""
    allowed = {
        "dspfx.hip": set(),
        "host_pipe.hip": {"dspfx_process_host"},        # two `static const` locals: read once per process
        "plan.hip": {"env_int", "read_env_switches"},
        "jit.hip": {"jit_headers_dir", "jit_cache_dir", "<file scope>"},
        "comm.hip": {"rccl", "want_rccl", "comm_timeout_ms", "mailbox_join"},
        "fir_kernels.hip": {"fir_configure"},
        "placement.hip": {"big_alloc", "tune_ring", "dspfx_tune_placement"},      # setup-time calls
        "state_util.hip": {"dspfx_describe"},
        "aux_kernels.hip": set(),
    }
    csrc = os.path.join(ROOT, "dsp-stuff_amd", "csrc")
    head = re.compile(r"^(?:extern \"C\" |static |inline )*[A-Za-z_][\w:<>\*&\s]*?\b([A-Za-z_]\w*)\s*\([^;]*$")
    for name, ok in allowed.items():
        func = "<file scope>"
        depth = 0
        for ln, line in enumerate(open(os.path.join(csrc, name), errors="ignore"), 1):
            code = line.split("//")[0]
            if re.match(r"^(namespace\b.*\{|\}\s*//\s*namespace)", line):      # namespaces do not nest functions
                continue
            if depth == 0 and not line.startswith((" ", "\t", "}", "#", "/")):
                m = head.match(code.rstrip())
                if m and "(" in code:
                    func = m.group(1)
            if "getenv(" in code:
                where = func if depth > 0 or "{" in code else "<file scope>"
                if depth == 0 and "=" in code and not code.rstrip().endswith("{"):
                    where = "<file scope>"
                assert where in ok, f"{name}:{ln}: getenv inside {where}() -- read it at setup into EnvSwitches / FirEnv"
            depth += code.count("{") - code.count("}")
            if depth == 0 and "}" in code:
                func = "<file scope>"
    # ... and the two functions every block goes through mention the snapshot, not the environment
    src = open(os.path.join(csrc, "dspfx.hip")).read()
    assert "e->env.mix_tail" in src and "e->env.xcd_remap" in src and "read_env_switches()" in src
