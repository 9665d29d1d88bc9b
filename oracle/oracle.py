"""ctypes front-end of the CPU oracle (oracle/dspfx_oracle.c).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see dspfx_oracle.h).  Imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the
product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

BUF_SIZE = 128

# kinds / modes (numeric values shared with include/dspfx.h)
GAIN, BIQUAD, LOW_PASS, HIGH_PASS, REVERB, DISTORT, OVERDRIVE, CHEBYSHEV, FIR, ADD, MIX, SIGNAL_GEN, ENVELOPE = range(13)
SIG_SINE, SIG_TRIANGLE, SIG_SQUARE, SIG_CONSTANT = range(4)
HARD_CLIP, SOFT_CLIP, TANH, RECIP_SOFT_CLIP, FUZZ, SIN, ATAN, SQUARE, CHEBYSHEV4 = range(9)
FIR_BALANCED, FIR_AVERAGE = 0, 1

LINK_INTERNAL = 1  # hops between chain nodes go through collect_and_average
LINK_INPUT = 2     # ... and so does the hop into the first node


def build(native: bool = False, force: bool = False) -> str:
    """Compile the oracle if needed and return the .so path."""
    name = "liboracle_native.so" if native else "liboracle.so"
    path = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "dspfx_oracle.c")
    hdr = os.path.join(_HERE, "dspfx_oracle.h")
    stale = (not os.path.exists(path)) or any(
        os.path.getmtime(s) > os.path.getmtime(path) for s in (src, hdr))
    if stale or force:
        subprocess.check_call(["make", "-C", _HERE, name], stdout=subprocess.DEVNULL)
    return path


_lib = None


def lib(native: bool = False):
    global _lib
    if native:
        return _bind(C.CDLL(build(native=True)))
    if _lib is None:
        _lib = _bind(C.CDLL(build()))
    return _lib


def _bind(L):
    vp, f32p, f64p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double)
    L.orc_node_new.restype = vp
    L.orc_node_new.argtypes = [C.c_int]
    L.orc_node_free.argtypes = [vp]
    L.orc_node_clone.restype = vp
    L.orc_node_clone.argtypes = [vp]
    L.orc_node_set_param.argtypes = [vp, C.c_int, C.c_float]
    L.orc_node_init_param.argtypes = [vp, C.c_int, C.c_float]
    L.orc_node_after_settings_change.argtypes = [vp]
    L.orc_node_set_mode.argtypes = [vp, C.c_int]
    L.orc_reverb_set_len.argtypes = [vp, C.c_uint32]
    L.orc_reverb_make_buffer.argtypes = [vp]
    L.orc_delay_len.restype = C.c_uint32
    L.orc_delay_len.argtypes = [C.c_float, C.c_int]
    L.orc_fir_set_taps.argtypes = [vp, f64p, C.c_uint32]
    L.orc_node_reset.argtypes = [vp]
    L.orc_link_divisor.restype = C.c_float
    L.orc_link_divisor.argtypes = [C.c_uint64]
    L.orc_collect_and_average.restype = C.c_int
    L.orc_collect_and_average.argtypes = [f32p, C.POINTER(f32p), C.c_int, C.c_size_t]
    L.orc_slider_input.argtypes = [f32p, f32p, C.c_float, C.c_float, f32p, C.c_size_t]
    L.orc_node_process.argtypes = [vp, f32p, f32p, C.POINTER(f32p), f32p, C.c_size_t]
    L.orc_chain_run.argtypes = [C.POINTER(vp), C.c_int, C.c_int, f32p, f32p, f32p, C.c_size_t, C.c_size_t]
    L.orc_chain_run_ctl.argtypes = [C.POINTER(vp), C.c_int, C.c_int, f32p, f32p, C.POINTER(f32p), f32p, C.c_size_t, C.c_size_t]
    L.orc_noise.restype = C.c_float
    L.orc_noise.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    if hasattr(L, "orc_bench_spin"):
        L.orc_bench_spin.restype = C.c_int
        L.orc_bench_spin.argtypes = [C.c_int, C.c_uint64, C.POINTER(C.c_double)]
    if hasattr(L, "orc_bench_chain"):
        L.orc_bench_chain.restype = C.c_int
        L.orc_bench_chain.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                      C.POINTER(C.c_double)]
    L.orc_run_noise_channels.restype = C.c_int
    L.orc_run_noise_channels.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, f32p, f64p, C.c_int]
    return L


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


class Node:
    """One reference node instance for one mono channel."""

    def __init__(self, kind: int, params=None, mode=None, delay_len=None, taps_reversed=None, _lib=None, restored=False):
        """NodeStatic::new + the field values: the fields are stored WITHOUT the node's after_settings_change hook (the
        generated new() / the setters of restore(), dsp-stuff-derive/src/lib.rs:196-210, 300-312).  A BiQuad's filter is
        then built from its sliders (what the engine's dspfx_chain_set does with the six params; restore() does the same
        through the hook).  A Reverb keeps make_buffer()'s 128-sample ring unless `delay_len` names the ring (the explicit
        D of this restatement) or `restored` runs the hook as restore() would (lib.rs:319-337: D from the seconds slider)."""
        self.L = _lib or lib()
        self.kind = kind
        self.h = C.c_void_p(self.L.orc_node_new(kind))
        if mode is not None:
            self.L.orc_node_set_mode(self.h, int(mode))
            if kind == REVERB:
                self.L.orc_reverb_make_buffer(self.h)      # make_buffer() under the mode's reading of rivulet: 128 / 1024 samples
        if params:
            for i, v in enumerate(params):
                if v is not None:
                    self.L.orc_node_init_param(self.h, i, float(v))
        if kind == REVERB and delay_len is not None and not (params and len(params) > 1 and params[1] is not None):
            self.L.orc_node_init_param(self.h, 1, 0.0)     # an explicit ring and no seconds slider given: nothing to refresh from
        if kind == BIQUAD or restored:
            self.L.orc_node_after_settings_change(self.h)
        if delay_len is not None:
            self.L.orc_reverb_set_len(self.h, int(delay_len))
        if taps_reversed is not None:
            t = np.ascontiguousarray(taps_reversed, dtype=np.float64)
            self.L.orc_fir_set_taps(self.h, t.ctypes.data_as(C.POINTER(C.c_double)), len(t))

    def set_param(self, idx, v):
        """A slider change in the GUI: the store AND the node's after_settings_change hook (biquad: new filter, state
        reset; reverb -- any slider, decay included -- a new zero ring: dspfx_oracle.h has the table)."""
        self.L.orc_node_set_param(self.h, idx, float(v))

    def set_delay_len(self, d):
        """reverb.rs:55-71 with D explicit: a new zero ring of d samples."""
        self.L.orc_reverb_set_len(self.h, int(d))

    def set_taps(self, taps_reversed):
        """Impulse-response reload (fir.rs:153-171): new taps, the deque of past samples is KEPT."""
        t = np.ascontiguousarray(taps_reversed, dtype=np.float64)
        self.L.orc_fir_set_taps(self.h, t.ctypes.data_as(C.POINTER(C.c_double)), len(t))

    def reset(self):
        self.L.orc_node_reset(self.h)

    def process(self, in_a, in_b=None, ctl=None):
        """SimpleNode::process on one block (<=128 frames)."""
        a = np.ascontiguousarray(in_a, dtype=np.float32)
        assert a.size <= BUF_SIZE
        b = np.ascontiguousarray(in_b, dtype=np.float32) if in_b is not None else None
        out = np.zeros_like(a)
        ctl_arr = None
        keep = []
        if ctl is not None:
            arr = (C.POINTER(C.c_float) * 3)()
            for i in range(3):
                c = ctl[i] if i < len(ctl) else None
                if c is not None:
                    c = np.ascontiguousarray(c, dtype=np.float32)
                    keep.append(c)
                    arr[i] = _f32p(c)
            ctl_arr = arr
        self.L.orc_node_process(self.h, _f32p(a), _f32p(b), ctl_arr, _f32p(out), a.size)
        return out

    def __del__(self):
        try:
            self.L.orc_node_free(self.h)
        except Exception:
            pass


def node_from_desc(d: dict, _lib=None) -> Node:
    """d = {"kind":..., "params":[...], "mode":..., "delay_len":..., "taps_reversed":...}"""
    return Node(d["kind"], d.get("params"), d.get("mode"), d.get("delay_len"),
                d.get("taps_reversed"), _lib=_lib)


def chain_run(nodes, x, link_flags=LINK_INTERNAL | LINK_INPUT, side=None, block=BUF_SIZE, ctl=None):
    """Run one channel through the chain (state is carried in `nodes`).
    ctl: {(node_index, slider_index): signal} for connected `as_input` control ports."""
    L = nodes[0].L if nodes else lib()
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    hs = (C.c_void_p * len(nodes))(*[n.h for n in nodes])
    s = np.ascontiguousarray(side, dtype=np.float32) if side is not None else None
    if ctl:
        arr = (C.POINTER(C.c_float) * (3 * len(nodes)))()
        keep = []
        for (k, j), sig in ctl.items():
            sig = np.ascontiguousarray(sig, dtype=np.float32)
            assert sig.size == x.size
            keep.append(sig)
            arr[3 * k + j] = _f32p(sig)
        L.orc_chain_run_ctl(hs, len(nodes), int(link_flags), _f32p(x), _f32p(s), arr, _f32p(out), x.size, block)
    else:
        L.orc_chain_run(hs, len(nodes), int(link_flags), _f32p(x), _f32p(s), _f32p(out), x.size, block)
    return out


def run_channels(descs, x, link_flags=LINK_INTERNAL | LINK_INPUT, side=None, block=BUF_SIZE, ctl=None, nodes_out=None):
    """x: [n_frames][n_channels] frame-major; fresh state per channel (or the per-channel node lists
    in `nodes_out`, which are created on first use and carry state across calls).
    ctl: {(node, slider): [n_frames][n_channels]}."""
    x = np.asarray(x, dtype=np.float32)
    out = np.empty_like(x)
    for c in range(x.shape[1]):
        if nodes_out is not None:
            if len(nodes_out) <= c:
                nodes_out.append([node_from_desc(d) for d in descs])
            nodes = nodes_out[c]
        else:
            nodes = [node_from_desc(d) for d in descs]
        s = side[:, c] if side is not None else None
        cc = {k: v[:, c] for k, v in ctl.items()} if ctl else None
        out[:, c] = chain_run(nodes, x[:, c], link_flags, s, block, cc)
    return out


def noise(seed, channels, n_abs):
    """Hashed white noise block [len(n_abs)][len(channels)] (vectorised restatement of orc_noise)."""
    c = np.asarray(channels, dtype=np.uint32)[None, :]
    n = np.asarray(n_abs, dtype=np.uint32)[:, None]
    with np.errstate(over="ignore"):
        h = np.uint32(seed) ^ (c * np.uint32(0x9E3779B9)) ^ (n * np.uint32(0x85EBCA6B))
        h ^= h >> np.uint32(16)
        h *= np.uint32(0x85EBCA6B)
        h ^= h >> np.uint32(13)
        h *= np.uint32(0xC2B2AE35)
        h ^= h >> np.uint32(16)
    return ((h >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -23) - np.float32(1.0)).astype(np.float32)


def run_noise_channels(descs, seed, c0, n_channels, n_abs0, n_blocks, link_flags=3, block=BUF_SIZE,
                       want_out=True, want_mix=False, n_threads=1, native=False):
    L = lib(native=native) if native else lib()
    protos = [node_from_desc(d, _lib=L) for d in descs]
    hs = (C.c_void_p * len(protos))(*[n.h for n in protos])
    nf = n_blocks * block
    out = np.empty((nf, n_channels), dtype=np.float32) if want_out else None
    mix = np.zeros(nf, dtype=np.float64) if want_mix else None
    L.orc_run_noise_channels(hs, len(protos), int(link_flags), seed, c0, n_channels, n_abs0, n_blocks,
                             block, _f32p(out), mix.ctypes.data_as(C.POINTER(C.c_double)) if want_mix else None,
                             n_threads)
    return out, mix


def link_divisor(n):
    return np.float32(lib().orc_link_divisor(int(n)))


def delay_len(seconds, page_round=False):
    return int(lib().orc_delay_len(float(seconds), int(page_round)))
