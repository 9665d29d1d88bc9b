"""dsp-stuff_amd -- Python binding of the MI355X-native effect-chain engine.

A thin ctypes layer over the C ABI in include/dspfx.h (csrc/libdspfx.so, hand-written
HIP for gfx950).  Node constructors carry the reference's names, slider fields,
ranges and defaults (dsp-stuff/src/nodes/*.rs) so tests read like the reference's
node definitions.  There is no CPU fallback: importing works anywhere (so the
symbol table can be checked), but creating an Engine without a HIP device raises.

The directory name has a hyphen (it is the repo's package directory, not an
importable identifier): load it with `__graft_entry__.load_package()`.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from dataclasses import dataclass, field
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DSPFX_LIB lets tools/ab.py load an experiment build of the same library side by side
LIB_PATH = os.environ.get("DSPFX_LIB") or os.path.join(_HERE, "csrc", "libdspfx.so")

BUF_SIZE = 128  # dsp-stuff/src/node.rs:257
ABI_VERSION = 2

# dspfx_kind
GAIN, BIQUAD, LOW_PASS, HIGH_PASS, REVERB, DISTORT, OVERDRIVE, CHEBYSHEV, FIR, ADD, MIX, SIGNAL_GEN, ENVELOPE = range(13)
SIG_SINE, SIG_TRIANGLE, SIG_SQUARE, SIG_CONSTANT = range(4)
# dspfx_distort_mode (nodes/distort.rs:18-28)
HARD_CLIP, SOFT_CLIP, TANH, RECIP_SOFT_CLIP, FUZZ, SIN, ATAN, SQUARE, CHEBYSHEV4 = range(9)
DISTORT_MODES = ["HardClip", "SoftClip", "Tanh", "RecipSoftClip", "Fuzz", "Sin", "Atan", "Square", "Chebyshev4"]
FIR_BALANCED, FIR_AVERAGE = 0, 1
FIR_PRECISION_DEFAULT, FIR_PRECISION_F32, FIR_PRECISION_SPLIT, FIR_PRECISION_HALF = 0, 1, 2, 3
LINK_INTERNAL, LINK_INPUT, LINK_SIDE_RAW = 1, 2, 4
MAX_LINKS = 16
GRAPH_MAX_NODES = 16
ERR_UNSUPPORTED = -5
GRAPH_INPUT, GRAPH_ZERO, GRAPH_INPUT2 = -1, -2, -3
GRAPH_MAX_IO = 16
GRAPH_INPUTS = (GRAPH_INPUT, GRAPH_INPUT2) + tuple(-(2 + k) for k in range(2, GRAPH_MAX_IO))     # link source of input block k
PORT_MAIN, PORT_SIDE, PORT_SLIDER = 0, 1, 2
PORT_RAW = 256

# every symbol include/dspfx.h declares
EXPORTS = [
    "dspfx_abi_version", "dspfx_strerror", "dspfx_device_count", "dspfx_node_defaults", "dspfx_delay_len",
    "dspfx_link_divisor", "dspfx_engine_create", "dspfx_engine_destroy", "dspfx_last_error", "dspfx_chain_set",
    "dspfx_chain_len", "dspfx_set_param", "dspfx_set_mode", "dspfx_set_delay_len", "dspfx_set_taps", "dspfx_set_fir_precision",
    "dspfx_reset", "dspfx_tune_placement", "dspfx_process", "dspfx_process_host", "dspfx_host_alloc", "dspfx_host_free", "dspfx_mix_finish", "dspfx_process_mixpipe", "dspfx_mixpipe_flush", "dspfx_link_average", "dspfx_graph_set", "dspfx_graph_source", "dspfx_state_size",
    "dspfx_state_export", "dspfx_state_import", "dspfx_fill_noise", "dspfx_sync", "dspfx_describe",
    "dspfx_algorithmic_bytes_per_sample", "dspfx_profile_enable", "dspfx_profile_read", "dspfx_verify_fast_division", "dspfx_verify_libm",
    "dspfx_process_partials", "dspfx_mix_collect", "dspfx_process_ctl",
    "dspfx_process_io", "dspfx_comm_unique_id", "dspfx_comm_create", "dspfx_comm_destroy", "dspfx_comm_size", "dspfx_comm_rank",
    "dspfx_comm_last_error", "dspfx_mix_allreduce",
    "dspfx_set_param_seq", "dspfx_param_log", "dspfx_frames_submitted", "dspfx_process_bus", "dspfx_kernels_ready", "dspfx_comm_backend",
    "dspfx_reserve_delay_len", "dspfx_ring_trim",
]
COMM_ID_BYTES = 128


class DspfxError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"dspfx error {status}: {msg}")
        self.status = status


class _EngineDesc(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("device", C.c_int32), ("channels", C.c_uint32),
                ("max_frames", C.c_uint32), ("link_flags", C.c_uint32), ("tile_channels", C.c_uint32),
                ("channel_offset", C.c_uint64)]


class _NodeDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("mode", C.c_int32), ("params", C.c_float * 8),
                ("delay_len", C.c_uint32), ("n_taps", C.c_uint32), ("taps", C.POINTER(C.c_double))]


class _GraphLink(C.Structure):
    _fields_ = [("src", C.c_int32), ("dst", C.c_int32), ("port", C.c_int32)]


class _ParamEvent(C.Structure):
    _fields_ = [("seq", C.c_uint64), ("frame", C.c_uint64), ("node", C.c_int32), ("param", C.c_int32),
                ("value", C.c_float), ("reserved", C.c_int32)]


class _Ctl(C.Structure):
    _fields_ = [("node", C.c_int32), ("param", C.c_int32), ("signal", C.c_void_p)]


_lib = None


def lib():
    """Load csrc/libdspfx.so; fails loudly when the HIP library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C {os.path.dirname(LIB_PATH)}` "
                          "(or __graft_entry__.build()); there is no fallback implementation")
    # One HIP runtime per process: PyTorch ships its own copy of the ROCm libraries, and a process that maps the system's
    # libamdhip64 (through libdspfx.so) BEFORE importing torch ends up with two runtimes, of which only the first to
    # initialise sees the GPU (measured: tools/probe_import_order.py -- either torch or this library reports no device).
    # Importing torch first makes libdspfx.so bind to the copy torch loaded.  Hosts without torch just use the system's.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(LIB_PATH)
    vp, f32p = C.c_void_p, C.c_void_p
    L.dspfx_abi_version.restype = C.c_uint32
    L.dspfx_strerror.restype = C.c_char_p
    L.dspfx_strerror.argtypes = [C.c_int]
    L.dspfx_device_count.restype = C.c_int
    L.dspfx_node_defaults.argtypes = [C.c_int, C.POINTER(_NodeDesc)]
    L.dspfx_delay_len.restype = C.c_uint32
    L.dspfx_delay_len.argtypes = [C.c_float, C.c_int]
    L.dspfx_link_divisor.restype = C.c_float
    L.dspfx_link_divisor.argtypes = [C.c_uint64]
    L.dspfx_engine_create.argtypes = [C.POINTER(_EngineDesc), C.POINTER(vp)]
    L.dspfx_engine_destroy.argtypes = [vp]
    L.dspfx_engine_destroy.restype = None
    L.dspfx_last_error.restype = C.c_char_p
    L.dspfx_last_error.argtypes = [vp]
    L.dspfx_chain_set.argtypes = [vp, C.POINTER(_NodeDesc), C.c_int]
    L.dspfx_chain_len.argtypes = [vp]
    L.dspfx_kernels_ready.argtypes = [vp, C.c_int]
    L.dspfx_graph_set.argtypes = [vp, C.POINTER(_NodeDesc), C.c_int, C.POINTER(_GraphLink), C.c_int]
    L.dspfx_graph_source.argtypes = [C.POINTER(_NodeDesc), C.c_int, C.POINTER(_GraphLink), C.c_int, C.c_char_p, C.c_size_t]
    L.dspfx_set_param.argtypes = [vp, C.c_int, C.c_int, C.c_float]
    L.dspfx_set_mode.argtypes = [vp, C.c_int, C.c_int]
    L.dspfx_set_param_seq.argtypes = [vp, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_uint64)]
    L.dspfx_param_log.argtypes = [vp, C.POINTER(_ParamEvent), C.c_int, C.c_uint64]
    L.dspfx_frames_submitted.restype = C.c_uint64
    L.dspfx_frames_submitted.argtypes = [vp]
    L.dspfx_set_delay_len.argtypes = [vp, C.c_int, C.c_uint32]
    L.dspfx_reserve_delay_len.argtypes = [vp, C.c_int, C.c_uint32]
    L.dspfx_ring_trim.argtypes = [vp]
    L.dspfx_set_taps.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.c_uint32, C.c_int]
    L.dspfx_set_fir_precision.argtypes = [vp, C.c_int, C.c_int]
    L.dspfx_reset.argtypes = [vp]
    L.dspfx_process.argtypes = [vp, f32p, f32p, f32p, f32p, C.c_uint32, vp]
    L.dspfx_process_host.argtypes = [vp, f32p, f32p, f32p, f32p, C.c_uint32]
    L.dspfx_process_bus.argtypes = [vp, f32p, f32p, f32p, f32p, C.c_uint32, C.c_uint64, vp]
    L.dspfx_mix_finish.argtypes = [vp, f32p, C.c_uint32, C.c_uint64, vp]
    L.dspfx_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    L.dspfx_host_free.argtypes = [C.c_void_p]
    L.dspfx_tune_placement.argtypes = [vp, f32p, f32p, f32p, C.c_uint32, vp]
    L.dspfx_process_mixpipe.argtypes = [vp, f32p, f32p, f32p, f32p, C.c_uint32, C.c_uint64, vp]
    L.dspfx_mixpipe_flush.argtypes = [vp, f32p, f32p, C.c_uint64, vp]
    L.dspfx_link_average.argtypes = [vp, C.POINTER(C.c_void_p), C.c_int, f32p, C.c_uint32, vp]
    L.dspfx_process_ctl.argtypes = [vp, f32p, f32p, f32p, f32p, C.c_uint32, C.POINTER(_Ctl), C.c_int, vp]
    L.dspfx_process_partials.argtypes = [vp, f32p, f32p, f32p, C.c_uint32, vp]
    L.dspfx_mix_collect.argtypes = [vp, f32p, C.c_uint32, vp]
    L.dspfx_state_size.restype = C.c_int64
    L.dspfx_state_size.argtypes = [vp, C.c_int]
    L.dspfx_state_export.argtypes = [vp, C.c_int, vp, C.c_size_t]
    L.dspfx_state_import.argtypes = [vp, C.c_int, vp, C.c_size_t]
    L.dspfx_fill_noise.argtypes = [vp, f32p, C.c_uint32, C.c_uint32, C.c_uint32, vp]
    L.dspfx_sync.argtypes = [vp, vp]
    L.dspfx_describe.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.dspfx_verify_fast_division.argtypes = [C.c_int, C.c_float, C.POINTER(C.c_uint64)]
    L.dspfx_verify_libm.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    L.dspfx_profile_enable.argtypes = [vp, C.c_int]
    L.dspfx_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.c_char_p, C.c_size_t, C.c_int]
    L.dspfx_process_io.argtypes = [vp, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.c_int, f32p, C.c_uint32, vp]
    L.dspfx_comm_unique_id.argtypes = [vp]
    L.dspfx_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.dspfx_comm_destroy.argtypes = [vp]
    L.dspfx_comm_destroy.restype = None
    L.dspfx_comm_size.argtypes = [vp]
    L.dspfx_comm_rank.argtypes = [vp]
    L.dspfx_comm_last_error.argtypes = [vp]
    L.dspfx_comm_last_error.restype = C.c_char_p
    L.dspfx_comm_backend.argtypes = [vp]
    L.dspfx_comm_backend.restype = C.c_char_p
    L.dspfx_mix_allreduce.argtypes = [vp, vp, f32p, C.c_uint32, C.c_uint64, vp]
    L.dspfx_algorithmic_bytes_per_sample.restype = C.c_double
    L.dspfx_algorithmic_bytes_per_sample.argtypes = [vp, C.c_uint32]
    _lib = L
    return L


def device_count() -> int:
    return int(lib().dspfx_device_count())


def delay_len(seconds: float, page_round: bool = False) -> int:
    """reverb.rs:58 (and the page-rounded reading of rivulet's ring capacity)."""
    return int(lib().dspfx_delay_len(float(seconds), int(page_round)))


def verify_fast_division(c: float, device: int = 0) -> int:
    """Exhaustive (2^32 inputs) check of the fast constant division for divisor c: mismatch count."""
    n = C.c_uint64()
    rc = lib().dspfx_verify_fast_division(device, float(c), C.byref(n))
    if rc != 0:
        raise DspfxError(rc, lib().dspfx_strerror(rc).decode())
    return int(n.value)


class PinnedArray:
    """A float32 numpy array over page-locked host memory from dspfx_host_alloc (`.array`); freed on close()/GC."""

    def __init__(self, shape):
        n = int(np.prod(shape))
        self._p = C.c_void_p()
        rc = lib().dspfx_host_alloc(n * 4, C.byref(self._p))
        if rc != 0:
            raise DspfxError(rc, lib().dspfx_strerror(rc).decode())
        self.array = np.ctypeslib.as_array((C.c_float * n).from_address(self._p.value)).reshape(shape)

    def close(self):
        if getattr(self, "_p", None) is not None and self._p.value:
            self.array = None
            lib().dspfx_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def verify_libm(func: int, device: int = 0):
    """Exhaustive comparison of the engine's f64 tanh (0) / sin (1) / atan (2) with the math library's: (differing inputs, max ulp)."""
    n, u = C.c_uint64(), C.c_uint32()
    rc = lib().dspfx_verify_libm(device, int(func), C.byref(n), C.byref(u))
    if rc != 0:
        raise DspfxError(rc, lib().dspfx_strerror(rc).decode())
    return int(n.value), int(u.value)


def link_divisor(n_connected: int) -> np.float32:
    return np.float32(lib().dspfx_link_divisor(int(n_connected)))


def to_layout(x: np.ndarray, tile_channels: int) -> np.ndarray:
    """[n_frames][N] frame-major -> the engine layout for `tile_channels` (flat view)."""
    if not tile_channels:
        return np.ascontiguousarray(x)
    nf, n = x.shape
    w = tile_channels
    return np.ascontiguousarray(x.reshape(nf, n // w, w).transpose(1, 0, 2))


def from_layout(t: np.ndarray, n_frames: int, channels: int, tile_channels: int) -> np.ndarray:
    """inverse of to_layout: -> [n_frames][N]"""
    if not tile_channels:
        return np.asarray(t).reshape(n_frames, channels)
    w = tile_channels
    return np.ascontiguousarray(np.asarray(t).reshape(channels // w, n_frames, w).transpose(1, 0, 2)).reshape(n_frames, channels)


# --------------------------------------------------------------------------- nodes

@dataclass
class NodeSpec:
    kind: int
    params: Sequence[float] = field(default_factory=list)
    mode: int = 0
    delay_len: int = 0
    taps_reversed: Optional[np.ndarray] = None   # as fir.rs:163,168 stores them

    def oracle_desc(self) -> dict:
        """Same node for the CPU oracle (tests only)."""
        return {"kind": self.kind, "params": list(self.params), "mode": self.mode,
                "delay_len": self.delay_len or None, "taps_reversed": self.taps_reversed}


def Gain(level: float = 1.0) -> NodeSpec:
    """nodes/gain.rs: slider level 0..=10, default 1.0"""
    return NodeSpec(GAIN, [level])


def BiQuad(a0=1.0, a1=-0.24, a2=0.0, b0=0.758, b1=0.0, b2=0.0) -> NodeSpec:
    """nodes/biquad.rs:18-41: raw sliders, each -10..=10; normalised by a0 on the engine."""
    return NodeSpec(BIQUAD, [a0, a1, a2, b0, b1, b2])


def LowPass(ratio: float = 0.5) -> NodeSpec:
    """nodes/low_pass.rs: slider ratio 0..=1, default 0.5"""
    return NodeSpec(LOW_PASS, [ratio])


def HighPass(ratio: float = 0.5) -> NodeSpec:
    """nodes/high_pass.rs"""
    return NodeSpec(HIGH_PASS, [ratio])


def Reverb(seconds: Optional[float] = None, decay: float = 0.5, delay_samples: Optional[int] = None,
           page_round: bool = False) -> NodeSpec:
    """nodes/reverb.rs: feedback delay; params = [decay, seconds], mode bit 0 = page_round.
      Reverb(seconds=s)         a restored node: refresh_seconds has run, the ring is reverb.rs:58's length for s;
      Reverb()                  a node fresh from the menu: make_buffer()'s 128-sample ring (reverb.rs:44-52; 1024 under
                                page_round -- the same rivulet calls as refresh_seconds, so the same reading) under the default
                                0.5 s slider -- its first slider change makes it a 24000-sample delay, like the reference's;
      Reverb(delay_samples=D)   an explicit ring and no seconds slider: a slider change swaps in a zero ring of the same D.
    Any set_param on the node -- decay included -- swaps in a NEW ZERO ring (reverb.rs:19, 55-71; include/dspfx.h)."""
    if delay_samples is None:
        if seconds is None:
            return NodeSpec(REVERB, [decay, 0.5], mode=int(page_round), delay_len=delay_len(0.0, page_round))
        delay_samples = delay_len(seconds, page_round)
    return NodeSpec(REVERB, [decay, 0.0 if seconds is None else float(seconds)], mode=int(page_round), delay_len=int(delay_samples))


def Distort(level: float = 0.0, mode: int = SOFT_CLIP) -> NodeSpec:
    """nodes/distort.rs: slider level 0..=30 default 0.0 (=> bypass), mode default SoftClip"""
    return NodeSpec(DISTORT, [level], mode=mode)


def Overdrive(boost: float = 0.0, drive: float = 0.0, level: float = 0.0) -> NodeSpec:
    """nodes/overdrive.rs:21-28 (field order boost, drive, level)"""
    return NodeSpec(OVERDRIVE, [boost, drive, level])


def Chebyshev(level_pos: float = 0.0, level_neg: float = 0.0) -> NodeSpec:
    """nodes/chebyshev.rs:21-25"""
    return NodeSpec(CHEBYSHEV, [level_pos, level_neg])


def Fir(impulse_response=(1.0,), mode: int = FIR_BALANCED) -> NodeSpec:
    """nodes/fir.rs: `impulse_response` is h[0..T) in natural order (what the WAV holds);
    it is stored time-reversed exactly like fir.rs:163,168."""
    h = np.ascontiguousarray(impulse_response, dtype=np.float64)
    return NodeSpec(FIR, [], mode=mode, taps_reversed=np.ascontiguousarray(h[::-1]))


def Add() -> NodeSpec:
    """nodes/add.rs: out = a + b (b = the engine's side input)"""
    return NodeSpec(ADD)


def Mix(ratio: float = 0.5) -> NodeSpec:
    """nodes/mix.rs: out = b*ratio + a*(1-ratio)"""
    return NodeSpec(MIX, [ratio])


def SignalGen(amplitude: float = 0.5, frequency: float = 100.0, mode: int = SIG_SINE) -> NodeSpec:
    """nodes/signal_gen.rs:41-55: a source -- it has no "in" port, so as a chain node it replaces the
    signal (put it first).  Sliders amplitude -1..=1 and frequency 0.1..=20000 Hz are both `as_input`."""
    return NodeSpec(SIGNAL_GEN, [amplitude, frequency], mode=mode)


def Envelope(attack: float = 0.0, release: float = 0.0) -> NodeSpec:
    """nodes/envelope.rs:27-30: peak envelope follower, attack / release in frames (sliders 0..=1000, default 0)."""
    return NodeSpec(ENVELOPE, [attack, release])


# -------------------------------------------------------------------------- engine

def _ptr(x):
    if x is None:
        return None
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):        # torch tensor on the device
        return C.c_void_p(x.data_ptr())
    raise TypeError(f"expected a device tensor or raw pointer, got {type(x)}")


class Engine:
    """N independent mono channels through one effect chain (include/dspfx.h)."""

    def __init__(self, channels: int, max_frames: int = BUF_SIZE, link_flags: int = LINK_INTERNAL | LINK_INPUT,
                 device: int = 0, channel_offset: int = 0, tile_channels: int = 0):
        self.L = lib()
        self.channels, self.max_frames = int(channels), int(max_frames)
        self.h = C.c_void_p()
        d = _EngineDesc(ABI_VERSION, device, channels, max_frames, link_flags, tile_channels, channel_offset)
        self.tile_channels = int(tile_channels)
        rc = self.L.dspfx_engine_create(C.byref(d), C.byref(self.h))
        if rc != 0:
            self.h = C.c_void_p()
            raise DspfxError(rc, self.L.dspfx_strerror(rc).decode())
        self._keep = []

    def _chk(self, rc):
        if rc != 0:
            raise DspfxError(rc, self.L.dspfx_last_error(self.h).decode() or self.L.dspfx_strerror(rc).decode())

    def close(self):
        h = getattr(self, "h", None)
        if h is not None and h.value:
            self.L.dspfx_engine_destroy(h)
            h.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown: module globals may already be gone
            pass

    def set_chain(self, nodes: Sequence[NodeSpec]):
        arr = (_NodeDesc * max(1, len(nodes)))()
        self._keep = []
        for i, n in enumerate(nodes):
            arr[i].kind, arr[i].mode = n.kind, n.mode
            for k, v in enumerate(n.params):
                arr[i].params[k] = float(v)
            arr[i].delay_len = int(n.delay_len)
            if n.taps_reversed is not None:
                t = np.ascontiguousarray(n.taps_reversed, dtype=np.float64)
                self._keep.append(t)
                arr[i].n_taps = len(t)
                arr[i].taps = t.ctypes.data_as(C.POINTER(C.c_double))
        self._chk(self.L.dspfx_chain_set(self.h, arr, len(nodes)))
        self.nodes = list(nodes)

    @staticmethod
    def _graph_arrays(nodes, links):
        arr = (_NodeDesc * max(1, len(nodes)))()
        for i, n in enumerate(nodes):
            arr[i].kind, arr[i].mode = n.kind, n.mode
            for k, v in enumerate(n.params):
                arr[i].params[k] = float(v)
            arr[i].delay_len = int(n.delay_len)
        larr = (_GraphLink * max(1, len(links)))()
        for i, (s, d, p) in enumerate(links):
            larr[i].src, larr[i].dst, larr[i].port = int(s), int(d), int(p)
        return arr, larr

    def set_graph(self, nodes: Sequence[NodeSpec], links: Sequence[Tuple[int, int, int]]):
        """A whole DAG as one generated kernel (include/dspfx.h, dspfx_graph_set).  `nodes` in an order in which every
        link goes forward; links = (src, dst, port): src a node index, GRAPH_INPUT or GRAPH_ZERO; dst a node index or
        len(nodes) for the Output node; port PORT_MAIN, PORT_SIDE or PORT_SLIDER + k.  Raises DspfxError with status
        ERR_UNSUPPORTED when the graph cannot be fused."""
        arr, larr = self._graph_arrays(nodes, links)
        self._chk(self.L.dspfx_graph_set(self.h, arr, len(nodes), larr, len(links)))
        self.nodes = list(nodes)

    def kernels_ready(self, wait_ms: int = 60000) -> bool:
        """Adopt the kernels the background compiler has finished for this engine's chain and wait up to wait_ms for the rest
        (dspfx_kernels_ready): True when nothing is pending any more.  Results never depend on it."""
        rc = self.L.dspfx_kernels_ready(self.h, int(wait_ms))
        if rc < 0:
            self._chk(rc)
        return rc == 1

    def set_param(self, node: int, param: int, value: float):
        self._chk(self.L.dspfx_set_param(self.h, node, param, float(value)))

    def set_mode(self, node: int, mode: int):
        self._chk(self.L.dspfx_set_mode(self.h, node, int(mode)))

    def set_param_seq(self, node: int, param: int, value: float) -> int:
        """set_param that also returns the store's sequence number (safe from any thread: the store is queued and
        applied at the next block boundary, dspfx.h)."""
        seq = C.c_uint64()
        self._chk(self.L.dspfx_set_param_seq(self.h, node, param, float(value), C.byref(seq)))
        return int(seq.value)

    def param_log(self, after_seq: int = 0, cap: int = 4096):
        """The stores applied so far with seq > after_seq: [(seq, frame, node, param, value)], oldest first; `frame` =
        frames submitted when the store took effect (param -1: a mode store)."""
        arr = (_ParamEvent * cap)()
        n = self.L.dspfx_param_log(self.h, arr, cap, int(after_seq))
        if n < 0:
            self._chk(n)
        return [(int(a.seq), int(a.frame), int(a.node), int(a.param), float(a.value)) for a in arr[:n]]

    def frames_submitted(self) -> int:
        return int(self.L.dspfx_frames_submitted(self.h))

    def set_delay_len(self, node: int, d: int):
        self._chk(self.L.dspfx_set_delay_len(self.h, node, int(d)))

    def reserve_delay_len(self, node: int, d: int):
        """Capacity hint (any thread, takes no engine lock): the groups a ring of d samples would need are allocated now."""
        self._chk(self.L.dspfx_reserve_delay_len(self.h, node, int(d)))

    def ring_trim(self):
        self._chk(self.L.dspfx_ring_trim(self.h))

    def set_taps(self, node: int, impulse_response, mode: int = FIR_BALANCED):
        t = np.ascontiguousarray(np.asarray(impulse_response, np.float64)[::-1])
        self._chk(self.L.dspfx_set_taps(self.h, node, t.ctypes.data_as(C.POINTER(C.c_double)), len(t), mode))

    def set_fir_precision(self, node: int, precision: int):
        """FIR_PRECISION_DEFAULT / _F32 / _SPLIT (dspfx_set_fir_precision): how the node's steady-state sweep multiplies."""
        self._chk(self.L.dspfx_set_fir_precision(self.h, node, precision))

    def reset(self):
        self._chk(self.L.dspfx_reset(self.h))

    def process(self, x, out=None, side=None, mix=None, n_frames: Optional[int] = None, stream: int = 0, ctl=None):
        """Device path: x/out/side are [n_frames][channels] f32 device tensors (or raw pointers).
        ctl: {(node, slider): device tensor} = connected `as_input` control ports for this block."""
        if n_frames is None:
            n_frames = x.shape[0]
        if out is None:
            out = x
        st = C.c_void_p(stream) if stream else None
        if ctl:
            arr = (_Ctl * len(ctl))()
            for i, ((node, param), sig) in enumerate(ctl.items()):
                arr[i].node, arr[i].param, arr[i].signal = int(node), int(param), _ptr(sig).value
            self._chk(self.L.dspfx_process_ctl(self.h, _ptr(x), _ptr(side), _ptr(out), _ptr(mix), int(n_frames),
                                               arr, len(ctl), st))
        else:
            self._chk(self.L.dspfx_process(self.h, _ptr(x), _ptr(side), _ptr(out), _ptr(mix), int(n_frames), st))
        return out

    def process_bus(self, x, out, mix, n_frames: int, n_connected: int = 0, side=None, stream: int = 0):
        """One block with the Output node complete (dspfx_process_bus): `mix` = this block's bus, summed inside the chain
        launch and divided by link_divisor(n_connected) when n_connected != 0."""
        self._chk(self.L.dspfx_process_bus(self.h, _ptr(x), _ptr(side), _ptr(out), _ptr(mix), int(n_frames),
                                           int(n_connected), C.c_void_p(stream) if stream else None))

    def process_io(self, ins, outs, n_frames: int, mix=None, stream: int = 0):
        """A graph engine with several input / output blocks (dspfx_process_io): ins[k] = input block k, outs[m] = output
        block m; unused entries may be None."""
        ia = (C.c_void_p * max(1, len(ins)))(*[(_ptr(t).value if t is not None else None) for t in ins])
        oa = (C.c_void_p * max(1, len(outs)))(*[(_ptr(t).value if t is not None else None) for t in outs])
        self._chk(self.L.dspfx_process_io(self.h, ia, len(ins), oa, len(outs), _ptr(mix), int(n_frames),
                                          C.c_void_p(stream) if stream else None))
        return outs[0]

    def process_partials(self, x, out=None, side=None, n_frames: Optional[int] = None, stream: int = 0):
        """Pipelined mix bus, part 1 (stream A): the chain, partial sums stay inside the engine."""
        if n_frames is None:
            n_frames = x.shape[0]
        if out is None:
            out = x
        self._chk(self.L.dspfx_process_partials(self.h, _ptr(x), _ptr(side), _ptr(out), int(n_frames),
                                                C.c_void_p(stream) if stream else None))
        return out

    def mix_collect(self, mix, n_frames: int, stream: int = 0):
        """Pipelined mix bus, part 2 (stream B): wait for the chain kernel, reduce partials -> mix[n_frames]."""
        self._chk(self.L.dspfx_mix_collect(self.h, _ptr(mix), int(n_frames), C.c_void_p(stream) if stream else None))

    def process_host(self, x: np.ndarray, side: Optional[np.ndarray] = None, want_mix: bool = False, out=None):
        """Host path (numpy in / numpy out): H2D, process, D2H.  `out` may be a caller-provided (e.g. pinned) array."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.ndim == 2 and x.shape[1] == self.channels, x.shape
        if out is None:
            out = np.empty_like(x)
        assert out.dtype == np.float32 and out.shape == x.shape and out.flags.c_contiguous
        s = np.ascontiguousarray(side, dtype=np.float32) if side is not None else None
        mix = np.empty(x.shape[0], np.float32) if want_mix else None
        self._chk(self.L.dspfx_process_host(self.h, x.ctypes.data, s.ctypes.data if s is not None else None,
                                            out.ctypes.data, mix.ctypes.data if want_mix else None, x.shape[0]))
        return (out, mix) if want_mix else out

    def mix_finish(self, mix, n_frames: int, n_connected: int, stream: int = 0):
        self._chk(self.L.dspfx_mix_finish(self.h, _ptr(mix), int(n_frames), int(n_connected),
                                          C.c_void_p(stream) if stream else None))

    def mix_allreduce(self, comm: "Comm", mix, n_frames: int, n_connected: int = 0, stream: int = 0):
        """Sum this rank's un-normalised bus over the communicator's ranks (ONE RCCL all-reduce of n_frames floats, in
        place, asynchronous on `stream`), then the Output hop with the global channel count when n_connected != 0."""
        self._chk(self.L.dspfx_mix_allreduce(self.h, comm.h, _ptr(mix), int(n_frames), int(n_connected),
                                             C.c_void_p(stream) if stream else None))

    def tune_placement(self, x, out, n_frames: int, side=None, stream: int = 0):
        """Re-tune the delay rings' placement with the real chain kernels on the caller's buffers (DSP state is kept)."""
        self._chk(self.L.dspfx_tune_placement(self.h, _ptr(x), _ptr(side), _ptr(out), int(n_frames),
                                              C.c_void_p(stream) if stream else None))

    def process_mixpipe(self, x, out, mix, n_frames: int, n_connected: int = 0, side=None, stream: int = 0):
        """One block with the mix bus pipelined inside the chain kernel: `mix` receives the bus of the block
        submitted two calls earlier (divided by link_divisor(n_connected) when n_connected != 0)."""
        self._chk(self.L.dspfx_process_mixpipe(self.h, _ptr(x), _ptr(side), _ptr(out), _ptr(mix), int(n_frames),
                                               int(n_connected), C.c_void_p(stream) if stream else None))

    def mixpipe_flush(self, mix_older, mix_newer, n_connected: int = 0, stream: int = 0):
        self._chk(self.L.dspfx_mixpipe_flush(self.h, _ptr(mix_older), _ptr(mix_newer), int(n_connected),
                                             C.c_void_p(stream) if stream else None))

    def link_average(self, srcs, dst, n_frames: int, stream: int = 0):
        """collect_and_average (node.rs:162-194) of `srcs` (device buffers, link order) into `dst`."""
        arr = (C.c_void_p * max(1, len(srcs)))(*[_ptr(t).value for t in srcs])
        self._chk(self.L.dspfx_link_average(self.h, arr, len(srcs), _ptr(dst), int(n_frames),
                                            C.c_void_p(stream) if stream else None))

    def state_export(self, node: int) -> np.ndarray:
        n = int(self.L.dspfx_state_size(self.h, node))
        if n < 0:
            self._chk(n)
        buf = np.empty(n, np.uint8)
        if n:
            self._chk(self.L.dspfx_state_export(self.h, node, buf.ctypes.data, n))
        return buf

    def state_import(self, node: int, buf: np.ndarray):
        buf = np.ascontiguousarray(buf).view(np.uint8)
        self._chk(self.L.dspfx_state_import(self.h, node, buf.ctypes.data, buf.size))

    def fill_noise(self, dst, n_frames: int, n_abs0: int, seed: int = 0x5EED0001, stream: int = 0):
        self._chk(self.L.dspfx_fill_noise(self.h, _ptr(dst), int(n_frames), int(n_abs0) & 0xFFFFFFFF, seed,
                                          C.c_void_p(stream) if stream else None))

    def sync(self, stream: int = 0):
        self._chk(self.L.dspfx_sync(self.h, C.c_void_p(stream) if stream else None))

    def describe(self) -> str:
        buf = C.create_string_buffer(1 << 16)
        self._chk(self.L.dspfx_describe(self.h, buf, 1 << 16))
        return buf.value.decode()

    def profile_enable(self, launches: int = 1):
        """0 = off; n > 0 = on, with events for n launches created up front."""
        self._chk(self.L.dspfx_profile_enable(self.h, int(launches)))

    def profile_read(self, reset: bool = True):
        """(total kernel ms, launches, kernel name) of the dominant stage since the last reset."""
        ms, n = C.c_double(), C.c_uint32()
        name = C.create_string_buffer(128)
        self._chk(self.L.dspfx_profile_read(self.h, C.byref(ms), C.byref(n), name, 128, int(reset)))
        return ms.value, n.value, name.value.decode()

    def algorithmic_bytes_per_sample(self, n_frames: int) -> float:
        return float(self.L.dspfx_algorithmic_bytes_per_sample(self.h, int(n_frames)))


def comm_unique_id(backend: Optional[str] = None) -> bytes:
    """The 128-byte id rank 0 creates and hands to every rank (dspfx_comm_unique_id).  backend: None = the library's default
    (DSPFX_COMM_BACKEND, else "mailbox"), or "mailbox" / "rccl" for this id."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    old = os.environ.get("DSPFX_COMM_BACKEND")
    if backend is not None:
        os.environ["DSPFX_COMM_BACKEND"] = backend
    try:
        rc = lib().dspfx_comm_unique_id(buf)
    finally:
        if backend is not None:
            if old is None:
                os.environ.pop("DSPFX_COMM_BACKEND", None)
            else:
                os.environ["DSPFX_COMM_BACKEND"] = old
    if rc != 0:
        raise DspfxError(rc, lib().dspfx_comm_last_error(None).decode() or lib().dspfx_strerror(rc).decode())
    return buf.raw


class Comm:
    """The mix bus' communicator: one per process / GPU (include/dspfx.h, dspfx_comm_create).  `uid` = the bytes of
    comm_unique_id() from rank 0 (may be None for a single rank: then no RCCL communicator is created)."""

    def __init__(self, device: int, n_ranks: int, rank: int, uid: Optional[bytes] = None):
        self.L = lib()
        self.h = C.c_void_p()
        self.n_ranks, self.rank = int(n_ranks), int(rank)
        rc = self.L.dspfx_comm_create(int(device), int(n_ranks), int(rank), uid, C.byref(self.h))
        if rc != 0:
            self.h = C.c_void_p()
            raise DspfxError(rc, self.L.dspfx_comm_last_error(None).decode() or self.L.dspfx_strerror(rc).decode())

    @property
    def backend(self) -> str:
        """"mailbox" (one-shot peer-write all-reduce, the default), "rccl" or "single"."""
        return self.L.dspfx_comm_backend(self.h).decode()

    def close(self):
        h = getattr(self, "h", None)
        if h is not None and h.value:
            self.L.dspfx_comm_destroy(h)
            h.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def graph_source(nodes: Sequence[NodeSpec], links: Sequence[Tuple[int, int, int]]) -> str:
    """The translation unit `Engine.set_graph` would compile for this graph (needs no device)."""
    L = lib()
    arr, larr = Engine._graph_arrays(nodes, links)
    buf = C.create_string_buffer(1 << 18)
    rc = L.dspfx_graph_source(arr, len(nodes), larr, len(links), buf, len(buf))
    if rc != 0:
        raise DspfxError(rc, L.dspfx_strerror(rc).decode())
    return buf.value.decode()
