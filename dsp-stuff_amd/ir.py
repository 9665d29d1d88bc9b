"""Impulse-response loading for the Fir node (the step before the FIR kernel).

Follows nodes/fir.rs:86-173: decode the file to f64 samples, average the channels of every
frame (`s.iter().sum::<f64>() / num_channels as f64`, fir.rs:140-144); the reversal
(fir.rs:163,168) happens in `Fir(...)` / `dspfx_set_taps`.  The reference decodes through
symphonia and, when the file is not 48 kHz, sinc-resamples it through dasp (fir.rs:153-171).
Neither crate is in the reference tree: this reader handles RIFF/WAVE PCM (8/16/24/32-bit
integer, 32/64-bit float; integer samples scaled like symphonia's SampleBuffer<f64> conversion:
u8 (x-128)/128, i16 /2^15, i24 /2^23, i32 /2^31), and files of another rate go through
`resample_dasp_sinc`, a restatement of dasp 0.11's Converter + 16-tap Sinc as recalled
(oracle/dspfx_oracle.h writes the algorithm out; KAT-12 checks its closed forms); pass
resample=False to refuse such files instead.  include/dspfx_ir.hpp is the C++ twin (bit-identical
taps: tests/test_cpp_ir.py); tests/test_gpu_parity.py takes WAV files through to the FIR kernels.
"""
from __future__ import annotations

import struct

import math

import numpy as np


class IrError(ValueError):
    pass


def read_wav(path: str):
    """-> (frames [n][channels] float64, sample_rate)"""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise IrError("not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            if size < 16:
                raise IrError("short fmt chunk")
            tag, ch, rate, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE and size >= 26:          # WAVE_FORMAT_EXTENSIBLE: sub-format GUID's first word
                tag = struct.unpack("<H", body[24:26])[0]
            fmt = (tag, ch, rate, bits)
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise IrError("missing fmt or data chunk")
    tag, ch, rate, bits = fmt
    if ch < 1:
        raise IrError("no channels")
    if tag == 1:     # integer PCM
        if bits == 8:
            x = (np.frombuffer(pcm, np.uint8).astype(np.float64) - 128.0) / 128.0
        elif bits == 16:
            x = np.frombuffer(pcm[:len(pcm) // 2 * 2], "<i2").astype(np.float64) / 32768.0
        elif bits == 24:
            b = np.frombuffer(pcm[:len(pcm) // 3 * 3], np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v & 0x800000, v - 0x1000000, v)
            x = v.astype(np.float64) / 8388608.0
        elif bits == 32:
            x = np.frombuffer(pcm[:len(pcm) // 4 * 4], "<i4").astype(np.float64) / 2147483648.0
        else:
            raise IrError(f"unsupported PCM width {bits}")
    elif tag == 3:   # IEEE float
        if bits == 32:
            x = np.frombuffer(pcm[:len(pcm) // 4 * 4], "<f4").astype(np.float64)
        elif bits == 64:
            x = np.frombuffer(pcm[:len(pcm) // 8 * 8], "<f8").astype(np.float64)
        else:
            raise IrError(f"unsupported float width {bits}")
    else:
        raise IrError(f"unsupported WAVE format tag {tag}")
    n = len(x) // ch
    return x[:n * ch].reshape(n, ch), rate


def resample_dasp_sinc(samples, source_hz: float, target_hz: float = 48000.0) -> np.ndarray:
    """fir.rs:153-165: `from_iter(samples).from_hz_to_hz(Sinc::new(Fixed::from([0.0; 16])), rate, 48000).until_exhausted()`.

    dasp 0.11.0 (dasp_signal::interpolate::Converter + dasp_interpolate::sinc::Sinc) is not under /root/reference;
    this follows its published source AS RECALLED (same caveat as the Envelope node): a 16-frame ring that new
    source frames enter at the back, an index that climbs to depth 8, a Hann-windowed sinc summed outward from the
    index with the depth clipped at the ring's ends, the converter stepping `interpolation_value` by
    source_hz / target_hz and pulling one source frame per whole step.  Sequential f64 arithmetic like the crate."""
    src = [float(v) for v in np.asarray(samples, np.float64)]
    ring = [0.0] * 16
    depth, idx = 8, 0
    ratio = float(source_hz) / float(target_hz)
    value, pos, out = 0.0, 0, []
    pi = math.pi
    while True:
        if pos >= len(src) and value >= 1.0:          # Converter::is_exhausted
            break
        while value >= 1.0:                            # advance whole source frames
            nxt = src[pos] if pos < len(src) else 0.0  # FromIterator yields equilibrium once the iterator has ended
            pos += 1
            ring.pop(0)
            ring.append(nxt)
            if idx < depth:
                idx += 1
            value -= 1.0
        phil, phir = value, 1.0 - value
        nl, nr = idx, idx + 1
        rightmost, leftmost = nl + depth, nr - depth
        if rightmost >= len(ring):
            max_depth = len(ring) - depth
        elif leftmost < 0:
            max_depth = depth + leftmost
        else:
            max_depth = depth
        v = 0.0
        for n in range(max_depth):
            a = pi * (phil + n)
            first = 1.0 if a == 0.0 else math.sin(a) / a
            second = 0.5 + 0.5 * math.cos(a / depth)
            v = v + (first * second) * ring[(nl - n) % 16]
            a = pi * (phir + n)
            first = 1.0 if a == 0.0 else math.sin(a) / a
            second = 0.5 + 0.5 * math.cos(a / depth)
            v = v + (first * second) * ring[(nr + n) % 16]     # ring_buffer::Fixed indexes modulo its length
        out.append(v)
        value += ratio
    return np.asarray(out, np.float64)


def load_impulse_response(path: str, resample: bool = True) -> np.ndarray:
    """h[0..T) in natural order, f64: channels averaged like fir.rs:140-144; a file that is not 48 kHz goes through
    `resample_dasp_sinc` like fir.rs:153-165 (pass resample=False to refuse it instead)."""
    frames, rate = read_wav(path)
    if rate != 48000 and not resample:
        raise IrError(f"{rate} Hz impulse response: convert the file to 48 kHz (or allow resampling)")
    if len(frames) == 0:
        raise IrError("empty impulse response")
    ch = frames.shape[1]
    # sequential f64 sum of the frame's channels, then one division (Iterator::sum order)
    acc = np.zeros(len(frames), np.float64)
    for c in range(ch):
        acc = acc + frames[:, c]
    mono = acc / float(ch)
    return mono if rate == 48000 else resample_dasp_sinc(mono, float(rate), 48000.0)
