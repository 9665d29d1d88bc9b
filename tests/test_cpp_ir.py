"""The C++ impulse-response loader (include/dspfx_ir.hpp) against the Python mirror (dsp-stuff_amd/ir.py): the same taps,
bit for bit, for every sample format, for a resampled file, and the same refusals."""
import os
import subprocess

import numpy as np

from test_config_ir_cpu import _wav

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_ir")


def _run(*args):
    return subprocess.run([EXE] + [str(a) for a in args], capture_output=True, text=True)


def test_cpp_ir_loader_matches_the_python_mirror(dspfx, tmp_path):
    from dsp_stuff_amd import ir
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-ffp-contract=off", "-o", EXE,
                           os.path.join(ROOT, "tests", "cpp", "test_ir.cpp")])
    rng = np.random.default_rng(1)
    q = np.round(rng.uniform(-0.9, 0.9, (96, 2)) * 32768) / 32768
    cases = [(1, 16, 48000, q), (1, 24, 48000, q), (3, 32, 48000, q), (1, 8, 48000, np.round(q[:, :1] * 128) / 128),
             (1, 16, 44100, q), (1, 16, 96000, q[:, :1]), (3, 32, 22050, q)]
    for k, (tag, bits, rate, frames) in enumerate(cases):
        p = tmp_path / f"ir{k}.wav"
        _wav(str(p), tag, bits, rate, frames)
        want = ir.load_impulse_response(str(p))
        r = _run(p)
        assert r.returncode == 0, r.stdout
        got = np.array([float.fromhex(v) for v in r.stdout.split()], np.float64)
        assert got.shape == want.shape and np.array_equal(got.view(np.uint64), want.view(np.uint64)), (tag, bits, rate)
    r = _run(tmp_path / "ir4.wav", "--no-resample")
    assert r.returncode == 3 and "48 kHz" in r.stdout
    bad = tmp_path / "bad.wav"
    bad.write_bytes(b"RIFFxxxxWAVE")
    assert _run(bad).returncode == 3
