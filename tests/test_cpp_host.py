"""The C++ host-side mirror (include/dspfx.hpp): compiles with plain g++ against the C ABI and, on the
GPU box, matches the oracle from C++ (tests/cpp/test_host.cpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_host")


def _build():
    import oracle as O
    O.build()
    cs = os.path.join(ROOT, "dsp-stuff_amd", "csrc")
    orc = os.path.join(ROOT, "oracle")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-o", EXE, os.path.join(ROOT, "tests", "cpp", "test_host.cpp"),
           f"-L{cs}", "-ldspfx", f"-L{orc}", "-loracle", f"-Wl,-rpath,{cs}", f"-Wl,-rpath,{orc}", "-L/opt/rocm/lib",
           "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined", "-pthread", "-rdynamic"]
    subprocess.check_call(cmd)


def test_cpp_header_compiles_and_fails_loudly_without_gpu():
    import torch
    _build()
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([EXE, "--expect-no-device"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "no HIP device" in r.stdout


@pytest.mark.gpu
def test_cpp_host_parity_on_gpu():
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "max ulp" in r.stdout
    assert "mix_allreduce over a 1-rank mailbox communicator ok" in r.stdout and "mix_allreduce over a 1-rank rccl communicator ok" in r.stdout
    assert "GpuBank call sequence: 64 ports" in r.stdout          # host/rust/src/gpu_bank.rs' C calls, in its order, against the oracle
    assert "slider stores from a second thread" in r.stdout
