"""The in-launch mix-bus hand-over, checked in the SHIPPED code object.

dsp-stuff_amd/csrc/chain_kernels.hip.h (mix_tail) hands rows between workgroups inside one launch without fences: write-through
(`sc1`) stores, `s_waitcnt vmcnt(0)` before ONE lane's ticket (`global_atomic_add`), `sc1` buffer loads by the workgroup that drew
the last ticket.  That argument rests on the ISA the compiler emits, not on the HIP memory model, and the soak tests would catch a
broken hand-over only with some probability -- so this test disassembles the gfx950 code objects inside libdspfx.so (and the fence
build libdspfx_busfence.so, which must show the opposite shape) and asserts the shape instruction by instruction.  Needs only the
ROCm toolchain (llvm-objdump), no GPU; a compiler update that inserts a cache write-back, drops an sc1 bit or moves the wait fails
here, at build time, instead of silently changing the bus between rounds."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dsp-stuff_amd", "csrc")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
TAIL_BATCH = 16            # chain_kernels.hip.h: rows in flight per lane in tail_reduce_rows

pytestmark = pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="no llvm-objdump in this image")


def bus_kernels(lib, tmp):
    """{kernel symbol: [instruction text, ...]} for every chain kernel of `lib` that takes tickets (global_atomic_add)."""
    work = os.path.join(tmp, os.path.basename(lib) + ".d")
    os.makedirs(work)
    so = shutil.copy(lib, work)                     # llvm-objdump --offloading writes the bundles next to its input
    subprocess.check_call([OBJDUMP, "--offloading", so], cwd=work, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = {}
    for f in sorted(os.listdir(work)):
        if "gfx950" not in f:
            continue
        text = subprocess.check_output([OBJDUMP, "-d", os.path.join(work, f)], text=True)
        for fn in re.split(r"\n(?=[0-9a-f]{16} <)", text):
            m = re.match(r"[0-9a-f]{16} <(\S+)>:", fn)
            if not m or not re.search(r"5dspfx(12chain_kernel|16chain_dyn_kernel|15chain_ts_kernel)", m.group(1)):
                continue
            ins = [l.split("//")[0].strip() for l in fn.splitlines()[1:]]
            ins = [i for i in ins if i]
            if any(i.startswith("global_atomic_add") for i in ins):
                out[m.group(1)] = ins
    return out


def _is_store(i):
    return i.startswith(("global_store", "buffer_store", "flat_store", "scratch_store"))


def check_write_through_form(kernels):
    """the default build: sc1 stores -> s_waitcnt vmcnt(0) -> ticket -> sc1 loads; no cache write-back / invalidate anywhere"""
    assert len(kernels) >= 30, len(kernels)                       # every compiled-in chain kernel carries the tail
    for name, ins in kernels.items():
        tickets = [k for k, i in enumerate(ins) if i.startswith("global_atomic_add")]
        for k in tickets:
            assert "sc0" in ins[k] and "sc1" not in ins[k], (name, ins[k])      # returning, agent scope (sc1 would be system scope)
            back = ins[max(0, k - 96):k][::-1]
            hit = next((j for j, i in enumerate(back) if _is_store(i) or re.match(r"s_waitcnt\b.*vmcnt\(0\)", i)), None)
            assert hit is not None and not _is_store(back[hit]), (name, "a store between the last s_waitcnt vmcnt(0) and the ticket", back[:hit + 1][::-1])
            nxt = next((i for i in ins[k + 1:] if i.startswith("buffer_load_dwordx2") or i.startswith("global_atomic_add")), None)
            assert nxt is not None and nxt.startswith("buffer_load_dwordx2") and " sc1" in nxt, (name, "the tail behind a ticket must read with sc1", nxt)
        sc1_loads = [i for i in ins if i.startswith("buffer_load_dwordx2") and " sc1" in i]
        assert len(sc1_loads) == TAIL_BATCH * len(tickets), (name, len(sc1_loads), len(tickets))
        sc1_stores = [i for i in ins if i.startswith("global_store_dword") and " sc1" in i]
        assert len(sc1_stores) >= 2 * len(tickets), (name, len(sc1_stores), len(tickets))   # rows / slice sums + the ticket resets
        assert not any(i.startswith(("buffer_wbl2", "buffer_inv")) for i in ins), (name, "a cache write-back / invalidate was inserted")


def check_fence_form(kernels):
    """DSPFX_BUS_FENCE: release (buffer_wbl2) before every ticket, acquire (buffer_inv) behind it, plain loads in the tail"""
    assert len(kernels) >= 30, len(kernels)
    for name, ins in kernels.items():
        tickets = [k for k, i in enumerate(ins) if i.startswith("global_atomic_add")]
        assert sum(i.startswith("buffer_wbl2") for i in ins) == len(tickets), name
        assert sum(i.startswith("buffer_inv") for i in ins) == len(tickets), name
        assert not any(i.startswith("buffer_load_dwordx2") and " sc1" in i for i in ins), name
        for k in tickets:
            back = ins[max(0, k - 96):k][::-1]
            hit = next((j for j, i in enumerate(back) if _is_store(i) or i.startswith("buffer_wbl2")), None)
            assert hit is not None and back[hit].startswith("buffer_wbl2"), (name, "no release in front of the ticket")


@pytest.fixture(scope="module")
def disassembly(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("isa"))
    libs = {n: os.path.join(CSRC, n) for n in ("libdspfx.so", "libdspfx_busfence.so")}
    for p in libs.values():
        if not os.path.exists(p):
            pytest.fail(f"{p} is missing: run __graft_entry__.build()")
    return {n: bus_kernels(p, tmp) for n, p in libs.items()}


def test_shipped_library_hands_the_bus_over_with_write_through_stores_a_wait_and_sc1_loads(disassembly):
    check_write_through_form(disassembly["libdspfx.so"])


def test_fence_build_has_the_release_acquire_shape(disassembly):
    check_fence_form(disassembly["libdspfx_busfence.so"])


def test_the_checks_tell_the_two_builds_apart(disassembly):
    """swapping the objects must fail: the checks are not vacuous"""
    with pytest.raises(AssertionError):
        check_write_through_form(disassembly["libdspfx_busfence.so"])
    with pytest.raises(AssertionError):
        check_fence_form(disassembly["libdspfx.so"])
