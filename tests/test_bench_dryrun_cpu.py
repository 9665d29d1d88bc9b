"""bench.py's --gpus N control flow without a GPU: `bench.py --dry-run` under the driver's own launcher
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 ...`) with the gloo backend.
It executes what an 8-GPU run executes on the host before and around the first GPU call: RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* handling, the --gpus check, the rendezvous, the communicator's unique-id broadcast and its all-ranks fallback
(there is no device here, so dspfx_comm_create fails on every rank and all of them must fall back together), weak
channel sharding, parallel.PipelinedMixBus over a real two-rank collective, the MAX over ranks and rank 0's line."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(nproc, extra=(), env=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "20", "--warmup", "5",
           "--dry-run", *extra]
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)


@pytest.mark.parametrize("mix", ["inline", "pipe"])
def test_two_rank_dry_run_prints_one_line_and_the_bus_adds_up(mix):
    r = _run(2, env={"DSPFX_BENCH_MIX": mix, "DSPFX_BENCH_MIX_BATCH": "4"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 only
    d = json.loads(lines[0])
    assert d["dry_run"] and d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak"
    c = d["config"]
    assert c["total_channels"] == 2 * c["channels_per_gpu"] and c["channel_offset_of_last_rank"] == c["channels_per_gpu"]
    assert c["parallelism"] == "channel-shard x2" and c["mix_bus"] == mix
    assert "fallback" in c["collective"]                   # no device: every rank took the torch.distributed path together
    assert d["bus_ok"] and d["bus_checked_blocks"] >= 8    # rows of the rings == sum over both ranks / f32(0.0001 + N)
    assert r.stderr.count("C-ABI communicator unavailable") == 2


def test_gpus_flag_must_match_the_launcher():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def _no_launcher_env():
    return {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                              "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}


def test_gpus_2_without_a_launcher_starts_its_own_ranks():
    """`python3 bench.py --gpus 2 ...` as the driver records the N = 1 command, with no torch.distributed.run in front of it: the
    process starts its two ranks itself (fresh children, before any GPU call of its own), rank 0's ONE line comes through on stdout,
    the exit status is the launcher's.  Both forms of the bus exchange are driven and checked against the sum over ranks."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-run"],
                       cwd=ROOT, env=_no_launcher_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] and d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["bus_ok"]
    assert "starting 2 ranks" in r.stderr and "torch.distributed.run" in r.stderr
    f = d["scaling_forms"]
    assert f["inline"]["bus_delay_blocks"] == 0 and f["overlapped"]["bus_delay_blocks"] == 2 and f["overlapped"]["bus_checked_blocks"] >= 3
    assert f["same_block_second_stream"]["bus_delay_blocks"] == 0 and f["same_block_second_stream"]["bus_checked_blocks"] >= 3


def test_gpus_2_without_enough_gpus_fails_with_one_line_and_never_runs_on_one():
    """No GPU in this container: the real (non-dry) command must exit non-zero with the reason, before starting anything."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       cwd=ROOT, env=_no_launcher_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "needs 2 visible GPU(s)" in r.stderr and "starting" not in r.stderr


def test_a_failing_rank_fails_the_self_launched_run():
    env = _no_launcher_env()
    env["DSPFX_BENCH_DRY_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "0", "--dry-run"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


def test_single_process_dry_run():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "9", "--warmup", "0"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["config"]["collective"] is None and d["bus_ok"]
