"""host/rust's integration recipe against the real reference tree (tools/check_rust_recipe.py): the shim's trait
implementations have the reference's signatures, the lines the README tells a maintainer to add fit their anchors in
nodes/mod.rs and main.rs.  Runs where /root/reference exists (the build container); the GPU box has no reference tree."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DSPFX_REFERENCE", "/root/reference")


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "dsp-stuff", "src", "nodes", "mod.rs")), reason="no reference tree here")
def test_the_rust_recipe_fits_the_reference_tree():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_rust_recipe.py")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok  ") >= 9, r.stdout


def test_the_checker_notices_a_wrong_signature(tmp_path):
    """The check is not vacuous: the same script over a copy of the shim whose `process` takes its arguments in the wrong order fails."""
    if not os.path.exists(os.path.join(REF, "dsp-stuff", "src", "node.rs")):
        pytest.skip("no reference tree here")
    import shutil
    root = tmp_path / "repo"
    shutil.copytree(os.path.join(ROOT, "host"), root / "host")
    shutil.copytree(os.path.join(ROOT, "tools"), root / "tools", ignore=shutil.ignore_patterns("__pycache__", "archive", "micro"))
    p = root / "host" / "rust" / "src" / "gpu_chain.rs"
    p.write_text(p.read_text().replace("fn process(&self, inputs: ProcessInput, mut outputs: ProcessOutput)", "fn process(&self, mut outputs: ProcessOutput, inputs: ProcessInput)", 1))
    r = subprocess.run([sys.executable, str(root / "tools" / "check_rust_recipe.py")], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "GpuChain::process" in r.stderr, r.stdout + r.stderr
