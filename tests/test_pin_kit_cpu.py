"""oracle/pin_kit -- the reference-side harness (golden_dump.rs), its case file and the comparator.

The Rust module cannot be compiled here; what runs on CPU: the case file is exactly what export_cases.py makes from the committed
golden vectors; a Python walk of the case file in golden_dump.rs's order of operations over the oracle's nodes reproduces every
expected vector and passes the comparator; the comparator notices a single flipped bit and a contradicting probe; and -- where the
reference tree exists (the build container) -- tools/check_pin_kit.py: the module's crate:: paths, trait calls, third-party call
spellings, the README's main.rs lines and every case's cfg against the real node.rs / nodes/*.rs / Cargo.toml."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DSPFX_REFERENCE", "/root/reference")
KIT = os.path.join(ROOT, "oracle", "pin_kit")


def _run(*args):
    return subprocess.run([sys.executable, *args], cwd=ROOT, capture_output=True, text=True, timeout=600)


def test_cases_json_is_what_the_exporter_makes_from_the_committed_vectors(tmp_path):
    out = tmp_path / "cases.json"
    r = _run(os.path.join(KIT, "export_cases.py"), str(out))
    assert r.returncode == 0, r.stdout + r.stderr
    a, b = json.load(open(out)), json.load(open(os.path.join(KIT, "cases.json")))
    assert a == b, "oracle/pin_kit/cases.json is stale: python oracle/pin_kit/export_cases.py"
    assert len(b["cases"]) >= 56 and {"fresh_reverb", "fresh_biquad", "chain_chain5_link3", "graph_fan_in_three", "dag_exact_7000"} <= {c["name"] for c in b["cases"]}


def test_emulated_harness_reproduces_every_vector_and_the_comparator_can_fail(tmp_path):
    out = tmp_path / "pin_out.json"
    r = _run(os.path.join(ROOT, "tools", "compare_pin.py"), "--emulate", str(out))
    assert r.returncode == 0, r.stdout + r.stderr
    r = _run(os.path.join(ROOT, "tools", "compare_pin.py"), str(out))
    assert r.returncode == 0 and r.stdout.count("PASS") >= 56 and "FAIL" not in r.stdout and "EMULATED" in r.stdout, r.stdout + r.stderr
    d = json.load(open(out))
    d["results"][3]["y"][0][200] ^= 0x10                       # one sample, 16 ulp off
    bad = tmp_path / "bad.json"
    json.dump(d, open(bad, "w"))
    r = _run(os.path.join(ROOT, "tools", "compare_pin.py"), str(bad))
    assert r.returncode == 1 and r.stdout.count("FAIL") == 1, r.stdout
    d = json.load(open(out))
    d["probes"]["biquad_probe"]["y"][-1] ^= 1                  # DirectForm1::run rounds differently than the oracle assumes
    json.dump(d, open(bad, "w"))
    r = _run(os.path.join(ROOT, "tools", "compare_pin.py"), str(bad))
    assert r.returncode == 1 and "NONE of the candidate orders" in r.stdout, r.stdout


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "dsp-stuff", "src", "nodes", "mod.rs")), reason="no reference tree here")
def test_the_kit_fits_the_reference_tree(tmp_path):
    r = _run(os.path.join(ROOT, "tools", "check_pin_kit.py"))
    assert r.returncode == 0 and r.stdout.count("ok  ") == 4, r.stdout + r.stderr
    # not vacuous: a case whose FIR cfg lacks `file_name` (Fir::restore would panic on it) is caught
    import shutil
    root = tmp_path / "repo"
    shutil.copytree(KIT, root / "oracle" / "pin_kit")
    os.makedirs(root / "tools")
    for f in ("check_pin_kit.py", "check_rust_recipe.py"):
        shutil.copy(os.path.join(ROOT, "tools", f), root / "tools" / f)
    p = root / "oracle" / "pin_kit" / "cases.json"
    d = json.load(open(p))
    hit = 0
    for c in d["cases"]:
        for n in c["doc"]["nodes"]:
            if n["typename"] == "fir":
                n["cfg"].pop("file_name", None)
                hit += 1
    assert hit
    json.dump(d, open(p, "w"))
    r = subprocess.run([sys.executable, str(root / "tools" / "check_pin_kit.py")], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "file_name" in r.stderr, r.stdout + r.stderr
