#!/usr/bin/env python3
"""In-process interleaved A/B of chain-kernel variants (guide rule 24: N variants x M rounds,
one process, report the distribution).  Each arm = "name[:key=val,...]" with keys
  lib=<path to .so>  variant=<DSPFX_VARIANT string with ; instead of ,>  tile=<W>
  link=<flags>  chain=chain5|chain3|delay|copy  mix=0|1
Example:
  python tools/ab.py --delay 4096 base:variant=static=1;f=8;cpl=2 nt:lib=dsp-stuff_amd/csrc/libdspfx_nt.so
"""
import argparse
import importlib.util
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def load_pkg(lib_path, tag):
    if lib_path:
        os.environ["DSPFX_LIB"] = os.path.join(ROOT, lib_path) if not os.path.isabs(lib_path) else lib_path
    else:
        os.environ.pop("DSPFX_LIB", None)
    pkg_dir = os.path.join(ROOT, "dsp-stuff_amd")
    name = f"dsp_stuff_amd_{tag}"
    spec = importlib.util.spec_from_file_location(name, os.path.join(pkg_dir, "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    mod.lib()
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("arms", nargs="+")
    ap.add_argument("--channels", type=int, default=1 << 20)
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--delay", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--steps", type=int, default=40)
    args = ap.parse_args()
    import torch
    import chains
    dev = torch.device("cuda", 0)
    N, B = args.channels, args.frames
    arms = []
    for i, spec in enumerate(args.arms):
        name, _, rest = spec.partition(":")
        kv = dict(x.split("=", 1) for x in rest.split(",") if x)
        pkg = load_pkg(kv.get("lib"), f"{i}")
        if "variant" in kv:
            os.environ["DSPFX_VARIANT"] = kv["variant"].replace(";", ",")
        else:
            os.environ.pop("DSPFX_VARIANT", None)
        chain_name = kv.get("chain", "chain5")
        chain = {"chain5": lambda: chains.chain5(pkg, args.delay), "chain3": lambda: chains.chain3(pkg, args.delay),
                 "delay": lambda: [pkg.Reverb(delay_samples=args.delay, decay=0.5)], "copy": lambda: []}[chain_name]()
        eng = pkg.Engine(N, B, link_flags=int(kv.get("link", 3)), tile_channels=int(kv.get("tile", 0)))
        eng.set_chain(chain)
        mix = torch.zeros(B, dtype=torch.float32, device=dev) if int(kv.get("mix", 1)) else None
        yoff = int(kv.get("yoff", 0)) // 4          # output buffer offset in bytes (breaks in/out congruence)
        ybuf = torch.empty(B * N + yoff, dtype=torch.float32, device=dev)
        arms.append(dict(name=name, eng=eng, mix=mix, ms=[], kern=None, bps=eng.algorithmic_bytes_per_sample(B),
                         y=ybuf[yoff:]))
        desc = eng.describe().strip().splitlines()
        print(f"# {name}: {[l for l in desc if l.startswith('stage')][-1]}", flush=True)
        arms[-1]["addr"] = " ".join(l.split("@")[1].split(" ")[0] for l in desc if "state @" in l) + f" y=0x{arms[-1]['y'].data_ptr():x}"
    x = torch.empty(B * N, dtype=torch.float32, device=dev)
    arms[0]["eng"].fill_noise(x, B, 0)
    stream = torch.cuda.current_stream().cuda_stream
    for a in arms:   # warm up: fill rings, create events
        a["eng"].profile_enable(args.steps + 4)
        a["eng"].profile_enable(0)
        for _ in range(max(8, args.delay // B + 2)):
            a["eng"].process(x, out=a["y"], mix=a["mix"], n_frames=B, stream=stream)
    torch.cuda.synchronize()
    for r in range(args.rounds):
        order = arms if r % 2 == 0 else arms[::-1]
        for a in order:
            a["eng"].profile_enable(1)
            for _ in range(args.steps):
                a["eng"].process(x, out=a["y"], mix=a["mix"], n_frames=B, stream=stream)
            torch.cuda.synchronize()
            a["eng"].profile_enable(0)
            ms, n, kern = a["eng"].profile_read()
            a["ms"].append(ms / max(n, 1))
            a["kern"] = kern
    base = statistics.median(arms[0]["ms"])
    for a in arms:
        med, mn = statistics.median(a["ms"]), min(a["ms"])
        gbs = a["bps"] * N * B / (med * 1e-3) / 1e9
        print(f"{a['name']:>14s} {a['kern']:>12s}  median {med:.4f} ms  min {mn:.4f}  max {max(a['ms']):.4f}  "
              f"{gbs:6.0f} GB/s  vs first {base / med:.3f}x  {a.get('addr', '')}")
    print(f"# x=0x{x.data_ptr():x}")


if __name__ == "__main__":
    main()
