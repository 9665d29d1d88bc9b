#!/usr/bin/env python3
"""Which of torch and libdspfx.so may touch HIP first?  (dsp-stuff_amd.lib() imports torch before dlopen because of this.)
modes: torch_first | lib_first | lib_child | count_first -- see the prints."""
import sys, subprocess
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
from __graft_entry__ import load_package
E = load_package()
if mode in ("lib_first", "lib_child"):
    E.lib()
    print("lib loaded, device_count:", end=" ")
if mode == "lib_child":
    subprocess.run(["true"])
if mode == "count_first":
    print("device_count before torch:", E.device_count())
import torch
torch.zeros(4, device="cuda")
print("torch ok;", "dspfx device_count =", E.device_count())
try:
    e = E.Engine(64, 128); print("engine ok"); e.close()
except Exception as ex:
    print("ENGINE FAILED:", ex)
