#!/usr/bin/env python
"""Per-phase times (100 MHz s_memrealtime stamps; launch span and late starters per launch) of the forward and backward chain launches of one config shape (DSDGP_FWD_TIMING=1 / DSDGP_BWD_TIMING=1: synchronous debug aids in
csrc/layer_sm_impl.hpp).  usage: python tools/bwd_phases.py 2 [3 ...]"""
import os
import sys

os.environ["DSDGP_BWD_TIMING"] = "1"
os.environ["DSDGP_FWD_TIMING"] = "1"
os.environ["DSDGP_NO_OVERLAP"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "doubly-stochastic-dgp_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_configs as BC  # noqa: E402

for a in sys.argv[1:]:
    import torch
    model, step = BC.build(int(a))
    for i in range(3):
        if i == 2:
            sys.stderr.write(f"== cfg {a}, third step\n")
        step()
    torch.cuda.synchronize()
