#!/usr/bin/env python
"""DSDGP_POTRF_TIMING=1: per-phase shader clocks of layer 0's Cholesky + inverse (k_potrf_trtri, one LDS-resident workgroup)."""
import os, sys
os.environ["DSDGP_POTRF_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "doubly-stochastic-dgp_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_configs as BC
import ctypes as C
for cfg in (2,):
    model, step = BC.build(cfg)
    step(); step()
    eng = model.engine()
    info = C.c_int(0)
    from doubly_stochastic_dgp import _lib
    _lib.check(eng.lib.dsdgp_model_theta_changed(eng.model))
    _lib.check(eng.lib.dsdgp_model_prepare(eng.model, C.byref(info)))
