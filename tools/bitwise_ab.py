#!/usr/bin/env python3
"""Development aid (GPU box): are two builds of libneo_mpc bit-identical on a workload?  Solves the first 4096 instances of
C3 / C5 / the "turn" set with the library NEO_MPC_LIB points at and prints a digest of the raw solutions.
usage: NEO_MPC_LIB=<lib> python tools/bitwise_ab.py"""
import hashlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from neo_mpc_planner2_amd import synthetic
from neo_mpc_planner2_amd.solver import BatchSolver
for wl, over in (("C3", {}), ("C5", {}), ("C2", bench.GENERAL_SETS["C2/turn"]), ("C2", {})):
    cfg, cmap, probs, st, warm = synthetic.make_workload(wl, seed=0, batch=4096)
    params = bench.readme_params(cfg["control_steps"]); params.update(over)
    with BatchSolver(params) as s:
        s.set_costmap(*cmap)
        cm, x = s.solve(probs, st, warm)
    print(wl, sorted(over)[:1], hashlib.md5(np.ascontiguousarray(x).tobytes()).hexdigest()[:16], "iterations %.3f" % cm["iterations"].mean())
