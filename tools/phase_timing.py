#!/usr/bin/env python3
"""Study: shader-clock cycles per phase of solver iteration 2 for a lone wave (count = 1), one residency
round (4096) and steady state (65 536).  Needs the timing build: make -C neo_mpc_planner2_amd/csrc timing;
NEO_MPC_LIB=neo_mpc_planner2_amd/libneo_mpc_timing.so python tools/phase_timing.py"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neo_mpc_planner2_amd import synthetic
from neo_mpc_planner2_amd.solver import BatchSolver
from neo_mpc_planner2_amd.mpc_optimization_server import README_PARAMS
workload = sys.argv[1] if len(sys.argv) > 1 else "C2"
n = synthetic.CONFIGS[workload]["control_steps"]
params = dict(README_PARAMS); params.update(control_steps=n)
names = ["adjoint sweep", "cone", "direction (Newton system / Riccati sweep)", "restrict/early", "candidates", "accept"]
for count in (1, 4096, 65536):
    cfg, cmap, probs, st, warm = synthetic.make_workload(workload, seed=0, batch=count)
    with BatchSolver(params) as s:
        s.set_costmap(*cmap)
        for rep in range(2):
            st, warm = synthetic.make_states(probs, n)
            cmds, x = s.solve(probs, st, warm)
    ok = cmds["iterations"] >= 3
    t = x[ok][:, :6]
    print("count %6d: cycles per phase (iteration 2, mean over %d instances): " % (count, ok.sum()) + ", ".join("%s %.0f" % (n, v) for n, v in zip(names, t.mean(axis=0))) + "  | total %.0f" % t.sum(axis=1).mean())
