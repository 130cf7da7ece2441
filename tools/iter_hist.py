#!/usr/bin/env python3
"""Study: iteration-count distribution of one workload on the GPU (tail instances set a launch's duration)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neo_mpc_planner2_amd import synthetic
from neo_mpc_planner2_amd.solver import BatchSolver
from neo_mpc_planner2_amd.mpc_optimization_server import README_PARAMS
w = sys.argv[1] if len(sys.argv) > 1 else "C5"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else None
cfg, cmap, probs, st, warm = synthetic.make_workload(w, seed=0, batch=batch)
params = dict(README_PARAMS); params.update(control_steps=cfg["control_steps"])
with BatchSolver(params) as s:
    s.set_costmap(*cmap)
    cmds, x = s.solve(probs, st, warm)
it = cmds["iterations"]
print(w, "mean %.2f p50 %d p90 %d p99 %d p99.9 %d max %d; above 20: %d, above 30: %d, above 50: %d, at the cap: %d of %d"
      % (it.mean(), np.percentile(it, 50), np.percentile(it, 90), np.percentile(it, 99), np.percentile(it, 99.9), it.max(),
         (it > 20).sum(), (it > 30).sum(), (it > 50).sum(), (cmds["status"] != 0).sum(), len(it)))
