#!/usr/bin/env python3
"""Study (round 6): shader-clock cycles per phase of solver iteration 2, apart for the two branches of the routed
control_steps-3 kernel (dense: free reach tile; stage-wise: anything else), for a lone wave of each kind and for one
residency round (4096).  Needs the timing build: make -C neo_mpc_planner2_amd/csrc timing;
NEO_MPC_LIB=neo_mpc_planner2_amd/libneo_mpc_timing.so python tools/phase_timing_routed.py"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neo_mpc_planner2_amd import synthetic
from neo_mpc_planner2_amd.solver import BatchSolver
from neo_mpc_planner2_amd.mpc_optimization_server import README_PARAMS
from oracle import c_oracle   # (test infrastructure: only to know which instances take which branch)
params = dict(README_PARAMS); params.update(control_steps=3)
names = ["adjoint", "cone", "direction", "restrict/early", "candidates", "accept"]
cfg, cmap, probs, st, warm = synthetic.make_workload("C2", seed=0, batch=4096)
routed = c_oracle.route_batch(params, cmap, probs).astype(bool)
def run(sel, tag):
    p = np.ascontiguousarray(probs[sel])
    with BatchSolver(params) as s:
        s.set_costmap(*cmap)
        for rep in range(2):
            st, warm = synthetic.make_states(p, 3)
            cmds, x = s.solve(p, st, warm)
    for name, m in (("dense", ~routed[sel]), ("stage-wise", routed[sel])):
        ok = (cmds["iterations"] >= 3) & m
        if not ok.any(): continue
        t = x[ok][:, :6]
        print("%-12s %-10s (%4d instances, mean iterations %.2f): " % (tag, name, ok.sum(), cmds["iterations"][m].mean()) +
              ", ".join("%s %.0f" % (n, v) for n, v in zip(names, t.mean(axis=0))) + " | total %.0f cycles" % t.sum(axis=1).mean())
idx_d = np.nonzero(~routed)[0][:8]; idx_r = np.nonzero(routed)[0][:8]
for i in idx_d[:3]: run(np.array([i]), "lone wave")
for i in idx_r[:3]: run(np.array([i]), "lone wave")
run(np.arange(4096), "4096")
