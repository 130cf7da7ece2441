#!/usr/bin/env python3
"""Development aid (GPU box): K1 against its CPU mirror on a benchmark workload -- where do results differ, and do the
iteration / evaluation counts (the flow: scan taken, search taken up again) agree?   usage: diag_scan.py [C2|C3|C5] [count]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neo_mpc_planner2_amd import synthetic
from neo_mpc_planner2_amd.solver import BatchSolver
from oracle import c_oracle, mpc_oracle as orc

wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
count = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
cfg, cmap, probs, st, warm = synthetic.make_workload(wl, seed=0, batch=count)
params = orc.make_params(control_steps=cfg["control_steps"])
st_c, warm_c = st.copy(), warm.copy()
with BatchSolver(params) as s:
    s.set_costmap(*cmap)
    cg, xg = s.solve(probs, st, warm)
cc, xc, _ = c_oracle.solve_batch(params, cmap, probs, st_c, warm_c)
dx = np.abs(xg - xc).max(axis=1)
dit = cg["iterations"] != cc["iterations"]
dev = cg["evaluations"] != cc["evaluations"]
print("%s: %d instances; |dx| > 1e-6: %d, > 1e-4: %d, max %.2e; iterations differ: %d; evaluations differ: %d; cost gpu-cpu max %.2e min %.2e"
      % (wl, count, (dx > 1e-6).sum(), (dx > 1e-4).sum(), dx.max(), dit.sum(), dev.sum(), (cg["cost"] - cc["cost"]).max(),
         (cg["cost"] - cc["cost"]).min()))
for j in np.argsort(-dx)[:8]:
    print("  inst %d dx %.2e it %d/%d ev %d/%d cost %.9f/%.9f" % (j, dx[j], cg["iterations"][j], cc["iterations"][j],
          cg["evaluations"][j], cc["evaluations"][j], cg["cost"][j], cc["cost"][j]))
    print("     gpu", np.array2string(xg[j][:12], precision=6, max_line_width=200))
    print("     cpu", np.array2string(xc[j][:12], precision=6, max_line_width=200))
