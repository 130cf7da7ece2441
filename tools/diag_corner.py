#!/usr/bin/env python3
"""Development aid (GPU box): K1's direction per iteration against the CPU mirror's on one instance of the corner
regression sets of tests/util.py (check_stop_rule_regressions)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import util
from neo_mpc_planner2_amd import synthetic
from neo_mpc_planner2_amd.solver import BatchSolver
from oracle import c_oracle

pset, seed, inst = {"corner3": (util.CORNER3_SET, 117, 151), "corner": (util.CORNER_SET, 110, 233)}[sys.argv[1] if len(sys.argv) > 1 else "corner3"]
params = util.orc.make_params(**pset)
n = params["control_steps"]
_, cmap, probs, _, _ = synthetic.make_workload("C2", seed=seed, batch=256)
free = (np.zeros_like(cmap[0]),) + tuple(cmap[1:])
pr = probs[inst:inst + 1]
lib = c_oracle.load()
np.set_printoptions(precision=5, suppress=True, linewidth=200)
with BatchSolver(params) as s:
    s.set_costmap(*free)
    for it in range(1, 8):
        dg = s.direction(pr, np.zeros((1, 3 * n)), it)[0]
        buf = np.full(3 * n, np.nan)
        lib.orc_capture_direction(it, C.c_void_p(buf.ctypes.data))
        st, w = synthetic.make_states(pr, n)
        c_oracle.solve_batch(params, free, pr, st, w)
        lib.orc_capture_direction(-1, None)
        _, xm, _ = c_oracle.solve_batch(dict(params, max_iterations=it), free, pr, *synthetic.make_states(pr, n))
        print("iteration", it, "\n  gpu   ", dg, "\n  mirror", buf, "\n  mirror iterate", xm[0])
    st, w = synthetic.make_states(pr, n)
    cm, x = s.solve(pr, st, w)
    print("gpu answer", x[0], cm["iterations"])
