#!/usr/bin/env python3
"""Development aid (GPU box): where do K1 and its CPU mirror (oracle/mpc_oracle.c part 2) part ways?
Solves the same cold problems with the iteration count capped at 1, 2, 3, ... and prints, per cap, the
largest difference between the two iterates -- the first cap with a large difference is the iteration
whose direction / candidates differ."""
import argparse
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neo_mpc_planner2_amd import synthetic
from neo_mpc_planner2_amd.solver import BatchSolver
from oracle import c_oracle, mpc_oracle as orc

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, nargs="+", default=[2, 5, 8, 32])
ap.add_argument("--method", type=int, default=0)
ap.add_argument("--count", type=int, default=128)
ap.add_argument("--zero-map", action="store_true")
ap.add_argument("--fuzz-seed", type=int, default=None, help="a configuration of tests/test_gpu_fuzz.py instead")
ap.add_argument("--show", type=int, default=0, help="print the iterates of the worst instance")
ap.add_argument("--from-iterate", type=int, default=0, help="k > 0: warm-start both sides from the mirror's iterate after k iterations")
args = ap.parse_args()
if args.fuzz_seed is not None:
    from tests.test_gpu_fuzz import random_params
    rng = np.random.default_rng(1000 + args.fuzz_seed)
    fparams = random_params(rng)
    fres = float(rng.choice([0.025, 0.05, 0.1]))
    args.steps = [fparams["control_steps"]]
for n in args.steps:
    cmap = synthetic.make_costmap(200, seed=3)
    if args.zero_map:
        cmap = (np.zeros_like(cmap[0]),) + cmap[1:]
    probs = synthetic.make_problems(args.count, 200, seed=4 + n)
    if args.fuzz_seed is not None:
        cmap = synthetic.make_costmap(240, seed=args.fuzz_seed, resolution=fres)
        probs = synthetic.make_problems(args.count, 240, seed=args.fuzz_seed + 50, resolution=fres)
        probs["cur_vel"] *= fparams["max_vel_trans"]
    for cap in (1, 2, 3, 4, 6, 8, 100):
        params = orc.make_params(control_steps=n, max_iterations=cap, method=args.method)
        if args.fuzz_seed is not None:
            params = dict(fparams, max_iterations=cap, method=args.method)
        st, warm = synthetic.make_states(probs, n)
        if args.from_iterate:
            _, w0, _ = c_oracle.solve_batch(dict(params, max_iterations=args.from_iterate), cmap, probs, st.copy(), warm.copy())
            warm = np.ascontiguousarray(w0)
        st_c, warm_c = st.copy(), warm.copy()
        with BatchSolver(params) as s:
            s.set_costmap(*cmap)
            cg, xg = s.solve(probs, st, warm)
        cc, xc, _ = c_oracle.solve_batch(params, cmap, probs, st_c, warm_c)
        dx = np.abs(xg - xc).max(axis=1)
        j = int(np.argmax(dx))
        print("n %2d cap %3d  max|dx| %.3e (instance %d)  frac>1e-6 %.3f  it gpu %.2f cpu %.2f  cost gpu-cpu max %.2e min %.2e"
              % (n, cap, dx.max(), j, (dx > 1e-6).mean(), cg["iterations"].mean(), cc["iterations"].mean(),
                 (cg["cost"] - cc["cost"]).max(), (cg["cost"] - cc["cost"]).min()), flush=True)
        if args.show and dx.max() > 1e-6:
            print("   gpu", np.array2string(xg[j], precision=5, max_line_width=250))
            print("   cpu", np.array2string(xc[j], precision=5, max_line_width=250))
            print("   v_cur", probs["cur_vel"][j], "flags gpu/cpu", cg["flags"][j], cc["flags"][j])
