#!/usr/bin/env python3
"""Development aid (GPU box): K1's search direction in one iteration against the CPU mirror's, per instance,
on one of the fuzz configurations (tests/test_gpu_fuzz.py) or a synthetic workload."""
import argparse, ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neo_mpc_planner2_amd import synthetic
from neo_mpc_planner2_amd.solver import BatchSolver
from oracle import c_oracle, mpc_oracle as orc
from tests.test_gpu_fuzz import random_params

ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--steps", type=int, default=0, help="0: the fuzz configuration of --seed; else README params at this control_steps")
ap.add_argument("--iteration", type=int, default=1)
ap.add_argument("--count", type=int, default=64)
ap.add_argument("--from-iterate", type=int, default=0,
                help="k > 0: start both sides from the mirror's iterate after k iterations (a warm start: the direction of "
                     "iteration 0 is then computed at exactly the same point on both sides)")
args = ap.parse_args()
rng = np.random.default_rng(1000 + args.seed)
if args.steps:
    params = orc.make_params(control_steps=args.steps)
    res = 0.05
else:
    params = random_params(rng)
    res = float(rng.choice([0.025, 0.05, 0.1]))
n = params["control_steps"]
cmap = synthetic.make_costmap(240, seed=args.seed, resolution=res)
probs = synthetic.make_problems(args.count, 240, seed=args.seed + 50, resolution=res)
probs["cur_vel"] *= params["max_vel_trans"]
warm = np.zeros((args.count, 3 * n))
if args.from_iterate:
    st, w = synthetic.make_states(probs, n)
    _, warm, _ = c_oracle.solve_batch(dict(params, max_iterations=args.from_iterate), cmap, probs, st, w)
    warm = np.ascontiguousarray(warm)
with BatchSolver(params) as s:
    s.set_costmap(*cmap)
    dg = s.direction(probs, warm, args.iteration)
lib = c_oracle.load()
dc = np.full_like(dg, np.nan)
for j in range(args.count):
    buf = np.full(3 * n, np.nan)
    lib.orc_capture_direction(args.iteration, C.c_void_p(buf.ctypes.data))
    st, w = synthetic.make_states(probs[j:j + 1], n)
    w[:] = warm[j:j + 1]
    c_oracle.solve_batch(params, cmap, probs[j:j + 1], st, w)
    lib.orc_capture_direction(-1, None)
    dc[j] = buf
both = ~np.isnan(dg).any(axis=1) & ~np.isnan(dc).any(axis=1)
print("control_steps", n, "box", params["max_vel_x"], params["min_vel_x"], params["max_vel_y"], params["min_vel_y"], "r", params["max_vel_trans"],
      "w", params["max_vel_theta"], params["min_vel_theta"])
print("instances with a direction on both sides:", both.sum(), "gpu only", (~np.isnan(dg).any(axis=1) & ~both).sum(),
      "cpu only", (~np.isnan(dc).any(axis=1) & ~both).sum())
err = np.abs(dg - dc).max(axis=1) / np.maximum(np.abs(dc).max(axis=1), 1e-12)
err[~both] = 0
order = np.argsort(-err)
print("relative direction error: median %.2e p90 %.2e max %.2e" % (np.median(err[both]), np.percentile(err[both], 90), err.max()))
for j in order[:3]:
    print("instance", j, "rel err %.3e" % err[j])
    print("  gpu", np.array2string(dg[j], precision=4, max_line_width=200))
    print("  cpu", np.array2string(dc[j], precision=4, max_line_width=200))
    print("  u  ", np.array2string(warm[j], precision=5, max_line_width=200), "v_cur", probs["cur_vel"][j])
