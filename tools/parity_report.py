#!/usr/bin/env python3
"""Parity report of the HIP solver against the reference's SciPy path (SURVEY.md §8c tiers
P2-P4), printed as text.  Runs on the GPU box; uses tests/golden/g3_solves.npz (SLSQP solves of
the REFERENCE's objective at ftol 1e-3 and 1e-12, cold start).

    python tools/parity_report.py > profiles/rNN_parity_report.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from neo_mpc_planner2_amd import synthetic  # noqa: E402
from neo_mpc_planner2_amd.solver import BatchSolver  # noqa: E402
from tests import util  # noqa: E402


def pct(a):
    return "median %.2e  p90 %.2e  p99 %.2e  max %.2e" % (np.median(a), np.percentile(a, 90), np.percentile(a, 99), a.max())


def main():
    g = util.load("g3_solves.npz")
    params = util.params_from(g["param_keys"], g["params"])
    probs = util.problems_from(g["problems"])
    hm = g["has_map"].astype(bool)
    print("# HIP solver vs SciPy SLSQP on the reference objective (%s)" % str(g["versions"]))
    for name, mask, cells in (("zero costmap (unique minimiser)", ~hm, np.zeros_like(g["cells"])),
                              ("synthetic costmap (local minima)", hm, g["cells"])):
        cmap = (cells,) + tuple(g["map_meta"])
        pr = probs[mask]
        st, warm = synthetic.make_states(pr, 3)
        with BatchSolver(params) as s:
            s.set_costmap(*cmap)
            cmds, x = s.solve(pr, st, warm)
        du_t = np.abs(x[:, :3] - g["x_tight"][mask][:, :3]).max(axis=1)
        du_l = np.abs(x[:, :3] - g["x_loose"][mask][:, :3]).max(axis=1)
        ref_ll = np.abs(g["x_loose"][mask][:, :3] - g["x_tight"][mask][:, :3]).max(axis=1)
        print("\n## %s, %d cold-start problems" % (name, mask.sum()))
        print("P2  |u0 - u0(SLSQP ftol=1e-12)|_inf : %s" % pct(du_t))
        print("P3  f - f(SLSQP ftol=1e-3)          : max %.3e  (bar: <= 1e-3)   f - f(SLSQP 1e-12): max %.3e min %.3e"
              % ((cmds["cost"] - g["f_loose"][mask]).max(), (cmds["cost"] - g["f_tight"][mask]).max(),
                 (cmds["cost"] - g["f_tight"][mask]).min()))
        print("P4  |u0 - u0(SLSQP ftol=1e-3)|_inf  : %s" % pct(du_l))
        print("    reference vs itself, |u0(SLSQP 1e-3) - u0(SLSQP 1e-12)|_inf : %s" % pct(ref_ll))
        print("    iterations: mean %.1f max %d; converged %d/%d" % (cmds["iterations"].mean(), cmds["iterations"].max(),
                                                                   (cmds["status"] == 0).sum(), len(cmds)))


if __name__ == "__main__":
    main()
