#!/usr/bin/env python3
"""The deployed (warm-started) mode against the REFERENCE ITSELF at random parameter sets (development container only:
imports /root/reference under the ROS stand-ins, like gen_golden.py).  For every seed a parameter set is drawn
(fuzz_reference.draw), the reference runs G11's protocol on it -- optimizer() episodes RUN TO CONVERGENCE (`opt_tolerance`
1e-12, SLSQP's cap raised to 500 inside the call of py:363-364) on an all-free map, robots moved by the reference's own
commands, carrot re-drawn every 10 calls, new goal mid-episode, every tick solved three times for the `settled` flag -- and
the CPU mirror of the build's search, at the set's own shipped tolerance, solves every call from the reference's own state
(tests/util.warm_gate).  Reported per set and in total: settled ticks, how many of them have the build's command more than
1e-3 from the reference's converged command, the largest difference.
COSTMAP MODE (round 6; `--costmap` in front of the seeds): P3w's protocol at random parameter sets -- the reference AS SHIPPED
(the set's own opt_tolerance) on the 200 x 200 costmap of G4, 4 episodes x 30 calls per set (gen_golden.warm_costmap_group:
what G17 holds for its first 16 seeds), every call solved by the mirror from the reference's own state: f(build) <=
f(reference's raw x.x) + 1e-3 (tests/util.p3w_group).
Test infrastructure: nothing here is shipped.   usage: fuzz_reference_warm.py [--costmap] <first seed> <last seed + 1> [episodes] [calls]"""
import contextlib
import io
import multiprocessing as mp
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def work(args):
    seed, n_ep, n_calls = args
    tmp = tempfile.mkdtemp(prefix="neo_warm_fuzz_")
    os.environ["NEO_MPC_GOLDEN_OUT"] = tmp
    from oracle import fuzz_reference, gen_golden, ros_stubs, c_oracle
    import util
    gen_golden.OUT = tmp
    n, over = fuzz_reference.draw(seed)
    tol = over["opt_tolerance"]
    mod = ros_stubs.load_reference()
    name = "warm_seed%d.npz" % seed
    with contextlib.redirect_stdout(io.StringIO()):
        gen_golden.gen_g4(mod, n_steps=n, n_ep=n_ep, n_calls=n_calls, fname=name, overrides=dict(over, opt_tolerance=1e-12),
                          free_map=True, maxiter=500, seed_base=50000 + 10 * seed, settle_check=True)
    util.GOLDEN = tmp
    g = util.load(name)
    call = [0]
    df = []

    def solve(params, cmap, rows, st, wm):
        p = dict(params, opt_tolerance=tol)
        cm, x, _ = c_oracle.solve_batch(p, cmap, rows, st, wm)
        # (the build's objective against the objective of the reference's converged x.x of the same call)
        df.append(cm["cost"] - c_oracle.objective_batch(p, cmap, rows, g["raw_x"][:, call[0]]))
        call[0] += 1
        return cm, x

    def post(params, cmap, rows, st, wm, x, ok):
        c_oracle.postprocess_batch(params, cmap, rows, st, wm, x, ok)
    c_oracle.set_threads(1)
    dv, du, its, settled = util.warm_gate(solve, post, name)
    df = np.array(df).T[g["success"].astype(bool)]
    os.remove(os.path.join(tmp, name))
    bad = settled & (dv > 1e-3)
    return seed, n, tol, int(dv.size), int(settled.sum()), int(bad.sum()), int((dv > 1e-3).sum()), \
        float(dv[settled].max()) if settled.any() else 0.0, float(dv.max()), float(its.mean()), \
        int((bad & (df > 1e-3)).sum()), int((bad & (df < -1e-3)).sum()), float(df[bad].max()) if bad.any() else 0.0


def work_costmap(args):
    seed, n_ep, n_calls = args
    from oracle import gen_golden, c_oracle
    import util
    seed, n, grp = gen_golden.warm_costmap_group(seed, n_ep, n_calls)
    params = util.params_from(np.array(gen_golden.PARAM_KEYS), grp["params"])
    cmap = (grp["cells"],) + tuple(grp["map_meta"])
    c_oracle.set_threads(1)

    def solve(p, cm, rows, st, wm):
        c, x, _ = c_oracle.solve_batch(p, cm, rows, st, wm)
        return c, x

    def post(p, cm, rows, st, wm, x, ok):
        c_oracle.postprocess_batch(p, cm, rows, st, wm, x, ok)
    worse, its, capped = util.p3w_group(solve, post, params, cmap, grp)
    return seed, n, params["opt_tolerance"], float(worse.max()), int((worse > 1e-3).sum()), int(worse.size), float(np.median(worse)), \
        float(its.mean()), int(its.max()), capped


def main_costmap(argv):
    seeds = list(range(int(argv[0]), int(argv[1])))
    n_ep = int(argv[2]) if len(argv) > 2 else 4
    n_calls = int(argv[3]) if len(argv) > 3 else 30
    with mp.Pool(max(1, (os.cpu_count() or 2) - 1)) as pool:
        rows = pool.map(work_costmap, [(s, n_ep, n_calls) for s in seeds], chunksize=1)
    miss = calls = caps = 0
    for seed, n, tol, mx, above, cnt, med, it, itmax, capped in rows:
        miss += above
        calls += cnt
        caps += capped
        print("seed %d control_steps %2d opt_tolerance %.0e: %3d calls, f(build) - f(reference's x.x): max %+.2e, median %+.2e, "
              "more than 1e-3 above on %d; iterations %.2f (max %d)%s"
              % (seed, n, tol, cnt, mx, med, above, it, itmax, "; %d searches ran into the iteration cap" % capped if capped else ""))
    print("%d parameter sets on the costmap, the reference as shipped, every call from the reference's own state: the build's "
          "objective more than 1e-3 above the reference's on %d of %d calls (largest f - f_ref %+.2e); %d searches ran into the "
          "iteration cap" % (len(rows), miss, calls, max(r[3] for r in rows), caps))


if __name__ == "__main__":
    if sys.argv[1] == "--costmap":
        main_costmap(sys.argv[2:])
        sys.exit(0)
    seeds = list(range(int(sys.argv[1]), int(sys.argv[2])))
    n_ep = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    n_calls = int(sys.argv[4]) if len(sys.argv) > 4 else 30
    with mp.Pool(max(1, (os.cpu_count() or 2) - 1)) as pool:
        rows = pool.map(work, [(s, n_ep, n_calls) for s in seeds], chunksize=1)
    tot = np.zeros(6, dtype=int)
    for seed, n, tol, ticks, ns, bad_s, bad, mx_s, mx, it, worse, better, dfmax in rows:
        tot += (ticks, ns, bad_s, bad, worse, better)
        print("seed %d control_steps %2d opt_tolerance %.0e: %4d converged ticks, %4d settled; command more than 1e-3 from the "
              "reference's: %d settled (%d of all); max %.2e settled (%.2e all); of those the build's objective is more than 1e-3 "
              "above / below the reference's on %d / %d (largest f - f_ref %.2e); iterations %.2f"
              % (seed, n, tol, ticks, ns, bad_s, bad, mx_s, mx, worse, better, dfmax, it))
    print("%d parameter sets: %d converged ticks, %d settled; command more than 1e-3 from the reference's converged command on %d "
          "settled ticks (%.3f %%), on %d of all; on %d of those settled ticks the build's objective is more than 1e-3 above the "
          "reference's, on %d more than 1e-3 below (the others: another KKT point of the same value)"
          % (len(rows), tot[0], tot[1], tot[2], 100.0 * tot[2] / max(1, tot[1]), tot[3], tot[4], tot[5]))
