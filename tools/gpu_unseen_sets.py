#!/usr/bin/env python3
"""The HIP build (not the CPU mirror) on random parameter sets that are in NO committed fixture.

oracle/fuzz_reference.py reports the CPU mirror against the reference on seeds drawn for the purpose; the fixtures G14-G16 and
G18 hold 266 of those sets and the GPU tests gate them.  This tool runs the product itself over the REST of a fuzz run:

  here (the reference's answers come from NEO_FUZZ_CACHE, or are made again from /root/reference -- slow):
      NEO_FUZZ_CACHE=/tmp/fuzz_cache python tools/gpu_unseen_sets.py --pack scratch/unseen_sets.npz 30000-30199 70048-70199
  on the GPU box (scratch/ travels with the snapshot, it is not committed):
      python tools/gpu_unseen_sets.py --run scratch/unseen_sets.npz > gpurun_out/unseen.txt

The protocol is tests/util.random_sets_miss_rates (G14's): P3 against SLSQP as shipped on 12 all-free-map and 12 costmap cold
problems per set, P2 against SLSQP run to the end on the cases the reference's own answers flag unique.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def seeds_of(specs, skip):
    out = []
    for s in specs:
        lo, _, hi = s.partition("-")
        out += [v for v in range(int(lo), int(hi or lo) + 1) if v not in skip]
    return out


def pack(path, specs):
    from oracle import gen_golden
    committed = set(gen_golden.G16_SEEDS) | set(gen_golden.G18_SEEDS) | set(gen_golden.G15_SEEDS) | set(range(gen_golden.G14_SEEDS))
    seeds = seeds_of(specs, committed)
    keep = gen_golden.OUT
    gen_golden.OUT = os.path.dirname(os.path.abspath(path))
    try:
        gen_golden._random_sets(seeds, os.path.basename(path), "unseen sets")
    finally:
        gen_golden.OUT = keep


def run(path):
    import util
    from neo_mpc_planner2_amd import solver as solver_mod, synthetic

    def solve(params, cmap, pr):
        st, warm = synthetic.make_states(pr, params["control_steps"])
        with solver_mod.BatchSolver(params) as s:
            s.set_costmap(*cmap)
            return s.solve(pr, st, warm)

    path = os.path.abspath(path)
    with np.load(path) as z:
        seeds, steps = z["seeds"], z["steps"]
    print("# tools/gpu_unseen_sets.py --run: the HIP build through the C-ABI on %d random parameter sets in no committed "
          "fixture (seeds %d..%d), G14's protocol" % (len(seeds), seeds.min(), seeds.max()))
    rows = [("all", None)] + [("control_steps %d" % n, n) for n in sorted(set(int(v) for v in steps))]
    for label, n in rows:
        m = util.random_sets_miss_rates(solve, path, only_steps=n)
        sets = int((steps == n).sum()) if n is not None else len(steps)
        print("%-17s %3d sets: P3 misses %d / %d all-free, %d / %d costmap (worst f - f_ref %+.2e); reference more than 1e-3 "
              "above the build on %d; P2 misses %d / %d unique (worst %.2e); %s"
              % (label, sets, m["p3_miss_free"], m["cases_free"], m["p3_miss_map"], m["cases_map"], m["p3_worst"],
                 m["ref_worse"], m["p2_miss"], m["p2_cases"], m["p2_worst"], m["misses"] or "no miss"))


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "--pack":
        pack(sys.argv[2], sys.argv[3:])
    elif len(sys.argv) == 3 and sys.argv[1] == "--run":
        run(sys.argv[2])
    else:
        sys.exit(__doc__)
