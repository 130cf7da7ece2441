#!/usr/bin/env python3
"""Study (round-4 review, item 7): does ONE 4096-instance host call get faster when it goes to the device as two or four
pieces on streams of their own (neo_mpc_solve_batch_begin / _wait on slices of the same page-locked arrays) -- the later
pieces' records crossing PCIe under the first piece's arithmetic?  Prints one JSON line; never the headline `value`."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from neo_mpc_planner2_amd import abi, synthetic  # noqa: E402
from neo_mpc_planner2_amd.mpc_optimization_server import README_PARAMS  # noqa: E402
from neo_mpc_planner2_amd.solver import BatchSolver  # noqa: E402


def pinned(a):
    t = torch.empty(a.nbytes, dtype=torch.uint8).pin_memory()
    v = t.numpy().view(a.dtype).reshape(a.shape)
    v[...] = a
    return v


def main():
    count = 4096
    cmap = synthetic.make_costmap(500, seed=0)
    probs = synthetic.make_problems(count, 500, seed=1000)
    st, warm = synthetic.make_states(probs, 3)
    p_probs, p_st, p_warm = pinned(np.ascontiguousarray(probs)), pinned(st), pinned(warm)
    p_cmd, p_sol = pinned(np.zeros(count, dtype=abi.COMMAND_DTYPE)), pinned(np.zeros((count, 9)))
    out = {"what": "one 4096-instance call on page-locked arrays, in place: whole, and as 2 / 4 slices in flight at once", "pieces": {}}
    with BatchSolver(README_PARAMS) as s:
        s.set_costmap(*cmap)
        ref = None
        for pieces in (1, 2, 4):
            edges = np.linspace(0, count, pieces + 1).astype(int)
            sl = [slice(a, b) for a, b in zip(edges[:-1], edges[1:])]

            def call():
                if pieces == 1:
                    s.solve(p_probs, p_st, p_warm, out=(p_cmd, p_sol))
                    return
                tickets = [s.solve_begin(p_probs[q], p_st[q], p_warm[q], (p_cmd[q], p_sol[q])) for q in sl]
                for t in tickets:
                    s.solve_wait(t)
            ms = []
            for rep in range(60):
                p_st[...] = st
                p_warm[...] = warm
                t0 = time.perf_counter()
                call()
                ms.append(1e3 * (time.perf_counter() - t0))
            if ref is None:
                ref = p_cmd.copy()
            assert p_cmd.tobytes() == ref.tobytes()      # (slices are the same instances: bit-identical results)
            ms = np.array(ms[10:])
            out["pieces"][pieces] = {"ms_per_call_median": float(np.median(ms)), "Msolves_per_s": count / np.median(ms) / 1e3}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
