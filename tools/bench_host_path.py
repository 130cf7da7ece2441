#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-array entry point `neo_mpc_solve_batch` on the C2 problem at 4096 ... 262 144 instances
per call: pageable arrays (staged by the runtime), page-locked arrays three ways (worked on in place / DMA in + in-place
results / staged by DMA; neo_mpc_set_host_path), and the count = 1 call of the plugin.  Never the headline `value`."""
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


def rate(fn, count, seconds=1.5, prep=None):
    """`prep` (untimed) puts the call's in/out arrays back to the cold-start values before every timed call."""
    if prep:
        prep()
    fn()
    reps, t_all = 0, 0.0
    while t_all < seconds and reps < 200:
        if prep:
            prep()
        t0 = time.perf_counter()
        fn()
        t_all += time.perf_counter() - t0
        reps += 1
    return {"Msolves_per_s": count * reps / t_all / 1e6, "ms_per_call": 1e3 * t_all / reps}


cmap = synthetic.make_costmap(500, seed=0)
out = {"entry_point": "neo_mpc_solve_batch (host arrays), C2 problem, cold start", "batches": {}}
with BatchSolver(README_PARAMS) as s:
    s.set_costmap(*cmap)
    for count in [int(x) for x in os.environ.get("NEO_MPC_HOST_PATH_COUNTS", "4096,32768,65536,262144").split(",")]:
        probs = synthetic.make_problems(count, 500, seed=1000)
        st, warm = synthetic.make_states(probs, 3)
        res = {}

        g_st, g_warm = st.copy(), warm.copy()

        def reset_pageable():
            g_st[...] = st
            g_warm[...] = warm
        res["pageable"] = rate(lambda: s.solve(probs, g_st, g_warm), count, prep=reset_pageable)
        if count >= 65536:   # (staged batches this large go through in four pieces on two streams: the one-piece path beside it)
            os.environ["NEO_MPC_NO_CHUNKS"] = "1"     # (read once, by neo_mpc_create: a handle of its own)
            with BatchSolver(README_PARAMS) as s1:
                s1.set_costmap(*cmap)
                res["pageable_one_piece"] = rate(lambda: s1.solve(probs, g_st, g_warm), count, prep=reset_pageable)
            del os.environ["NEO_MPC_NO_CHUNKS"]
        p_probs, p_st, p_warm = pinned(np.ascontiguousarray(probs)), pinned(st), pinned(warm)
        p_cmd, p_sol = pinned(np.zeros(count, dtype=abi.COMMAND_DTYPE)), pinned(np.zeros((count, 9)))

        def reset_locked():
            p_st[...] = st
            p_warm[...] = warm
        for mode in ("zerocopy", "zerocopy_out", "staged"):
            s.set_host_path(mode)
            res["page_locked_" + mode] = rate(lambda: s.solve(p_probs, p_st, p_warm, out=(p_cmd, p_sol)), count,
                                              prep=reset_locked)
        s.set_host_path("auto")
        out["batches"][count] = res
    one = synthetic.make_problems(1, 500, seed=5)
    s1, w1 = synthetic.make_states(one, 3)
    out["single_robot_call_ms"] = rate(lambda: s.solve(one, s1.copy(), w1.copy()), 1, 0.5)["ms_per_call"]
out["note"] = ("each timed call is BatchSolver.solve on cold-start arrays (reset untimed), ctypes marshalling included; "
               "single_robot_call_ms is the plugin's per-tick latency (count = 1)")
print(json.dumps(out))
