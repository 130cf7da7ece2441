#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point `neo_mpc_solve_batch` (staging through
device memory each call) on BASELINE config 2; never the headline `value`."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from neo_mpc_planner2_amd import synthetic  # noqa: E402
from neo_mpc_planner2_amd.mpc_optimization_server import README_PARAMS  # noqa: E402
from neo_mpc_planner2_amd.solver import BatchSolver  # noqa: E402

cfg, cmap, probs, st, warm = synthetic.make_workload("C2", seed=0)
with BatchSolver(README_PARAMS) as s:
    s.set_costmap(*cmap)
    for _ in range(5):
        s.solve(probs, st.copy(), warm.copy())
    sets = [(st.copy(), warm.copy()) for _ in range(100)]
    t0 = time.perf_counter()
    for a, b in sets:
        s.solve(probs, a, b)
    dt = (time.perf_counter() - t0) / len(sets)
    one = synthetic.make_problems(1, 500, seed=5)
    s1, w1 = synthetic.make_states(one, 3)
    for _ in range(20):
        s.solve(one, s1.copy(), w1.copy())
    t0 = time.perf_counter()
    for _ in range(200):
        s.solve(one, s1.copy(), w1.copy())
    lat = (time.perf_counter() - t0) / 200
print(json.dumps({"entry_point": "neo_mpc_solve_batch (host buffers, H2D + K1 + D2H per call)",
                  "batch": 4096, "ms_per_call": dt * 1e3, "solves_per_s": 4096 / dt,
                  "single_robot_call_ms": lat * 1e3,
                  "note": "single_robot_call_ms is the plugin's per-tick latency (count = 1) incl. ctypes"}))
