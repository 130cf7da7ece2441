#!/usr/bin/env python3
"""K3 (costmap ingest) on its own: raw nav2 cells in HBM -> bordered, pitched device maps.  HBM-streaming:
bytes = raw cells read + padded map written.  usage: bench_ingest.py [maps] [size]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from neo_mpc_planner2_amd.mpc_optimization_server import README_PARAMS  # noqa: E402
from neo_mpc_planner2_amd.solver import BatchSolver  # noqa: E402

maps = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
size = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = "cuda:0"
params = dict(README_PARAMS)
params.update(control_steps=3)
with BatchSolver(params) as s:
    cells = torch.randint(0, 255, (maps, size, size), dtype=torch.uint8, device=dev)
    orig = torch.zeros((maps, 2), dtype=torch.float64, device=dev)
    border = 16 if maps > 1 else 64
    pitch = (size + 2 * border + 127) // 128 * 128
    written = maps * pitch * (size + 2 * border)
    # 20 back-to-back launches per event pair: an event pair around a single 30-us launch measures the event records
    # (two barrier packets) as much as the kernel; rocprofv3's per-kernel time (profiles/r03_*_k3_ingest.txt) agrees
    # with this figure, not with the single-launch one
    def launch():
        if maps > 1:
            s.set_costmap_pool(cells, 0.05, orig)
        else:
            s.set_costmap(cells[0], 0.05, 0.0, 0.0)
    for _ in range(5):
        launch()
    reps, per = 8, 20
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in evs:
        e0.record()
        for _ in range(per):
            launch()
        e1.record()
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in evs])) / per
print(json.dumps({"kernel": "k_ingest", "maps": maps, "size": size, "ms": ms, "read_bytes": maps * size * size,
                  "written_bytes": written, "achieved_GBps": (maps * size * size + written) / (ms * 1e-3) / 1e9,
                  "peak_GBps": 8000.0, "frac": (maps * size * size + written) / (ms * 1e-3) / 1e9 / 8000.0}))
