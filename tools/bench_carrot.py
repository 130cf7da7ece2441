#!/usr/bin/env python3
"""HBM-roofline measurement of K4 (carrot selection): the plans are streamed once for the
closest-pose search (24 B per pose) -- the one kernel of the path that IS bandwidth-bound.

    python tools/bench_carrot.py [--robots 65536] [--poses 512] [--steps 20]
Prints one JSON line: achieved GB/s = (24 B x total poses + per-robot records) / kernel time.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--robots", type=int, default=65536)
    ap.add_argument("--poses", type=int, default=512)
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    import torch
    from neo_mpc_planner2_amd import abi
    from neo_mpc_planner2_amd.solver import BatchSolver
    dev = "cuda:0"
    B, P = args.robots, args.poses
    g = torch.Generator(device=dev).manual_seed(0)
    # synthetic plans generated on the device: smooth polylines, 0.05 m spacing
    heading = torch.cumsum(torch.randn((B, P), device=dev, dtype=torch.float64, generator=g) * 0.03, dim=1)
    heading += (torch.rand((B, 1), device=dev, dtype=torch.float64, generator=g) * 2 - 1) * np.pi
    xy = torch.cumsum(torch.stack([torch.cos(heading), torch.sin(heading)], dim=2) * 0.05, dim=1)
    poses = torch.cat([xy, heading[..., None]], dim=2).contiguous()            # [B, P, 3]
    k = torch.randint(0, P, (B,), device=dev, generator=g)
    robots = poses[torch.arange(B, device=dev), k].clone()
    offsets = (torch.arange(B + 1, device=dev, dtype=torch.int64) * P).to(torch.int32)
    slow = torch.ones(B, dtype=torch.int32, device=dev)
    fcost = torch.zeros(B, dtype=torch.float64, device=dev)
    carrots = torch.zeros((B, abi.CARROT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    lp = abi.NeoMpcLookaheadParams(0.4, 0.4, 0.4, 5.0)
    s = BatchSolver({})
    stream = torch.cuda.current_stream()
    for _ in range(3):
        s.select_carrots_device(lp, poses, offsets, robots, slow, carrots, fcost)
    torch.cuda.synchronize()
    ms = []
    for _ in range(args.steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        s.select_carrots_device(lp, poses, offsets, robots, slow, carrots, fcost)
        b.record(stream)
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    k_ms = float(np.median(ms))
    algo = 24.0 * B * P + B * (24 + 8 + 4 + 4 + 80 + 4)
    out = {"kernel": "k_carrot", "robots": B, "poses_per_plan": P, "kernel_ms": k_ms,
           "algorithmic_bytes": algo, "achieved_GBps": algo / (k_ms * 1e-3) / 1e9, "peak_GBps": 8000.0,
           "frac": algo / (k_ms * 1e-3) / 1e9 / 8000.0,
           "carrots_per_s": B / (k_ms * 1e-3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
