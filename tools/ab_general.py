#!/usr/bin/env python3
"""Same-box A/B of the occupancy variants of the GENERAL (non-"tame") kernels on the parameter sets that take them
(bench.GENERAL_SETS: "cut" = dense Newton, "turn" = stage-wise direction), 4096 and 65 536 instances."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import json, sys, os
sys.path.insert(0, %r)
import torch, bench
for batch in (4096, 65536):
    for label in sorted(bench.GENERAL_SETS):
        import neo_mpc_planner2_amd.synthetic as syn
        syn.CONFIGS["C2"]["batch"] = batch
        r = bench.other_workload("C2", "cuda:0", 0, steps=5, warmup=2, params_over=bench.GENERAL_SETS[label], label=label)
        print(json.dumps({"waves": os.environ.get("NEO_MPC_SOLVE_WAVES", "default"), "set": label, "batch": batch,
                          "Msolves": r["value"] / 1e6, "kernel_ms": r["kernel_ms"], "it": r["solver"]["mean_iterations"]}))
''' % ROOT
for waves in ("default", "3", "4"):
    env = dict(os.environ)
    if waves != "default":
        env["NEO_MPC_SOLVE_WAVES"] = waves
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("\n".join(l for l in out.stdout.splitlines() if l.startswith("{")) or out.stderr[-400:])
