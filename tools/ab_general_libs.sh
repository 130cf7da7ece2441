#!/bin/bash
# same-box A/B of builds of libneo_mpc on the parameter sets that take the GENERAL kernels (bench.GENERAL_SETS), 4096 and
# 65 536 instances, default occupancy   usage: bash tools/ab_general_libs.sh lib1.so lib2.so ...
for r in 1 2; do
for lib in "$@"; do
NEO_MPC_LIB=$lib python - <<PY 2>/dev/null
import json, sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
import neo_mpc_planner2_amd.synthetic as syn
for batch in (4096, 65536):
    for label in sorted(bench.GENERAL_SETS):
        syn.CONFIGS["C2"]["batch"] = batch
        r = bench.other_workload("C2", "cuda:0", 0, steps=12 if batch == 4096 else 4, warmup=2, params_over=bench.GENERAL_SETS[label], label=label)
        print("%-24s %-8s batch %6d  %.2f Msolves/s  kernel %.4f ms  it %.2f max %d" % (os.path.basename("$lib"), label, batch, r["value"] / 1e6, r["kernel_ms"], r["solver"]["mean_iterations"], r["solver"]["max_iterations_seen"]))
PY
done
done
