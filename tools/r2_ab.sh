#!/bin/bash
# A/B of two builds of the library on one box: C2 / C3 / C5 lines, each build twice (interleaved)
mkdir -p gpurun_out
A=neo_mpc_planner2_amd/libneo_mpc.so; B=$1
for rep in 1 2; do for lib in $A $B; do
  for w in C2 C3 C5; do
    steps=3; [ $w = C2 ] && steps=300
    NEO_MPC_LIB=$lib timeout 300 python bench.py --workload $w --steps $steps --warmup 2 --no-cpu-baseline --no-pcie 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', '$w', '%.4g solves/s kernel %.4f ms' % (d['value'], d['roofline']['kernel_ms']))"
  done; done; done 2>&1 | tee gpurun_out/r2_ab.log
