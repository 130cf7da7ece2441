#!/bin/bash
mkdir -p gpurun_out
{
for w in 3 4; do NEO_MPC_SOLVE_WAVES=$w timeout 300 python bench.py --workload C5 --steps 4 --warmup 1 --no-cpu-baseline --no-pcie 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('waves $w', d['config']['workload'][:3], '%.4g solves/s kernel %.3f ms' % (d['value'], d['roofline']['kernel_ms']))"; done
} > gpurun_out/r2_waves3.log 2>&1
cat gpurun_out/r2_waves3.log
