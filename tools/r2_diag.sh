#!/bin/bash
mkdir -p gpurun_out
{
for wl in C3 C5; do timeout 300 python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline --no-pcie 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:3], '%.4g solves/s kernel %.3f ms' % (d['value'], d['roofline']['kernel_ms']))"; done
for n in 12 16 20 24; do timeout 300 python bench.py --workload C5 --control-steps $n --steps 3 --warmup 1 --no-cpu-baseline --no-pcie 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('N=$n', '%.4g solves/s kernel %.3f ms its %.2f' % (d['value'], d['roofline']['kernel_ms'], d['solver']['mean_iterations']))"; done
} > gpurun_out/r2_waves2.log 2>&1
cat gpurun_out/r2_waves2.log
