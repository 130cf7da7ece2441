#!/bin/bash
# same-box A/B of two builds on the run-time-sized (Riccati) kernel: C3, C5, and control_steps 16
A=$1; B=$2
run() { local lib=$1 label=$2; shift 2
  NEO_MPC_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-pcie --no-others "$@" 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label %-24s %.4g solves/s kernel_ms %.4f it %.3f max %d' % ('$(basename $lib)', d['value'], d['roofline']['kernel_ms'], d['solver']['mean_iterations'], d['solver']['max_iterations_seen']))"
}
for r in 1 2; do
for lib in $A $B; do run $lib C3 --workload C3 --steps 5 --warmup 1; done
for lib in $A $B; do run $lib C5 --workload C5 --steps 4 --warmup 1; done
done
for lib in $A $B; do run $lib C5n16 --workload C5 --control-steps 16 --steps 4 --warmup 1; done
