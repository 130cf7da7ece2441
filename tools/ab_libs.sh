#!/bin/bash
# same-box A/B of two builds of libneo_mpc (box-to-box variance is ~5 %, so only same-call numbers compare)
# usage: bash tools/ab_libs.sh <libA.so> <libB.so> [rounds]
A=$1; B=$2; R=${3:-2}
run() {  # lib, label, bench args...
  local lib=$1 label=$2; shift 2
  NEO_MPC_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-pcie --no-others "$@" 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label %-28s %.4g solves/s kernel_ms %.4f it %.2f' % ('$(basename $lib)', d['value'], d['roofline']['kernel_ms'], d['solver']['mean_iterations']))"
}
for r in $(seq $R); do
  for lib in $A $B; do run $lib C2 --steps 300; done
done
for lib in $A $B; do run $lib C2x64 --batch 262144 --steps 30 --warmup 3; done
for lib in $A $B; do run $lib C3 --workload C3 --steps 5 --warmup 1; done
for lib in $A $B; do run $lib C5 --workload C5 --steps 3 --warmup 1; done
