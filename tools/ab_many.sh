#!/bin/bash
# same-box A/B of several builds of libneo_mpc on chosen workloads   usage: bash tools/ab_many.sh "C2 C3 C5" lib1.so lib2.so ...
WL=$1; shift
run() {  # lib, label, bench args...
  local lib=$1 label=$2; shift 2
  NEO_MPC_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-pcie --no-others "$@" 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label %-28s %.4g solves/s kernel_ms %.4f it %.2f max %d' % ('$(basename $lib)', d['value'], d['roofline']['kernel_ms'], d['solver']['mean_iterations'], d['solver']['max_iterations_seen']))"
}
for r in 1 2; do
  for w in $WL; do
    for lib in "$@"; do
      case $w in
        C2) run $lib C2 --steps 300 ;;
        C2x64) run $lib C2x64 --batch 262144 --steps 30 --warmup 3 ;;
        C3) run $lib C3 --workload C3 --steps 5 --warmup 1 ;;
        C5) run $lib C5 --workload C5 --steps 3 --warmup 1 ;;
      esac
    done
  done
done
