#!/bin/bash
# one gpurun call while developing (round 6): the GPU tests (optionally a -k expression), the default bench line, the same
# line with the dense direction for every instance (method 2 = round 5's AUTO: the A/B partner of the routed kernel), a
# warm-tick loop.   usage: bash tools/r6_check.sh <tag> ["pytest -k expression"]
TAG=${1:-r06a}; K=${2:-}
O=gpurun_out/check_$TAG
mkdir -p $O
if [ -n "$K" ]; then timeout 2400 python -m pytest tests -m gpu -q -k "$K" 2>&1 | tail -40 > $O/gpu_tests.log
else timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/gpu_tests.log; fi
timeout 900 python bench.py > $O/bench_c2.log 2>&1; tail -1 $O/bench_c2.log > $O/bench_c2.json
for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-pcie --no-others 2>/dev/null | tail -1 > $O/ab_routed_$rep.json
  timeout 300 python bench.py --no-cpu-baseline --no-pcie --no-others --method 2 2>/dev/null | tail -1 > $O/ab_dense_$rep.json
done
timeout 300 python tools/bench_fleet_loop.py 2>/dev/null | tail -1 > $O/fleet_loop.json
cat $O/gpu_tests.log; cut -c1-2500 $O/bench_c2.json; echo
for f in $O/ab_*.json; do echo $f; python - $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print({k:d.get(k) for k in ("value","ms_per_step")}, d.get("roofline",{}).get("kernel_ms"), d.get("solver"), d.get("warm_tick"))
except Exception as e: print("unreadable", e)
PY
done
cut -c1-600 $O/fleet_loop.json
