#!/bin/bash
# one gpurun call while developing: the GPU tests (optionally a -k expression), the default bench line, a warm-tick loop
# usage: bash tools/r5_check.sh <tag> ["pytest -k expression"]
TAG=${1:-r05a}; K=${2:-}
O=gpurun_out/check_$TAG
mkdir -p $O
if [ -n "$K" ]; then timeout 2400 python -m pytest tests -m gpu -q -x -k "$K" 2>&1 | tail -25 > $O/gpu_tests.log
else timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/gpu_tests.log; fi
timeout 900 python bench.py > $O/bench_c2.log 2>&1; tail -1 $O/bench_c2.log > $O/bench_c2.json
timeout 300 python tools/bench_fleet_loop.py 2>/dev/null | tail -1 > $O/fleet_loop.json
cat $O/gpu_tests.log; cut -c1-2500 $O/bench_c2.json; echo; cut -c1-600 $O/fleet_loop.json
