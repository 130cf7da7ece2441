#!/bin/bash
# One gpurun call that regenerates every profiles/ artefact of the current build.
# usage: bash tools/final_refresh.sh <tag>    -> gpurun_out/final_<tag>/ (+ gpurun_out/prof_<tag>/)
set -u
TAG=${1:-r01}
O=gpurun_out/final_$TAG
mkdir -p $O
timeout 900 bash tools/profile_all.sh $TAG > $O/profile_all.txt 2>&1
timeout 400 python bench.py > $O/bench_c2.log 2>&1; tail -1 $O/bench_c2.log > $O/bench_c2.json
timeout 600 python tools/parity_report.py > $O/parity_report.txt 2>&1
{
  timeout 120 python tools/bench_host_path.py 2>/dev/null | tail -1
  timeout 120 python tools/bench_tick_latency.py 2>/dev/null | tail -1
  timeout 300 python bench.py --workload C3 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1
  timeout 300 python bench.py --workload C5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1
} > $O/other_configs.jsonl
for b in 4096 32768 262144; do
  timeout 300 python bench.py --batch $b --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch $b %.4g solves/s kernel_ms %.4f' % (d['value'], d['roofline']['kernel_ms']))"
done > $O/batch_scaling.txt
timeout 300 python tools/bench_fleet_loop.py 2>/dev/null | tail -1 > $O/fleet_loop.json
timeout 300 python tools/bench_fleet_loop.py --pool 2>/dev/null | tail -1 > $O/fleet_loop_pool.json
for p in 128 512 2048; do timeout 120 python tools/bench_carrot.py --poses $p 2>/dev/null | tail -1; done > $O/carrot_hbm.jsonl
