#!/bin/bash
# the bench lines of r2_profile.sh on their own (after profiles/hbm_traffic.json matches the build)
TAG=${1:-r02a}
O=gpurun_out/final_$TAG
mkdir -p $O
timeout 600 python bench.py > $O/bench_c2.log 2>&1; tail -1 $O/bench_c2.log > $O/bench_c2.json
{
  timeout 300 python bench.py --workload C3 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1
  timeout 300 python bench.py --workload C4 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1
  timeout 300 python bench.py --workload C5 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1
  timeout 120 python tools/bench_tick_latency.py 2>/dev/null | tail -1
} > $O/other_configs.jsonl
cut -c1-600 $O/bench_c2.json
