#!/bin/bash
# all GPU tests (no -x) + the K1-vs-mirror diagnostic + C3/C5 lines
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r2_tests.log
timeout 300 python tools/diag_mirror.py --steps 8 32 > gpurun_out/r2_diag.log 2>&1
{
timeout 300 python bench.py --workload C3 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1
timeout 300 python bench.py --workload C5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1
} > gpurun_out/r2_bench_2.jsonl
cat gpurun_out/r2_tests.log; cat gpurun_out/r2_diag.log
python - <<'PY'
import json
for l in open("gpurun_out/r2_bench_2.jsonl"):
    d=json.loads(l)
    print(d["config"]["workload"][:40], "| %.4g solves/s | kernel %.3f ms | iters %.2f conv %.4f" % (d["value"], d["roofline"]["kernel_ms"], d["solver"]["mean_iterations"], d["solver"]["converged_frac"]))
PY
