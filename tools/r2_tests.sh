#!/bin/bash
# all GPU tests (no -x) + the K1-vs-mirror diagnostic + C2/C3/C5 lines
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r2_tests.log
timeout 300 python tools/diag_mirror.py --steps 3 8 32 > gpurun_out/r2_diag.log 2>&1
{
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pcie 2>/dev/null | tail -1
timeout 300 python bench.py --workload C3 --steps 3 --warmup 1 --no-cpu-baseline --no-pcie 2>/dev/null | tail -1
timeout 300 python bench.py --workload C5 --steps 3 --warmup 1 --no-cpu-baseline --no-pcie 2>/dev/null | tail -1
} > gpurun_out/r2_bench_2.jsonl
cat gpurun_out/r2_tests.log; grep -v amdgpu gpurun_out/r2_diag.log | grep "cap 100"
python - <<'PY'
import json
for l in open("gpurun_out/r2_bench_2.jsonl"):
    d=json.loads(l)
    print(d["config"]["workload"][:40], "| %.4g solves/s | kernel %.4f ms | iters %.2f max %d conv %.4f" % (d["value"], d["roofline"]["kernel_ms"], d["solver"]["mean_iterations"], d["solver"]["max_iterations_seen"], d["solver"]["converged_frac"]))
PY
