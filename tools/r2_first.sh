#!/bin/bash
# round 2, first GPU call: parity tests + C3/C5 with the Riccati direction against the round-1 directions
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity.py::test_p2_p3_against_reference_slsqp_solves 2>&1 | tail -30 > gpurun_out/r2_tests_1.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "p2_p3" 2>&1 | tail -30 > gpurun_out/r2_tests_1b.log
{
for m in 0 1 2; do timeout 300 python bench.py --workload C3 --steps 3 --warmup 1 --no-cpu-baseline --method $m 2>/dev/null | tail -1; done
for m in 0 1; do timeout 300 python bench.py --workload C5 --steps 3 --warmup 1 --no-cpu-baseline --method $m 2>/dev/null | tail -1; done
for w in 2 3 4; do NEO_MPC_SOLVE_WAVES=$w timeout 300 python bench.py --workload C5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1; done
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --method 3 2>/dev/null | tail -1
} > gpurun_out/r2_bench_1.jsonl
cat gpurun_out/r2_tests_1.log gpurun_out/r2_tests_1b.log
python - <<'PY'
import json
for l in open("gpurun_out/r2_bench_1.jsonl"):
    try: d=json.loads(l)
    except Exception: print("bad line", l[:100]); continue
    print(d["config"]["workload"][:40], "| %.4g solves/s | kernel %.3f ms | iters %.2f conv %.4f" % (d["value"], d["roofline"]["kernel_ms"], d["solver"]["mean_iterations"], d["solver"]["converged_frac"]))
PY
