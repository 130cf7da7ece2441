#!/bin/bash
mkdir -p gpurun_out
{
echo "== fuzz seed 0"; timeout 300 python tools/diag_mirror.py --fuzz-seed 0 --count 64
timeout 600 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_edges.py -m gpu -q 2>&1 | tail -5
} > gpurun_out/r2_diag9.log 2>&1
cat gpurun_out/r2_diag9.log
