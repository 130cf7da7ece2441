#!/bin/bash
# what the driver does at round end, on one box: pytest -m gpu (-x), smoke(), bench.py
mkdir -p gpurun_out
{
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -3
timeout 900 python bench.py --gpus 1 --steps 200 --warmup 20 2>&1 | tail -1
} > gpurun_out/r2_verify.log 2>&1
cat gpurun_out/r2_verify.log | cut -c1-1500
