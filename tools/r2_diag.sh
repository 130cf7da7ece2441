#!/bin/bash
mkdir -p gpurun_out
{
timeout 300 python tools/diag_direction.py --steps 8 --iteration 1 --count 32
} > gpurun_out/r2_diag10.log 2>&1
cat gpurun_out/r2_diag10.log
