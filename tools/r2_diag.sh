#!/bin/bash
mkdir -p gpurun_out
{
for w in C5 C3; do echo "== phase timing $w"; NEO_MPC_LIB=$PWD/neo_mpc_planner2_amd/libneo_mpc_timing.so timeout 300 python tools/phase_timing.py $w; done
} > gpurun_out/r2_phase2.log 2>&1
cat gpurun_out/r2_phase2.log
