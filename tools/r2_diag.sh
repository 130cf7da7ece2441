#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
{
echo "== phase timing C5"; NEO_MPC_LIB=$PWD/neo_mpc_planner2_amd/libneo_mpc_timing.so timeout 300 python tools/phase_timing.py C5
cd /tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"; do
  n=$(echo $set | cut -d" " -f1)
  rm -rf /tmp/pmcq; rocprofv3 --pmc $set --output-format csv -d /tmp/pmcq -o p -- python $R/bench.py --workload C5 --steps 2 --warmup 1 --no-cpu-baseline --no-pcie > /tmp/pmcq.log 2>&1
  python $R/tools/rocprof_summary.py pmc /tmp/pmcq
done
} > $R/gpurun_out/r2_pmc_c5.log 2>&1
cat $R/gpurun_out/r2_pmc_c5.log
