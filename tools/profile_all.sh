#!/bin/bash
# Collect the rocprofv3 evidence for `python bench.py --workload W` on the GPU box:
#   kernel trace + stats, then PMC counters in SEPARATE passes (never combined with trace domains).
# usage: bash tools/profile_all.sh <tag> [workload=C2] [steps=50]   -> gpurun_out/prof_<tag>[_<workload>]/
set -u
TAG=${1:-r02}
W=${2:-C2}
STEPS=${3:-50}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
[ "$W" != "C2" ] && OUT=${OUT}_$W
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
PSTEPS=$(( STEPS < 10 ? STEPS : 10 ))
BENCH="python $R/bench.py --workload $W --steps $STEPS --warmup 2 --no-cpu-baseline --no-pcie --no-others"
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- $BENCH > $OUT/trace.log 2>&1
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $set | cut -d" " -f1)
  rocprofv3 --pmc $set --output-format csv -d $OUT -o pmc_$n -- python $R/bench.py --workload $W --steps $PSTEPS --warmup 1 --settle-ms 0 --no-cpu-baseline --no-pcie --no-others > $OUT/pmc_$n.log 2>&1
done
cd $R
python tools/rocprof_summary.py kernels $OUT/trace_results.db > $OUT/kernels.txt 2>&1
python tools/rocprof_summary.py pmc $OUT > $OUT/pmc.txt 2>&1
# (the FETCH_SIZE / WRITE_SIZE calibration on K1's access shapes, when tools/profile_k1_traffic.sh has run for this tag)
CAL=$R/gpurun_out/prof_${TAG}_k1cal/calibration.json
[ -f $CAL ] || CAL=""
python tools/rocprof_summary.py traffic $OUT $W k_solve $CAL > $OUT/traffic_entry.json 2>&1
cat $OUT/kernels.txt $OUT/pmc.txt $OUT/traffic_entry.json
