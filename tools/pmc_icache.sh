#!/bin/bash
# study: instruction-cache behaviour of K1 on C2 (separate --pmc passes, no trace domains)
R=$PWD; OUT=$R/gpurun_out/pmc_icache; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQC_TC_INST_REQ"; do
  n=$(echo $set | cut -d" " -f1)
  rocprofv3 --pmc $set --output-format csv -d $OUT -o pmc_$n -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/pmc_$n.log 2>&1
done
cd $R; python tools/rocprof_summary.py pmc $OUT
