#!/bin/bash
# usage: tools/pmc_stall_pass.sh <tag> <env> <N> -- where do the non-issuing quads go: instruction cache, LDS waits, fetch queue
set -e
tag=$1; env=$2; N=$3
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
i=0
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_MISC" \
           "SQ_WAVES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pmc -d $out/p$i -o p$i --output-format csv -- python3 tools/prof_fd.py $env $N 2 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; }
done
python3 tools/pmc_summary.py $out
