#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <env> <N> -- runs the rocprofv3 counter passes (one run each) into gpurun_out/pmc_<tag>
set -e
tag=$1; env=$2; N=$3
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
i=0
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM" \
           "SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_CBRANCH SQ_WAVE32_INSTS" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pmc -d $out/p$i -o p$i --output-format csv -- python3 tools/prof_fd.py $env $N 2 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; }
done
python3 tools/pmc_summary.py $out
