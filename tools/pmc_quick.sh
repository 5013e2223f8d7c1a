#!/bin/bash
# usage: tools/pmc_quick.sh <tag> [lib.so]  -- three rocprofv3 counter passes (instruction counts, wait/active cycles,
# fp64 mix) of the planned HalfCheetah FD launch at the bench size, library variant optional; summary in gpurun_out/pmcq_<tag>
tag=$1; lib=$2
export TMPDIR=/tmp
[ -n "$lib" ] && export MBRL_LIB=$(realpath $lib)
out=$GRAFT_REPO_ROOT/gpurun_out/pmcq_$tag
mkdir -p $out
i=0
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_WAVES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pmc -d $out/p$i -o p$i --output-format csv -- python3 tools/prof_fd.py HalfCheetahEnv ${UNITS:-163840} 2 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; exit 1; }
done
python3 tools/pmc_summary.py $out | grep per-wave
