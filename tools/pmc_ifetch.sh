#!/bin/bash
# usage: tools/pmc_ifetch.sh <variant.so | default> ... -- instruction-fetch and LDS latency per library variant
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" != default ]; then export MBRL_LIB=$GRAFT_REPO_ROOT/tests/_build/variants/$v.so; else unset MBRL_LIB; fi
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_if_$v
  mkdir -p $out
  for pmc in InstrFetchLatency "SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_IFETCH SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"; do
    n=$(echo $pmc | cut -c1-12 | tr ' ' '_')
    timeout -s KILL 120 rocprofv3 --pmc $pmc -d $out/$n -o p --output-format csv -- python3 tools/prof_fd.py HalfCheetahEnv 163840 2 > $out/$n.log 2>&1 || { echo "pass failed"; tail -5 $out/$n.log; }
  done
  echo "== $v"; python3 tools/pmc_summary.py $out | grep -v "^per-wave SQ_INSTS_[LSV][DM]"
done
