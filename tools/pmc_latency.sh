set -e
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_lat
mkdir -p $out
i=0
for pmc in "InstrFetchLatency" "LdsLatency" "SQ_WAVES SQ_INSTS_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pmc -d $out/p$i -o p$i --output-format csv -- python3 tools/prof_fd.py HalfCheetahEnv 163840 2 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; }
done
python3 tools/pmc_summary.py $out
