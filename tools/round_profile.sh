#!/bin/bash
# usage (on the GPU box): bash tools/round_profile.sh <tag>   -- bench line + kernel-trace stats + PMC passes at the bench size
set -e
tag=$1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_$tag
timeout -k 10 600 python3 bench.py --steps 20 --warmup 3 > $R/gpurun_out/prof_$tag/bench.json
cat $R/gpurun_out/prof_$tag/bench.json
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag/kt -o kt --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu > $R/gpurun_out/prof_$tag/kt.log 2>&1
bash tools/pmc_pass.sh $tag HalfCheetahEnv 163840 > $R/gpurun_out/prof_$tag/pmc.log 2>&1
tail -40 $R/gpurun_out/prof_$tag/pmc.log
