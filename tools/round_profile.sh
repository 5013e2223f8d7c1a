#!/bin/bash
# usage (on the GPU box): bash tools/round_profile.sh <tag>
#   bench line + rocprofv3 kernel-trace stats of the same command + PMC passes at the bench size; everything the judge
#   reads is gathered in gpurun_out/prof_<tag>/ under the names profiles/ uses (copy them there afterwards)
set -e
tag=$1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=$R/gpurun_out/prof_$tag
mkdir -p $P
timeout -k 10 600 python3 bench.py --steps 20 --warmup 3 > $P/${tag}_bench.json
cat $P/${tag}_bench.json
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $P/kt -o kt --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-extra > $P/kt.log 2>&1
cp $(ls $P/kt/*/*kernel_stats.csv $P/kt/*kernel_stats.csv 2>/dev/null | head -1) $P/${tag}_kernel_stats.csv
head -5 $P/${tag}_kernel_stats.csv
bash tools/pmc_pass.sh $tag HalfCheetahEnv 163840 > $P/pmc.log 2>&1
tail -50 $P/pmc.log
cp $R/gpurun_out/pmc_$tag/summary.txt $P/${tag}_pmc_summary.txt
python3 tools/traffic.py $R/gpurun_out/pmc_$tag 163840 $P/${tag}_traffic.json
python3 tools/pmc_flops.py $R/gpurun_out/pmc_$tag 163840 0 $P/${tag}_pmc_flops.json
