#!/bin/bash
# Evidence in one GPU call: bench line (both minibatch regimes), rocprofv3 kernel summaries of an update at minibatch
# 32768 and at 2048, SAC and SHAC bench lines.  Usage: tools/r03_profiles.sh <tag>
set -o pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -u bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || { tail -5 $O/${TAG}_bench.err; exit 1; }
timeout -k 10 300 python -u tools/sac_bench.py $O/${TAG}_sac_bench.json > $O/${TAG}_sac.log 2>&1 || { tail -5 $O/${TAG}_sac.log; exit 1; }
timeout -k 10 300 python -u tools/shac_bench.py $O/${TAG}_shac_bench.json > $O/${TAG}_shac.log 2>&1 || { tail -5 $O/${TAG}_shac.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
export MRL_ANY_ORDER=0 MRL_TOWER_STREAMS=0   # per-kernel durations want every launch in order
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o k -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-secondary > $O/${TAG}_prof.log 2>&1 || exit 1
python3 $R/tools/prof_summary.py $(find $O/${TAG}_prof -name 'k_kernel_stats.csv' | head -1) 40 > $O/${TAG}_kernel_summary.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof2048 -o k -- python3 $R/bench.py --steps 1 --warmup 1 --minibatch 2048 --no-cpu-baseline --no-kernel-timing --no-secondary > $O/${TAG}_prof2048.log 2>&1 || exit 1
python3 $R/tools/prof_summary.py $(find $O/${TAG}_prof2048 -name 'k_kernel_stats.csv' | head -1) 40 > $O/${TAG}_kernel_summary_mb2048.txt
head -12 $O/${TAG}_kernel_summary_mb2048.txt
rm -rf $O/${TAG}_prof/*trace.csv $O/${TAG}_prof2048/*trace.csv
