#!/bin/bash
# One rocprofv3 counter pass over the 2,048-row optimizer step.  Usage: tools/chain_pmc.sh <tag> "<counters>" [MRL_CHAIN]
TAG=$1; CTRS=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
export MRL_CHAIN=${3:-3}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $CTRS --output-format csv -d $R/gpurun_out/${TAG} -o p -- python3 $R/tools/chain_ab.py time 2048 2048 32 40 > $R/gpurun_out/${TAG}.log 2>&1
rc=$?
f=$(find $R/gpurun_out/${TAG} -name 'p_counter_collection.csv' | head -1)
[ -n "$f" ] && python3 $R/tools/pmc_summary.py $f | tee $R/gpurun_out/${TAG}_summary.txt
exit $rc
