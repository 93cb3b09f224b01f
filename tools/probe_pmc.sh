#!/bin/bash
# rocprofv3 counter pass over the stand-alone probe.  Usage: tools/probe_pmc.sh <tag> "<counters>" [probe args]
TAG=$1; CTRS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $CTRS --output-format csv -d $R/gpurun_out/${TAG} -o p -- $R/tools/bin/chain_probe_0 "$@" > $R/gpurun_out/${TAG}.log 2>&1
rc=$?
f=$(find $R/gpurun_out/${TAG} -name 'p_counter_collection.csv' | head -1)
[ -n "$f" ] && python3 $R/tools/pmc_summary.py $f | grep -A12 -E "chain_|wgrad_small|reblock"
exit $rc
