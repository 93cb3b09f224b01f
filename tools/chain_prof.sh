#!/bin/bash
# Kernel-trace summary of the optimizer step at 2,048 rows.  Usage: tools/chain_prof.sh <tag> [MRL_CHAIN]
set -o pipefail
TAG=${1:-cp}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export MRL_CHAIN=${2:-3}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o k -- python3 $R/tools/chain_ab.py time > $R/gpurun_out/${TAG}_prof.log 2>&1
rc=$?
f=$(find $R/gpurun_out/${TAG}_prof -name 'k_kernel_stats.csv' | head -1)
[ -n "$f" ] && python3 $R/tools/prof_summary.py $f 30 | tee $R/gpurun_out/${TAG}_summary.txt
exit $rc
