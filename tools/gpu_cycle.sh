#!/bin/bash
# One GPU round trip: parity tests, bench line, rocprofv3 kernel stats.  Usage: tools/gpu_cycle.sh <tag>
# Every GPU step runs under its own time limit and the chain stops at the first failure.
set -o pipefail
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/${TAG}_tests.log 2>&1
rc=$?; tail -3 gpurun_out/${TAG}_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python -u bench.py --steps 3 --warmup 1 > gpurun_out/${TAG}_bench.log 2>&1
rc=$?; tail -1 gpurun_out/${TAG}_bench.log; [ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
# per-kernel durations want the single-stream launch order (both towers in every launch), as in the bench's own timing pass
export MRL_ANY_ORDER=0 MRL_TOWER_STREAMS=0
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o k -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-secondary > $R/gpurun_out/${TAG}_prof.log 2>&1
rc=$?; ls $R/gpurun_out/${TAG}_prof | head; exit $rc
