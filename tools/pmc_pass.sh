#!/bin/bash
# One rocprofv3 counter pass over a short bench run.  Usage: tools/pmc_pass.sh <tag> "<counters>"
TAG=$1; CTRS=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export MRL_ANY_ORDER=0 MRL_TOWER_STREAMS=0   # both towers in every launch, as in the bench's own per-kernel timing pass
timeout -s KILL 170 rocprofv3 --pmc $CTRS --output-format csv -d $R/gpurun_out/${TAG} -o p -- python3 $R/bench.py --num-envs 4096 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-secondary > $R/gpurun_out/${TAG}.log 2>&1
echo rc=$?; ls $R/gpurun_out/${TAG} | head
