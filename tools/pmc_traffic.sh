#!/bin/bash
# HBM traffic of the benchmark's kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 counter passes over one
# short full-size bench run each (the TCC block cannot hold both at once).  Usage: tools/pmc_traffic.sh <tag>
TAG=${1:-traffic}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export MRL_ANY_ORDER=0 MRL_TOWER_STREAMS=0   # both towers in every launch, as in the bench's own per-kernel timing pass
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 170 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/${TAG}_$c -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-secondary > $R/gpurun_out/${TAG}_$c.log 2>&1
  rc=$?; echo "$c rc=$rc"; [ $rc -ne 0 ] && exit $rc
done
exit 0
