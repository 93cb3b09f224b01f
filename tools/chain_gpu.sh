#!/bin/bash
# A/B of the small-batch kernels on the GPU box: gradients of the per-layer path against the chain kernels for several networks,
# then step times.  Usage: tools/chain_gpu.sh [masks to compare with 0, default "3 7 15"]
set -o pipefail
mkdir -p gpurun_out
MASKS=${1:-"3 7 15"}
st=0
for m in 0 $MASKS; do
  echo "== dump MRL_CHAIN=$m"; MRL_CHAIN=$m timeout -k 10 240 python -u tools/chain_ab.py dump gpurun_out/ab_$m.pt || exit 1
done
for m in $MASKS; do
  timeout 120 python tools/chain_ab.py cmp gpurun_out/ab_0.pt gpurun_out/ab_$m.pt || st=1
done
echo "cmp status: $st"
for m in 0 $MASKS; do
  MRL_CHAIN=$m timeout -k 10 240 python -u tools/chain_ab.py time || exit 1
done
exit $st
