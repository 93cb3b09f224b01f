#!/bin/bash
# A/B of the small-batch kernels on the GPU box: gradients old vs new for several networks, then step times.
set -o pipefail
mkdir -p gpurun_out
for m in 0 3 7; do
  echo "== dump MRL_CHAIN=$m"; MRL_CHAIN=$m timeout -k 10 240 python -u tools/chain_ab.py dump gpurun_out/ab_$m.pt || exit 1
done
timeout 120 python tools/chain_ab.py cmp gpurun_out/ab_0.pt gpurun_out/ab_3.pt; r1=$?
timeout 120 python tools/chain_ab.py cmp gpurun_out/ab_0.pt gpurun_out/ab_7.pt; r3=$?
echo "cmp status: separate chains $r1, one launch $r3"
for m in 0 3 7; do
  MRL_CHAIN=$m timeout -k 10 240 python -u tools/chain_ab.py time || exit 1
done
[ $r1 = 0 ] && [ $r3 = 0 ]
