#!/bin/bash
# Ablation of the GEMM main loop in the instrumented build: per-step cycles with parts of the loop switched off.
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export MRL_KTRACE="$1" MRL_KTRACE_M="$2" MRL_KTRACE_Z=1
for a in 0 1 2 4 3 7 8 9; do
  echo "== ablate mask $a (1 staging off, 2 fragment reads off, 4 barrier off, 8 MFMA off)"
  MRL_KTRACE_ABLATE=$a timeout -k 10 120 python3 -u tools/ktrace_run.py | grep "wave 0: total\|blocks (us" || exit $?
done
