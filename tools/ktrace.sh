#!/bin/bash
# Runs tools/ktrace_run.py for the GEMM instantiations given as "<template args>|M" pairs; stops at the first failure.
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for spec in "$@"; do
  export MRL_KTRACE="${spec%%|*}" MRL_KTRACE_M="${spec##*|}"
  timeout -k 10 120 python3 -u tools/ktrace_run.py || exit $?
done
