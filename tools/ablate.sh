#!/bin/bash
# Diagnostic: builds of the library with parts of one split-bf16 kernel switched off (results are garbage; timing only) and
# the per-kernel table of one full-size update for each.  Build on the CPU: tools/ablate.sh build "MRL_ABLATE_DGRAD=1 MRL_ABLATE_FWD=1 MRL_ABLATE_WGRAD=2 ..." (any -D macro)
# Run on the GPU box: tools/ablate.sh run "MRL_ABLATE_DGRAD=1 ..."
R=${GRAFT_REPO_ROOT:-$(pwd)}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -shared -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops -Wno-unused-value"
if [ "$1" = build ]; then
  mkdir -p $R/tools/bin
  for v in $2; do
    /opt/rocm/bin/hipcc $FLAGS -D$v $R/mineral_amd/csrc/ppo_capi.hip $R/mineral_amd/csrc/replay.hip -o $R/tools/bin/libmrl_abl_$v.so 2>/dev/null &
  done
  wait; ls $R/tools/bin/*.so
else
  for v in base $2; do
    lib=$R/tools/bin/libmrl_abl_$v.so; [ $v = base ] && lib=$R/mineral_amd/libmineral_ppo.so
    echo "== $v"
    MRL_LIB=$lib timeout -k 10 120 python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step %.1f' % d['ms_per_step'])
for k in d['roofline']['kernels'][:9]:
    print('  %8.1f us  %s' % (k['avg_us'], k['name']))" || exit 1
  done
fi
