# A/B of the two kernel families around the row threshold (run on the GPU box): ms per update at several minibatch sizes
# with the boundary MRL_SMALL_M below and above the minibatch.
set -o pipefail
B="python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-kernel-timing"
for cfg in "8192 8192" "8192 4096" "4096 8192" "4096 2048" "16384 8192" "16384 16384"; do
  set -- $cfg
  echo "== minibatch $1 MRL_SMALL_M=$2"
  MRL_SMALL_M=$2 timeout -k 10 200 $B --minibatch $1 2>gpurun_out/thr.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step_median'],1), d['param_sha256'][:12])" || exit 1
done
