# Matrix-pipe busy counters of both minibatch regimes in one GPU call (one rocprofv3 --pmc pass each); tables under gpurun_out/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16"
bash tools/pmc_pass.sh r3m "$C" || exit 1
python tools/mfma_busy_table.py $(find gpurun_out/r3m -name 'p_counter_collection.csv' | head -1) > gpurun_out/r3m_table.txt || exit 1
cd /tmp && export TMPDIR=/tmp MRL_ANY_ORDER=0 MRL_TOWER_STREAMS=0
timeout -s KILL 170 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/r3m2048 -o p -- python3 $R/bench.py --num-envs 4096 --minibatch 2048 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-secondary > $R/gpurun_out/r3m2048.log 2>&1 || exit 1
cd $R && python tools/mfma_busy_table.py $(find gpurun_out/r3m2048 -name 'p_counter_collection.csv' | head -1) 10 > gpurun_out/r3m2048_table.txt
