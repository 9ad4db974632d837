#!/bin/bash
# Collect one round's profiles on the GPU box (run from the repository root):  tools/profile_round.sh <tag>
# One rocprofv3 run per counter set; the program itself follows "--".  Output under gpurun_out/<tag>_*.
TAG=$1; R=$(pwd); O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
step() { echo "== $1" >&2; shift; "$@" || { echo "FAILED: $*"; exit 1; }; }
step "headline kernel trace" timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -o b -- python3 $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
step "headline MFMA utilisation" timeout -s KILL 300 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d $O/${TAG}_pmc_mfma -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/${TAG}_pmc_mfma.log 2>&1
step "headline FETCH" timeout -s KILL 300 rocprofv3 --output-format csv --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $O/${TAG}_pmc_ea -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/${TAG}_pmc_ea.log 2>&1
step "ResUNet kernel trace" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_resunet -o res -- python3 $R/tools/bench_configs.py --only resunet --steps 5 --warmup 2 > $O/${TAG}_resunet.json 2> $O/${TAG}_resunet.err
echo done
