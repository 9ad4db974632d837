#!/bin/bash
# Collect one round's profiles on the GPU box (run from the repository root):  tools/profile_round.sh <tag>
# One rocprofv3 run per counter set; the program itself follows "--".  Output under gpurun_out/<tag>_*.
TAG=$1; R=$(pwd); O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
step() { echo "== $1" >&2; shift; "$@" || { echo "FAILED: $*"; exit 1; }; }
step "headline kernel trace" timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -o b -- python3 $R/bench.py --no-other-configs > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
step "headline MFMA utilisation" timeout -s KILL 300 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d $O/${TAG}_pmc_mfma -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $O/${TAG}_pmc_mfma.log 2>&1
# HBM traffic from the L2's memory-side request counters, by request size (4 TCC counters per pass at most):
# reads = 32 B x RDREQ_32B + 64 B x RDREQ_64B + 128 B x RDREQ_128B, writes = 64 B x WRREQ_64B + 32 B x (WRREQ - WRREQ_64B)
step "headline read requests" timeout -s KILL 300 rocprofv3 --output-format csv --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $O/${TAG}_pmc_rd -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $O/${TAG}_pmc_rd.log 2>&1
step "headline write requests" timeout -s KILL 300 rocprofv3 --output-format csv --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $O/${TAG}_pmc_wr -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $O/${TAG}_pmc_wr.log 2>&1
step "ResUNet kernel trace" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_resunet -o res -- python3 $R/tools/bench_configs.py --only resunet --steps 5 --warmup 2 > $O/${TAG}_resunet.json 2> $O/${TAG}_resunet.err
step "ResUNet bf16 kernel trace" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_resunet_bf16 -o res -- python3 $R/tools/bench_configs.py --only resunet --dtype bf16 --steps 5 --warmup 2 > $O/${TAG}_resunet_bf16.json 2> $O/${TAG}_resunet_bf16.err
step "DSUNet bf16 kernel trace" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_dsunet_bf16 -o ds -- python3 $R/tools/bench_configs.py --only dsunet --dtype bf16 --steps 5 --warmup 2 > $O/${TAG}_dsunet_bf16.json 2> $O/${TAG}_dsunet_bf16.err
step "UNet bf16 kernel trace" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_unet_bf16 -o un -- python3 $R/tools/bench_configs.py --only unet --dtype bf16 --steps 5 --warmup 2 > $O/${TAG}_unet_bf16.json 2> $O/${TAG}_unet_bf16.err
step "train step kernel trace" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_train -o tr -- python3 $R/bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > $O/${TAG}_train.json 2> $O/${TAG}_train.err
echo done
