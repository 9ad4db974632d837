#!/bin/bash
# Kernel traces of the bfloat16-input workloads only (run from the repository root on the GPU box):
#   tools/profile_bf16.sh <tag>      output under gpurun_out/<tag>_*
TAG=$1; R=$(pwd); O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
step() { echo "== $1" >&2; shift; "$@" || { echo "FAILED: $*"; exit 1; }; }
for w in unet resunet dsunet unet20; do
  step "$w bf16 kernel trace" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_${w}_bf16 -o k -- python3 $R/tools/bench_configs.py --only $w --dtype bf16 --steps 5 --warmup 2 > $O/${TAG}_${w}_bf16.json 2> $O/${TAG}_${w}_bf16.err
done
echo done
