#!/bin/bash
# Kernel trace of the training step on the GPU box (run from the repository root):  tools/profile_train.sh <tag>
TAG=$1; R=$(pwd); O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_train -o t -- python3 $R/bench.py --mode train --steps 10 --warmup 3 > $O/${TAG}_train.json 2> $O/${TAG}_train.err || { echo FAILED; exit 1; }
python3 $R/tools/trace_by_grid.py "$O/${TAG}_train/*/*kernel_trace.csv" 70 > $O/${TAG}_train_by_grid.txt 2>&1 || python3 $R/tools/trace_by_grid.py "$O/${TAG}_train/*kernel_trace.csv" 70 > $O/${TAG}_train_by_grid.txt
echo done
