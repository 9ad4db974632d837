#!/bin/bash
# HBM traffic counters of the headline bench on the GPU box (run from the repository root):  tools/profile_traffic.sh <tag>
# One rocprofv3 --pmc pass per counter; the program itself follows "--".
TAG=$1; R=$(pwd); O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/${TAG}_fetch -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/${TAG}_fetch.log 2>&1 || { echo "FETCH pass failed"; exit 1; }
timeout -s KILL 300 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/${TAG}_write -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/${TAG}_write.log 2>&1 || { echo "WRITE pass failed"; exit 1; }
echo done
