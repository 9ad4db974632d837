#!/bin/bash
# PMC passes over DSUNet's forward on a bfloat16 input (the 16-bit conv kernels), one rocprofv3 run per counter set.
# usage: tools/pmc_b16.sh <tag>     (run on the GPU box from the repository root)
TAG=$1; R=$(pwd); OUT=$R/gpurun_out/pmc_b16_$TAG
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
run() { timeout -s KILL 120 rocprofv3 --output-format csv --pmc $2 -d "$OUT/$1" -o p -- python3 $R/tools/bench_configs.py --only dsunet --dtype bf16 --steps 1 --warmup 1 > "$OUT/$1.log" 2>&1 || { echo "pass $1 failed"; tail -n 5 "$OUT/$1.log"; exit 1; }; }
run sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
run sq2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in ("sq", "sq2"):
    f = glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True)
    if not f:
        print(d, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "conv_b16" not in k:
            continue
        k = k[k.index("conv_b16"):k.index(">") + 1] + " grid " + r["Grid_Size"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    for k in sorted(acc):
        print(d, k, " ".join("{}={:.4g}".format(c, acc[k][c] / cnt[k][c]) for c in sorted(acc[k])))
PY
