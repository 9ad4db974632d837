#!/bin/bash
# PMC passes over one timing binary of the residual-block kernel (one rocprofv3 run per counter set).
# usage: tools/pmc_rb.sh <binary> <tag>     (run on the GPU box from the repository root)
BIN=$(realpath "$1"); TAG=$2; OUT=$(realpath gpurun_out)/pmc_$TAG
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
run() { timeout -s KILL 90 rocprofv3 --output-format csv --pmc $2 -d "$OUT/$1" -o p -- "$BIN" 8 257 501 > "$OUT/$1.log" 2>&1 || { echo "pass $1 failed"; tail -n 5 "$OUT/$1.log"; exit 1; }; }
run sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS"
run sq2 "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU"
run tcp "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum"
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
run tcc2 "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_READ_REQ_sum TCC_WRITE_REQ_sum"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in ("sq", "sq2", "tcp", "tcc", "tcc2"):
    f = glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True)
    if not f:
        print(d, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        k = k[k.index("resblock_dma_kernel"):k.index(">") + 1] if "resblock_dma_kernel" in k else k[:30]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print(d, k, " ".join("{}={:.4g}".format(c, sum(v) / len(v)) for c, v in sorted(acc[k].items())))
PY
