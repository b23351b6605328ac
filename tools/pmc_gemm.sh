#!/bin/bash
# PMC passes over one GEMM shape (tools/gemm_one.py): usage pmc_gemm.sh <outdir> kind M N K cfg
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU" \
           "SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_SALU" \
           "SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
    i=$((i+1))
    timeout -k 10 120 rocprofv3 --pmc $set -d $OUT/p$i --output-format csv -- python3 $R/tools/gemm_one.py "$@" > $OUT.p$i.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_bf16_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:32s} n={len(v)} mean={sum(v)/len(v):.4g}")
PY
