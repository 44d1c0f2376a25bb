#!/bin/bash
# PMC breakdown of one lab binary (run through gpurun): wave-cycle accounting + LDS / VMEM pressure.
BIN=${1:-tools/lab/va_lab}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_lab
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d gpurun_out/pmc_lab/a -o p -- $BIN > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc_lab/b -o p -- $BIN > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_IFETCH SQ_VMEM_TA_ADDR_FIFO_FULL --output-format csv -d gpurun_out/pmc_lab/c -o p -- $BIN > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob("gpurun_out/pmc_lab/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
        a = acc[k][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
for k, cs in acc.items():
    if "pack" in k: continue
    print(k)
    wc = cs.get("SQ_WAVE_CYCLES", [1, 1.0]); wcv = wc[1] / max(wc[0], 1)
    for c, (n, v) in sorted(cs.items()):
        print(f"   {c:32s} {v / n:16.0f}   {v / n / wcv:8.3f} of WAVE_CYCLES")
PY
