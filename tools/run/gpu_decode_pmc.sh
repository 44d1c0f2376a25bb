#!/bin/bash
# PMC passes over the N1 operators (tools/bench_ops.py decode, our kernels only).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/decode_pmc; mkdir -p $O; rm -f $O/pmc.txt
export POEM_NO_EAGER=1
for pass in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" \
            "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
            "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/pmc -o pmc -- python tools/bench_ops.py decode > /dev/null 2>> $O/err.txt
  C=$(find $O/pmc -name "*counter_collection.csv" | head -1)
  python tools/pmc_kernels.py "$C" conv3x3 pool_head conv1x1 | tee -a $O/pmc.txt
  rm -rf $O/pmc
done
