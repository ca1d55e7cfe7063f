#!/bin/bash
OUT=gpurun_out/${1:-pmc}; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -o "Name:\s*[A-Za-z0-9_]*" $OUT/counters.txt | sort -u | awk '{print $2}' | tr '\n' ' ' > $OUT/counter_names.txt
for t in 64064 3064 3128; do
  echo "== tile $t"
  python tools/pmc_kernel.py dcn_ "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
     "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
     "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
     "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
     "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCC_HIT_sum TCC_MISS_sum" \
     -- python tools/bench_conv.py d64_128,d128_64 $t 2>&1 | tee $OUT/pmc_$t.txt
done
