#!/bin/bash
# PMC counters of the Winograd kernels on one shape each (micro-benchmark launches): F(2x4) generic, F(2x2) <1,2>, <1,1>, V-stationary head
OUT=gpurun_out/${1:-pmc_wino}; mkdir -p $OUT; export TMPDIR=/tmp
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
G2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
G3="SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"
G4="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum"
for spec in "wino24 c128_64 -24" "wino_kernel c128_64 -12" "wino_kernel c128_64 -11" "vs64 head 0"; do
  set -- $spec
  echo "== $1 $2 tile $3"
  python tools/pmc_kernel.py $1 "$G1" "$G2" "$G3" "$G4" -- python tools/bench_conv.py $2 $3 2>&1 | tee $OUT/pmc_$1_$2_$3.txt
done
rm -rf gpurun_out/pmc
