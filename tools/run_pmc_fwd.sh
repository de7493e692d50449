#!/bin/bash
# PMC passes over the DCNv2 forward kernel (one counter group per pass, no trace domains besides --kernel-trace).
# Usage: tools/run_pmc_fwd.sh <tag>  -> gpurun_out/pmc_fwd_<tag>/...  (PMC_SHAPE / PMC_OFF_STD / PMC_CONTRACTION pass through; PMC_SCRIPT=tools/pmc_conv.py for the MFMA convolution)
set -u
cd "$(dirname "$0")/.."
tag=${1:-r03}
out=gpurun_out/pmc_fwd_$tag
mkdir -p $out
export TMPDIR=/tmp
python3 -c "import torch" 2>/dev/null
i=0
while read -r group; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $group --output-format csv -d $out/p$i -o p$i -- python3 ${PMC_SCRIPT:-tools/pmc_dcn.py} > $out/p$i.log 2>&1
  rc=$?
  echo "pass $i ($group): rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping"; exit 1; fi
done <<GROUPS
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum
GROUPS
python3 tools/pmc_reduce.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
