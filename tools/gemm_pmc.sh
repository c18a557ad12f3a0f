#!/bin/bash
# PMC passes over tools/gemm_bench.py for one shape (run ON the GPU box): bash tools/gemm_pmc.sh TILE M N K
set -e
T=$1; M=$2; N=$3; K=$4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/gemm_pmc_$T
rm -rf $O && mkdir -p $O
export VIS_GEMM_TILE=$T
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p$i -o g -- python3 tools/gemm_bench.py $M $N $K 3 > $O/p$i.log 2>&1 || echo "pass $i failed"
done
python tools/pmc_kernels.py $O gemm_bf16 > $O/summary.txt
cat $O/summary.txt
