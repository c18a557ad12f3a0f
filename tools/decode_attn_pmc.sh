#!/bin/bash
# PMC passes over tools/decode_attn_bench.py (run ON the GPU box): bash tools/decode_attn_pmc.sh B
# (at most two TA_* counters per pass: four at once exceed the block's counter registers - rocprofv3 error 38)
B=${1:-64}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/dattn_pmc
rm -rf $O && mkdir -p $O
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS" "TA_BUSY_avr TA_BUSY_max" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p$i -o g -- python3 tools/decode_attn_bench.py $B 2300 > $O/p$i.log 2>&1 || echo "pass $i ($C) failed" >> $O/errors.txt
done
python tools/pmc_kernels.py $O decode_attn > $O/summary.txt
cat $O/errors.txt >> $O/summary.txt 2>/dev/null
cat $O/summary.txt
