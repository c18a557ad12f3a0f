#!/bin/bash
# MFMA-busy / wait / LDS counters of the fp8 ping-pong tile on the tower shapes VERDICT r4 item 3b names (grids 1463 + remainder and
# 1155 at 4 images) and on gate/up, with the direct epilogue (VIS_GEMM_WIDE=0 = r04's code path) and the LDS-staged one (r05).
# Run ON the GPU box: bash tools/fp8_gemm_pmc.sh  ->  gpurun_out/fp8_pmc/summary.txt (copy to profiles/r05_fp8_gemm_pmc.txt)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/fp8_pmc
rm -rf $O && mkdir -p $O
export VIS_GEMM8_TILE=4
: > $O/summary.txt
for SH in "fc1 19600 5120 1280 1" "qkv 19600 3840 1280 0" "gateup 5156 37888 3584 3"; do
  set -- $SH
  for W in 0 1; do
    export VIS_GEMM_WIDE=$W
    D=$O/$1_w$W
    mkdir -p $D
    i=0
    for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"; do
      i=$((i+1))
      rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D/p$i -o g -- python3 tools/gemm8_one.py $2 $3 $4 $5 3 > $D/p$i.log 2>&1 || echo "$1 wide=$W pass $i failed" >> $O/summary.txt
    done
    echo "## $1: M=$2 N=$3 K=$4 act=$5, VIS_GEMM_WIDE=$W ($([ $W = 0 ] && echo 'direct epilogue, r04' || echo 'LDS-staged epilogue, r05'))" >> $O/summary.txt
    python3 tools/pmc_kernels.py $D gemm_fp8_256x256_pp >> $O/summary.txt
    python3 - $D >> $O/summary.txt <<'PY'
import csv, glob, sys, statistics
d = []
for f in glob.glob(sys.argv[1] + "/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_fp8_256x256_pp" in r["Kernel_Name"]:
            d.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
if d:
    print(f"   median duration under the counter pass: {statistics.median(d):.1f} us ({len(d)} dispatches)")
PY
  done
done
cat $O/summary.txt
