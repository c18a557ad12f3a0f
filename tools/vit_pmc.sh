#!/bin/bash
# Run ON the GPU box: PMC counters of the ViT attention kernels (tools/attn_bench.py --which vit runs the 4-wave and the
# 12-wave pipelined 32x32x16 kernels; VIS_ATTN80=16 the older 16x16x32 one).  Separate rocprofv3 passes, --pmc +
# --kernel-trace only.  Output: gpurun_out/pmc_vit_<tag>/summary.txt
R=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_vit_$R
rm -rf $O && mkdir -p $O
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/vit_$i -o p -- python3 tools/attn_bench.py --which vit --plan 0 --reps 3 > $O/vit_$i.log 2>&1 || echo "pass $i ($C) failed" >> $O/errors.txt
done
{
  echo "# rocprofv3 --pmc passes (separate runs), MI355X, tools/attn_bench.py --which vit: 4900 patches x 16 heads x d 80"
  python3 tools/pmc_kernels.py $O attn_
  python3 - "$O" <<'PY'
import csv, glob, sys, collections
dur = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/vit_1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn_" in r["Kernel_Name"]:
            dur[r["Kernel_Name"].split("(")[0][-40:]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in dur.items():
    print(f"{k}: median duration under the counter pass {sorted(v)[len(v)//2] / 1e3:.1f} us ({len(v)} dispatches)")
PY
  cat $O/errors.txt 2>/dev/null
} > $O/summary.txt
cat $O/summary.txt
find $O -name "*.csv" -size +2M -delete
