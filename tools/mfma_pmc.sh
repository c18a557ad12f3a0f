#!/bin/bash
# Run ON the GPU box: MFMA-busy / wait / cache counters for the prefill GEMM (LLM gate/up shape, ping-pong 256x256 tile)
# and both prefill attention kernels, condensed into gpurun_out/<out>/mfma_busy.txt (copy into profiles/).
#   bash tools/mfma_pmc.sh r02
# Counter passes are separate rocprofv3 runs (--pmc + --kernel-trace only).  MFMA busy fraction of a kernel =
# SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES * 4 SIMDs)  [both summed over the chip]; effective clock =
# GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (MI355X_MICROARCH.md, DVFS give-back).
R=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_$R
rm -rf $O && mkdir -p $O
run() {  # name, counters, program args...
  local name=$1; shift; local ctr=$1; shift
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/$name -o p -- python3 "$@" > $O/$name.log 2>&1 || echo "pass $name failed" >> $O/errors.txt
}
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  run gemm_$i "$C" tools/gemm_bench.py 2249 37888 3584 3
  run vit_$i "$C" tools/attn_bench.py --which vit --reps 3
  run llm_$i "$C" tools/attn_bench.py --which llm --reps 3
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o p -- python3 tools/attn_bench.py --which vit --reps 3 > $O/trace_vit.log 2>&1
{
  echo "# rocprofv3 --pmc passes, MI355X; per-dispatch averages (tools/mfma_pmc.sh)"
  for k in gemm vit llm; do
    echo "== $k"
    python tools/pmc_kernels.py $O "$( [ $k = gemm ] && echo gemm_bf16 || echo attn_prefill )" 2>/dev/null | awk -v K=$k 'BEGIN{keep=0} {print}' > /dev/null
  done
} > /dev/null
python - "$O" <<'PY' > $O/mfma_busy.txt
import collections, csv, glob, sys
O = sys.argv[1]
def collect(prefix, want):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{O}/{prefix}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if want in r["Kernel_Name"]:
                rows[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for f in glob.glob(f"{O}/{prefix}_1/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if want in r["Kernel_Name"]:
                dur[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return rows, dur
print("# MFMA-busy / wait / cache counters per dispatch (rocprofv3 --pmc, separate passes; MI355X). tools/mfma_pmc.sh")
for prefix, want, what in (("gemm", "gemm_bf16", "LLM gate/up GEMM 2249 x 37888 x 3584 (tools/gemm_bench.py)"),
                           ("vit", "attn_", "ViT attention 4900 x 16 heads x d80 (tools/attn_bench.py --which vit): attn_vit32_kernel"),
                           ("llm", "attn_", "LLM causal attention S=2249 28/4 heads d128, unpaired + paired kernels")):
    rows, dur = collect(prefix, want)
    print(f"\n## {what}")
    for k, cs in rows.items():
        d = dur.get(k, [])
        med = sorted(d)[len(d) // 2] if d else 0
        print(f"{k}: median duration under the counter pass {med / 1e3:.1f} us")
        avg = {c: sum(v) / len(v) for c, v in cs.items()}
        for c in sorted(avg):
            print(f"    {c:34s} {avg[c]:18.1f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "SQ_BUSY_CU_CYCLES" in avg and avg["SQ_BUSY_CU_CYCLES"]:
            print(f"    -> MFMA busy = MFMA_BUSY_CYCLES / (BUSY_CU_CYCLES x 4 SIMD) = {avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * avg['SQ_BUSY_CU_CYCLES']):.3f}")
        if "GRBM_GUI_ACTIVE" in avg and med:
            print(f"    -> effective clock = GRBM_GUI_ACTIVE / 8 / duration = {avg['GRBM_GUI_ACTIVE'] / 8 / (med * 1e-9) / 1e9:.2f} GHz")
        if "TCC_HIT_sum" in avg:
            print(f"    -> L2 hit rate = {avg['TCC_HIT_sum'] / max(1.0, avg['TCC_HIT_sum'] + avg['TCC_MISS_sum']):.3f}")
PY
cat $O/mfma_busy.txt
find $O -name "*.csv" -size +2M -delete
