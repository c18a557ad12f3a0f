#!/bin/bash
# Round-4 profile refresh, run ON the GPU box: bash tools/profile_r04.sh [part ...]   (outputs under gpurun_out/prof_r04/profiles)
# parts (default: all):
#  head    headline step: rocprofv3 --kernel-trace --stats + FETCH_SIZE / WRITE_SIZE passes of gemv_bf16_kernel, bf16 prompt-pass table
#  fp8     the fp8 configuration (configs[4] slice, 4 images per GPU): prompt-pass table, FETCH / WRITE passes of
#          gemm_decode_stream_kernel<fp8>, MFMA-busy counters of the fp8 ping-pong GEMM
#  mllama  Llama-3.2-11B-Vision (the Auditor): kernel stats of the single-image run, prompt-pass table at 32 images per step
#  dual    configs[2] at 32 images per step: kernel stats
#  b64     FETCH / WRITE passes of gemm_decode_stream_kernel at 64 sequences (batch64 block's roofline.traffic)
R=r04
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$R
P=$O/profiles
mkdir -p $P
PARTS="${@:-head fp8 mllama dual b64}"
has() { [[ " $PARTS " == *" $1 "* ]]; }
fail() { echo "$1 failed" >> $O/errors.txt; }
trace() {  # name, program args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o t -- python3 "$@" > $O/$name.log 2>&1 || fail "trace $name"
}
pmc() {    # name, counters, program args...
  local name=$1; shift; local ctr=$1; shift
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/$name -o p -- python3 "$@" > $O/$name.log 2>&1 || fail "pmc $name"
}
if has head; then
  bash tools/refresh_profiles.sh $R > $O/refresh.log 2>&1 || fail refresh_profiles
  cp gpurun_out/prof_$R/profiles/* $P/ 2>/dev/null
  trace pre16 bench.py --steps 3 --warmup 1 --no-extras --no-blocks --no-cpu-baseline --new-tokens 2
  python3 tools/prefill_breakdown.py $O/pre16 4 $P/${R}_prefill_breakdown.csv > $O/pre16_table.log 2>> $O/errors.txt
  echo "head done"; date
fi
F8="bench.py --batch 4 --prompt-order text-first --prefill-dtype fp8 --decode-weights fp8 --no-extras --no-blocks --no-cpu-baseline"
if has fp8; then
  trace pre8 $F8 --steps 2 --warmup 1 --new-tokens 2
  python3 tools/trace_breakdown.py $O/pre8 12 $P/${R}_prefill_breakdown_fp8.csv \
      --exclude "at::native,__amd_rocclr,Custom_Cijk,gemm_decode,skinny_,decode_attn,argmax_,gemv_" > $O/pre8_table.log 2>> $O/errors.txt
  pmc f8_fetch FETCH_SIZE $F8 --steps 1 --warmup 0 --new-tokens 4 --no-graph
  pmc f8_write WRITE_SIZE $F8 --steps 1 --warmup 0 --new-tokens 4 --no-graph
  python3 tools/summarize_profile.py --round $R --fetch $O/f8_fetch --write $O/f8_write --out $P \
      --traffic-kernel "gemm_decode_stream_kernel<true" --traffic-name decode_stream_fp8 || fail "fp8 traffic summary"
  i=0
  for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    VIS_FP8=1 VIS_NOCHECK=1 pmc g8_$i "$C" tools/gemm_bench.py 5156 37888 3584 3
  done
  python3 tools/pmc_kernels.py $O gemm_fp8 > $P/${R}_fp8_gemm_pmc.txt 2>> $O/errors.txt
  python3 tools/pmc_kernels.py $O quant_rows >> $P/${R}_fp8_gemm_pmc.txt 2>> $O/errors.txt
  echo "fp8 done"; date
fi
if has mllama; then
  trace ml1 tools/mllama_bench.py --steps 2
  python3 tools/summarize_profile.py --round ${R}_mllama --stats $O/ml1 --out $P || fail "mllama stats"
  rm -f $P/${R}_mllama_gemv_by_shape.csv
  trace ml32 tools/mllama_bench.py --batch 32 --steps 1 --new-tokens 2
  python3 tools/trace_breakdown.py $O/ml32 64 $P/${R}_mllama_prefill_breakdown.csv \
      --exclude "at::native,__amd_rocclr,Custom_Cijk,gemm_decode,skinny_,decode_attn,argmax_,gemv_" > $O/ml32_table.log 2>> $O/errors.txt
  echo "mllama done"; date
fi
if has dual; then
  trace dual tools/dual_bench.py --batch 32 --steps 1
  python3 tools/summarize_profile.py --round ${R}_dual --stats $O/dual --out $P || fail "dual stats"
  rm -f $P/${R}_dual_gemv_by_shape.csv
  echo "dual done"; date
fi
if has b64; then
  B64="bench.py --batch 64 --prompt-order text-first --steps 1 --warmup 0 --new-tokens 4 --no-extras --no-blocks --no-cpu-baseline --no-graph"
  pmc b64_fetch FETCH_SIZE $B64
  pmc b64_write WRITE_SIZE $B64
  python3 tools/summarize_profile.py --round $R --fetch $O/b64_fetch --write $O/b64_write --out $P \
      --traffic-kernel gemm_decode_stream_kernel --traffic-name decode_stream || fail "b64 summary"
  echo "b64 done"; date
fi
find gpurun_out -name "*.csv" -size +2M -delete
ls -la $P; cat $O/errors.txt 2>/dev/null
