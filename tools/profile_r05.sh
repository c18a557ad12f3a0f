#!/bin/bash
# Round-5 profile refresh, run ON the GPU box: bash tools/profile_r05.sh [part ...]   (outputs under gpurun_out/prof_r05/profiles)
#  head    headline step: rocprofv3 --kernel-trace --stats + FETCH_SIZE / WRITE_SIZE passes of gemv_bf16_kernel, bf16 prompt-pass table
#  seam    kernel trace of run_batch_inspection over 64 PNG files (Inspector local, Auditor canned): GPU-idle time inside the
#          prompt-pass phase by gap size (tools/seam_gaps.py) -> r05_seam64_gpu_gaps.json
#  dual    configs[2] at 32 images per step: kernel stats + the Auditor's decode step by kernel (r05_dual_decode_step.csv)
#  fp8     fp8 prompt-pass table (4 images per step)
#  b64     per-image prompt-pass table at 64 images per step (r05_prefill_breakdown_b64.csv)
#  traffic FETCH_SIZE / WRITE_SIZE passes of the batched decode projection at 64 sequences: the default pair's stream kernel
#          (r05_decode_stream_traffic.json) and the opt-in single-launch stream-K form (r05_decode_proj_fused_traffic.json)
R=r05
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$R
P=$O/profiles
mkdir -p $P
PARTS="${@:-head seam dual fp8 b64 traffic}"
has() { [[ " $PARTS " == *" $1 "* ]]; }
fail() { echo "$1 failed" >> $O/errors.txt; }
trace() {  # name, program args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o t -- python3 "$@" > $O/$name.log 2>&1 || fail "trace $name"
}
if has head; then
  bash tools/refresh_profiles.sh $R > $O/refresh.log 2>&1 || fail refresh_profiles
  cp gpurun_out/prof_$R/profiles/* $P/ 2>/dev/null
  trace pre16 bench.py --steps 3 --warmup 1 --no-extras --no-blocks --no-cpu-baseline --new-tokens 2
  python3 tools/prefill_breakdown.py $O/pre16 4 $P/${R}_prefill_breakdown.csv > $O/pre16_table.log 2>> $O/errors.txt
  echo "head done"; date
fi
if has seam; then
  trace seam tools/ingest_bench.py --images 64 --auditor mock --threads 4
  python3 tools/seam_gaps.py $O/seam 64 $P/${R}_seam64_gpu_gaps.json > $O/seam_gaps.log 2>> $O/errors.txt
  grep -h "^{" $O/seam.log | tail -1 > $P/${R}_seam64_under_rocprof.json
  echo "seam done"; date
fi
if has dual; then
  trace dual tools/dual_bench.py --batch 32 --steps 1
  python3 tools/summarize_profile.py --round ${R}_dual --stats $O/dual --out $P || fail "dual stats"
  rm -f $P/${R}_dual_gemv_by_shape.csv
  python3 tools/trace_by_grid.py $O/dual gemm_decode skinny_ decode_attn decode_cross argmax gather_rows norm_rows decode_proj decode_colpar > $P/${R}_dual_decode_step.txt 2>> $O/errors.txt
  echo "dual done"; date
fi
F8="bench.py --batch 4 --prompt-order text-first --prefill-dtype fp8 --decode-weights fp8 --no-extras --no-blocks --no-cpu-baseline"
if has fp8; then
  trace pre8 $F8 --steps 2 --warmup 1 --new-tokens 2
  python3 tools/trace_breakdown.py $O/pre8 12 $P/${R}_prefill_breakdown_fp8.csv \
      --exclude "at::native,__amd_rocclr,Custom_Cijk,gemm_decode,skinny_,decode_attn,argmax_,gemv_,decode_proj,decode_colpar,decode_prep" > $O/pre8_table.log 2>> $O/errors.txt
  echo "fp8 done"; date
fi
if has b64; then
  trace pre64 bench.py --batch 64 --prompt-order text-first --steps 1 --warmup 1 --new-tokens 2 --no-extras --no-blocks --no-cpu-baseline
  python3 tools/trace_breakdown.py $O/pre64 128 $P/${R}_prefill_breakdown_b64.csv \
      --exclude "at::native,__amd_rocclr,Custom_Cijk,gemm_decode,skinny_,decode_attn,argmax_,gemv_,decode_proj,decode_colpar,decode_prep" > $O/pre64_table.log 2>> $O/errors.txt
  echo "b64 done"; date
fi
if has traffic; then
  pmc() { local name=$1; shift; local ctr=$1; shift
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/$name -o p -- python3 "$@" > $O/$name.log 2>&1 || fail "pmc $name"; }
  B64="bench.py --batch 64 --prompt-order text-first --steps 1 --warmup 0 --new-tokens 4 --no-extras --no-blocks --no-cpu-baseline --no-graph"
  pmc b64_fetch FETCH_SIZE $B64
  pmc b64_write WRITE_SIZE $B64
  python3 tools/summarize_profile.py --round $R --fetch $O/b64_fetch --write $O/b64_write --out $P \
      --traffic-kernel gemm_decode_stream_kernel --traffic-name decode_stream || fail "b64 traffic summary"
  VIS_DECODE_FUSED=1 VIS_DOWN_PAIR=0 VIS_DECODE_PROJ_FORM=streamk pmc b64f_fetch FETCH_SIZE $B64
  VIS_DECODE_FUSED=1 VIS_DOWN_PAIR=0 VIS_DECODE_PROJ_FORM=streamk pmc b64f_write WRITE_SIZE $B64
  python3 tools/summarize_profile.py --round $R --fetch $O/b64f_fetch --write $O/b64f_write --out $P \
      --traffic-kernel decode_proj_kernel --traffic-name decode_proj_fused || fail "fused traffic summary"
  echo "traffic done"; date
fi
find gpurun_out -name "*.csv" -size +2M -delete
ls -la $P; cat $O/errors.txt 2>/dev/null; true
