#!/bin/bash
# Round-3 profile refresh, run ON the GPU box: bash tools/profile_r03.sh   (all outputs under gpurun_out/prof_r03/profiles)
#  1. headline step: rocprofv3 --kernel-trace --stats + FETCH_SIZE / WRITE_SIZE passes of gemv_bf16_kernel
#  2. batch 64: FETCH_SIZE / WRITE_SIZE passes of gemm_decode_stream_kernel (the batch blocks' roofline.traffic)
#  3. MFMA-busy / wait / LDS / L2 counters of the ping-pong GEMM, the ViT attention kernel and the causal kernels
#  4. per-image prompt-pass breakdown (kernel trace of a --no-extras run)
#  5. decode attention counters (corrected counter sets)
R=r03
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$R
mkdir -p $O/profiles
bash tools/refresh_profiles.sh $R > $O/refresh.log 2>&1 || echo "refresh_profiles failed" >> $O/errors.txt
B64="bench.py --batch 64 --prompt-order text-first --steps 1 --warmup 0 --new-tokens 4 --no-extras --no-cpu-baseline --no-graph"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/b64_fetch -o bench -- python3 $B64 > $O/b64_fetch.log 2>&1 || echo "b64 fetch failed" >> $O/errors.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/b64_write -o bench -- python3 $B64 > $O/b64_write.log 2>&1 || echo "b64 write failed" >> $O/errors.txt
python3 tools/summarize_profile.py --round $R --fetch $O/b64_fetch --write $O/b64_write --out $O/profiles \
    --traffic-kernel gemm_decode_stream_kernel --traffic-name decode_stream || echo "b64 summary failed" >> $O/errors.txt
bash tools/mfma_pmc.sh $R > $O/mfma.log 2>&1; cp gpurun_out/pmc_$R/mfma_busy.txt $O/profiles/${R}_mfma_busy_pmc.txt
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o bench -- python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --new-tokens 2 > $O/trace.log 2>&1
python3 tools/prefill_breakdown.py $O/trace 4 $O/profiles/${R}_prefill_breakdown.csv > $O/breakdown.log 2>> $O/errors.txt   # 1 warm-up + 3 timed prompt passes
bash tools/decode_attn_pmc.sh 1 > $O/dattn1.log 2>&1; cp gpurun_out/dattn_pmc/summary.txt $O/profiles/${R}_decode_attn_pmc_b1.txt
bash tools/decode_attn_pmc.sh 64 > $O/dattn64.log 2>&1; cp gpurun_out/dattn_pmc/summary.txt $O/profiles/${R}_decode_attn_pmc_b64.txt
find gpurun_out -name "*.csv" -size +2M -delete
ls -la $O/profiles; cat $O/errors.txt 2>/dev/null
