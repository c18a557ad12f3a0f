#!/bin/bash
# Round-5 decode-step tables, run ON the GPU box: per-(kernel, grid) durations of the batched decode step at 64 sequences (bf16)
# and 4 sequences (fp8), eager launches.  Outputs under gpurun_out/$1 (default prof_r05_dec).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-prof_r05_dec}
mkdir -p $O
CYC=$(python3 -c "print(','.join(['qkv','o','gateup','down']*28+['lm_head']))")
COMMON="--prompt-order text-first --steps 1 --warmup 0 --new-tokens 12 --no-extras --no-blocks --no-cpu-baseline --no-graph"
rocprofv3 --kernel-trace --output-format csv -d $O/b64 -o t -- python3 bench.py --batch 64 $COMMON > $O/b64.log 2>&1 || echo "b64 failed" >> $O/errors.txt
python3 tools/trace_by_grid.py $O/b64 decode_proj decode_colpar decode_prep gemm_decode skinny_ decode_attn argmax gather_rows norm_rows > $O/b64_decode_by_grid.txt 2>> $O/errors.txt
rocprofv3 --kernel-trace --output-format csv -d $O/f8b4 -o t -- python3 bench.py --batch 4 --prefill-dtype fp8 --decode-weights fp8 $COMMON > $O/f8b4.log 2>&1 || echo "f8b4 failed" >> $O/errors.txt
python3 tools/trace_by_grid.py $O/f8b4 decode_proj decode_colpar decode_prep gemm_decode skinny_ decode_attn argmax gather_rows quant_rows norm_rows > $O/f8b4_decode_by_grid.txt 2>> $O/errors.txt
find $O -name "*.csv" -size +2M -delete
cat $O/b64_decode_by_grid.txt $O/f8b4_decode_by_grid.txt; cat $O/errors.txt 2>/dev/null; true
