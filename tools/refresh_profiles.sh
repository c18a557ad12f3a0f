#!/bin/bash
# Run ON the GPU box (via gpurun): rocprofv3 stats + the two PMC passes of the default bench command's HEADLINE step
# (--no-blocks: the configs[2]/[3]/[4] blocks of the default line are profiled separately), condensed
# into profiles/<round>_* by tools/summarize_profile.py.  Usage: bash tools/refresh_profiles.sh r01
# The PMC passes run eagerly (--no-graph) and for 4 tokens: counter collection on hipGraph replays crashes rocprofv3.
set -e
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$R
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --steps 3 --warmup 1 --no-blocks > $O/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o bench -- python3 bench.py --steps 1 --warmup 0 --new-tokens 4 --no-cpu-baseline --no-blocks --no-graph > $O/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o bench -- python3 bench.py --steps 1 --warmup 0 --new-tokens 4 --no-cpu-baseline --no-blocks --no-graph > $O/bench_write.log 2>&1
# the roofline kernel of the chained step is the <1, false> instance (gate/up + down); the lm_head is <1, true>
python tools/summarize_profile.py --round $R --stats $O/stats --fetch $O/fetch --write $O/write --bench-log $O/bench_stats.log --out $O/profiles \
    --traffic-kernel "gemv_bf16_kernel<1, false>"
# HBM traffic of the chained layer-head launch (algorithmic: 58.7 MB of weights + the KV rows of the context): what its polling costs
python tools/summarize_profile.py --round $R --fetch $O/fetch --write $O/write --out $O/profiles \
    --traffic-kernel "decode_chain_kernel" --traffic-name chain
ls -la $O/profiles
