#!/bin/bash
# small-batch decode: the multi-row GEMV path (VIS_ROWS_GEMV=<largest batch>, here 4) against the stream-K path (0),
# bf16 and fp8 weights: bash tools/probes/rows_ab.sh <batch>
cd "$GRAFT_REPO_ROOT"
for rg in 0 4; do for dt in bf16 fp8; do
  if [ $dt = fp8 ]; then X="--prefill-dtype fp8 --decode-weights fp8"; else X=""; fi
  VIS_ROWS_GEMV=$rg python bench.py --batch ${1:-4} --prompt-order text-first --steps 3 --warmup 1 --no-extras --no-cpu-baseline $X 2>/dev/null > /tmp/rows_ab.json
  python - <<PY
import json
d=json.loads(open("/tmp/rows_ab.json").read().strip().splitlines()[-1])
print("rows=$rg $dt images/s %.3f ms/step %.1f" % (d["value"], d["ms_per_step"]), {k: (round(v,3) if isinstance(v,float) else v) for k,v in d.get("decode",{}).items()} if isinstance(d.get("decode"),dict) else "")
PY
done; done
