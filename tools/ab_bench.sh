#!/bin/bash
# Same-box A/B of two builds of csrc/libvis_hip.so (boxes differ by several % in clock): alternates the product
# library and tools/probes/libvis_old.so under bench.py, 3 rounds each.  Run ON the GPU box: bash tools/ab_bench.sh [bench args]
set -e
cd "$GRAFT_REPO_ROOT"
L=vision-inspection-system_amd/csrc/libvis_hip.so
cp $L /tmp/libvis_new.so
for round in 1 2 3; do
  for which in old new; do
    if [ $which = old ]; then cp tools/probes/libvis_old.so $L; else cp /tmp/libvis_new.so $L; fi
    python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$which', 'images/s %.4f' % d['value'], 'prefill ms %.3f' % d['prefill_mfma']['ms'], 'frac %.4f' % d['prefill_mfma']['frac'], 'decode ms/token %.4f' % d['decode']['ms_per_token'])"
  done
done
cp /tmp/libvis_new.so $L
