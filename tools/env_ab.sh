#!/bin/bash
# Same-box A/B of an environment switch (boxes differ by several % in clock): alternates VAR=a / VAR=b under bench.py,
# 3 rounds each.  Run ON the GPU box:  bash tools/env_ab.sh VIS_DECODE_CHAIN 0 1 [bench args]
set -e
cd "$GRAFT_REPO_ROOT"
VAR=$1; A=$2; B=$3; shift 3
for round in 1 2 3; do
  for val in $A $B; do
    env $VAR=$val python bench.py --no-extras --no-blocks --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$val', 'images/s %.4f' % d['value'], 'prefill ms %.3f' % d['prefill_mfma']['ms'], 'decode ms/token %.4f' % d['decode']['ms_per_token'])"
  done
done
