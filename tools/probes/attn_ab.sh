#!/bin/bash
# Same-box A/B of attention builds tools/probes/libvis_att{A,B,C,D}.so under tools/attn_bench.py (run ON the GPU box)
cd "$GRAFT_REPO_ROOT"
L=vision-inspection-system_amd/csrc/libvis_hip.so
cp $L /tmp/libvis_keep.so
for round in 1 2 3; do
  for v in ${VARIANTS:-A B C D}; do
    cp tools/probes/libvis_att$v.so $L
    echo "$v vit $(python tools/attn_bench.py --which vit --reps 30 2>/dev/null | grep '^vit')  $(python tools/attn_bench.py --which llm --reps 30 2>/dev/null | grep 'paired')"
  done
done
cp /tmp/libvis_keep.so $L
