#!/bin/bash
# same-box A/B of tools/probes/libvis_old.so against the product library on the batched-decode stream kernel
cd "$GRAFT_REPO_ROOT"
L=vision-inspection-system_amd/csrc/libvis_hip.so
cp $L /tmp/libvis_keep.so
for r in 1 2; do
  for v in old new; do
    if [ $v = new ]; then cp /tmp/libvis_keep.so $L; else cp tools/probes/libvis_old.so $L; fi
    for b in ${BATCHES:-64 16}; do echo "== $v"; python tools/probes/dec_depth_bench.py $b 2>/dev/null; done
  done
done
cp /tmp/libvis_keep.so $L
