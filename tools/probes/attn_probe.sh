#!/bin/bash
# Build csrc/attn_prefill.hip with each timing probe (ATT_PROBE bits, see the source) into tools/probes/libattn_probe_<bits>.so
# and time the ViT (d = 80, 4900 x 16 heads) and LLM (d = 128 causal) shapes with each: python tools/probes/attn_probe.py
set -e
cd "$(dirname "$0")"
for b in 0 1 2 4 8 16 3 6 7; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DATT_PROBE=$b -I../../vision-inspection-system_amd/csrc \
      -o libattn_probe_$b.so ../../vision-inspection-system_amd/csrc/attn_prefill.hip
done
