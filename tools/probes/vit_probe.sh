#!/bin/bash
# Build tools/probes/attn_vit_variants.hip (the experimental ViT attention kernels + the product source it includes) with
# each timing probe (VIT_PROBE bits, see the source) into tools/probes/libvit_probe_<bits>.so; time them on the GPU box
# with python tools/probes/vit_probe.py
set -e
cd "$(dirname "$0")"
build() {
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DVIT_PROBE=$1 -I../../vision-inspection-system_amd/csrc \
      -o libvit_probe_$1.so attn_vit_variants.hip
}
for b in ${@:-0 1 2 4 8 12 16 28 29 32}; do build $b & if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi; done
wait
