#!/bin/bash
# Build csrc/gemm_bf16.hip with the workgroup timeline probe (-DGEMM_PROBE=<v>) into tools/probes/libgemm_probe_<v>.so;
# python tools/probes/gemm_probe.py prints where a 256x256 ping-pong tile's time goes (dispatch skew, prologue, loop, epilogue).
#   1 = timeline only; 2 = every workgroup stores to tile (0,0) (no HBM write traffic); 3 = non-temporal C stores; 4 = no C stores
set -e
cd "$(dirname "$0")"
for v in 1 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DGEMM_PROBE=$v -I../../vision-inspection-system_amd/csrc \
      -o libgemm_probe_$v.so ../../vision-inspection-system_amd/csrc/gemm_bf16.hip &
done
wait
