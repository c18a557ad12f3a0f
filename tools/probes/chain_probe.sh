#!/bin/bash
# Build the decode kernels with the chained layer head's workgroup timeline probe (-DCHAIN_PROBE) into
# tools/probes/libchain_probe.so; python tools/probes/chain_probe.py prints where the launch's time goes per role.
set -e
cd "$(dirname "$0")"
C=../../vision-inspection-system_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DCHAIN_PROBE -I$C -o libchain_probe.so $C/decode_chain.hip
