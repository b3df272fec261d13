#!/bin/bash
# tools/isa.sh <unit> : device assembly of gridpp_amd/csrc/<unit>.hip with the product flags -> /tmp/isa/<unit>.s
set -e
mkdir -p /tmp/isa
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
    -fno-strict-aliasing --cuda-device-only -S gridpp_amd/csrc/$1.hip -o /tmp/isa/$1.s $GPP_HIP_DEFS
echo /tmp/isa/$1.s
