#!/bin/bash
# tools/kres.sh <unit> [name filter]: registers / scratch / occupancy of the kernels of gridpp_amd/csrc/<unit>.hip (product flags)
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
    -fno-strict-aliasing -Rpass-analysis=kernel-resource-usage --cuda-device-only -c gridpp_amd/csrc/$1.hip -o /tmp/kres_$1.o $GPP_HIP_DEFS 2>&1 |
  awk -v f="${2:-.}" '/Function Name/ {name=$NF; sub(/\[.*/,"",name); n=$0; sub(/.*Function Name: /,"",n); sub(/ \[.*/,"",n); show = (n ~ f)} 
       show && /(VGPRs:|SGPRs Spill|VGPRs Spill|ScratchSize|Occupancy)/ { line=$0; sub(/.*remark: +/,"",line); sub(/ \[-Rpass.*/,"",line); printf "%s  %s\n", n, line }'
