#!/bin/bash
# tools/hostile/build.sh: the poison helpers of the hostile soak (tools/oi_hostile_soak.py) and the -DGPP_POISON variant of the library
set -e
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -fPIC -shared tools/hostile/poison.hip -o tools/hostile/libpoison.so
bash tools/variant.sh poison oi -DGPP_POISON
echo tools/hostile/libpoison.so
