#!/bin/bash
# tools/variant.sh <name> <unit> <defs...> : an A/B build of the library with one translation unit recompiled under extra
# defines -> gridpp_amd/lib/var_<name>.so (select it with GPP_LIB=...; diagnostic builds never replace the product library)
set -e
cd "$(dirname "$0")/.."
name=$1; unit=$2; shift 2
mkdir -p build/var/$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
    -fno-strict-aliasing "$@" -c gridpp_amd/csrc/$unit.hip -o build/var/$name/$unit.o 2>&1 | grep -E "error" -A3 || true
objs=$(ls build/obj/*.o | grep -v "/$unit.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build/var/$name/$unit.o -o gridpp_amd/lib/var_$name.so
echo gridpp_amd/lib/var_$name.so
