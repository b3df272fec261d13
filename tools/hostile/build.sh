#!/bin/bash
# tools/hostile/build.sh: the poison helpers of the hostile soaks (tools/*_hostile_soak.py) and the -DGPP_POISON variant of the WHOLE
# library (every translation unit: DevBuf poisons fresh allocations, and each kernel family exports its gpp_debug_poison_*_workspace)
# -> gridpp_amd/lib/var_poison.so, selected with GPP_LIB=...; the product library is never replaced by it.
set -e
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -fPIC -shared tools/hostile/poison.hip -o tools/hostile/libpoison.so
mkdir -p build/var/poison
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-strict-aliasing -DGPP_POISON"
pids=()
for src in gridpp_amd/csrc/*.hip; do
    unit=$(basename "$src" .hip)
    obj=build/var/poison/$unit.o
    if [ ! -f "$obj" ] || [ -n "$(find gridpp_amd/csrc include "$0" -newer "$obj" \( -name '*.hip' -o -name '*.h' -o -name 'build.sh' \) | head -1)" ]; then
        extra=$(head -40 "$src" | grep '^// hipcc-flags:' | cut -d: -f2-)
        ( /opt/rocm/bin/hipcc $FLAGS $extra -c "$src" -o "$obj" 2>&1 | grep -E "error" -A3 || true ) &
        pids+=($!)
    fi
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/var/poison/*.o -o gridpp_amd/lib/var_poison.so
echo tools/hostile/libpoison.so gridpp_amd/lib/var_poison.so
