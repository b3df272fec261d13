#!/bin/bash
# tools/ab_bench.sh <steps> <lib...>: the headline bench for the product library and every listed variant on ONE box, alternating,
# three rounds (ms per step and the dominant kernel's HIP-event time).  bench.py refuses GPP_* overrides, so the variants are timed
# through tools/oi_time.py.
steps=$1; shift
for round in 1 2 3; do
  for lib in product "$@"; do
    if [ "$lib" = product ]; then unset GPP_LIB; else export GPP_LIB=$PWD/gridpp_amd/lib/var_$lib.so; fi
    python tools/oi_time.py $steps | sed "s/^/round $round $lib: /"
  done
done
