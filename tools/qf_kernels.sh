#!/bin/bash
# kernel times of the C4 quantile_fast call for the product library and every gridpp_amd/lib/var_*.so
cd "$(dirname "$0")/.."
R=$PWD
for lib in gridpp_amd/lib/libgridpp_hip.so gridpp_amd/lib/var_*.so; do
    [ -f "$lib" ] || continue
    echo "== $lib"
    GPP_LIB=$R/$lib bash tools/kstats.sh python $R/tools/prof_nb.py | grep "k_qf\|k_member_pass<2>"
done
