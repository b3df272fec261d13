#!/bin/bash
# Box pass of quantile_fast (k_qf_box<15, false>, C4) with parts switched off (variants qb1: no E-fold sums, qb2: no interpolation, qb4: no window
# sums, qb7: none of them) and with other unroll factors of the E-fold loop (qbu1 / qbu4 / qbu10); tools/variant.sh NAME qf_box -DQB_ABL=n | -DQB_UNR=n.
for lib in product "$@"; do
  if [ "$lib" = product ]; then unset GPP_LIB; else export GPP_LIB=$PWD/gridpp_amd/lib/var_$lib.so; fi
  echo -n "$lib: "; bash tools/kstats.sh python $PWD/tools/prof_nb.py 2>/dev/null | grep "k_qf_box<15, false>" | awk '{print $(NF-3), $(NF-2)}'
done
