#!/bin/bash
# k_box_march (neighbourhood Mean of one 4000 x 4000 plane, halfwidth 15) with its loads (bm1), its stores (bm2) or both (bm3) switched off:
# variants built by tools/variant.sh NAME neighbourhood -DBM_ABL=n
for lib in product "$@"; do
  if [ "$lib" = product ]; then unset GPP_LIB; else export GPP_LIB=$PWD/gridpp_amd/lib/var_$lib.so; fi
  echo -n "$lib: "; bash tools/kstats.sh python $PWD/tools/box2d_time.py 2>/dev/null | grep "k_box" | awk '{print $1, $(NF-3), $(NF-2)}' | tr '\n' ' '; echo
done
