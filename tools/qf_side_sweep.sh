#!/bin/bash
# Premise of an overlapped quantile_fast (round 5): the count pass (HBM-bound) with FEWER workgroups per CU (GPP_QF_WAVES; QC_G loads in flight per lane:
# variants side5 / side8 / side12 built by tools/variant.sh with -DQF_SIDE_EXPERIMENT) and the box pass (VALU-bound) of the whole field side by side on two
# streams with no dependence (timing only -- the box pass reads the planes of the previous call).  Prints ms per call of C4.
for lib in side5 side8 side12; do
  for w in 14 10 8 6 4; do
    for side in 0 1; do
      if [ $side = 1 ]; then export GPP_QF_SIDE=1; else unset GPP_QF_SIDE; fi
      echo -n "$lib waves=$w side=$side: "; GPP_QF_WAVES=$w GPP_LIB=$PWD/gridpp_amd/lib/var_$lib.so python tools/qf_c4.py 2>/dev/null | grep -o '"ms": [0-9.]*'
    done
  done
done
