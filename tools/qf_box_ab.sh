# quantile_fast (config 4) with the box pass's two forms, alternating on one box: product (E-fold sums where interpolate() looks, round 6) and
# var_nolazy (the round-5 form: every threshold's E-fold sum; tools/variant.sh nolazy qf_box -DQB_LAZY=0)
for rep in 1 2 3; do for lib in product var_nolazy; do
  if [ $lib = product ]; then unset GPP_LIB; else export GPP_LIB=gridpp_amd/lib/$lib.so; fi
  echo "$lib: $(timeout 200 python tools/qf_q_sweep.py 0.5 0.55 2>&1 | grep "q =" | tr '\n' ' ')"
done; done
