#!/bin/bash
# tools/ensi_order_sweep.sh: config-5 time of the EnSI path for the orders of the perturbation series (variant builds o33 / o43 / o34 / o44 of
# tools/variant.sh: square-root steps / Neumann products) against the stopping threshold of the Jacobi sweeps (GPP_ENSI_JTOL = |E| / c)
cd "$(dirname "$0")/.."
for v in o33 o43 o34 o44; do
  for t in 0.02 0.03 0.04 0.05 0.06; do
    ms=$(GPP_LIB=gridpp_amd/lib/var_$v.so GPP_ENSI_JTOL=$t python tools/ensi_c5.py 2>/dev/null | tail -1 | python -c "import json,sys; print('%.1f' % json.loads(sys.stdin.read())['ms'])")
    echo "$v jtol $t: $ms ms"
  done
done
