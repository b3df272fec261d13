#!/bin/bash
# C5 EnSI time as a function of the park size (GPP_ENSI_PARK_MB): does keeping the park in the last-level cache pay?
cd "$(dirname "$0")/.."
for mb in 128 256 512 1024 4096 16384 0; do
  if [ "$mb" = 0 ]; then unset GPP_ENSI_PARK_MB; else export GPP_ENSI_PARK_MB=$mb; fi
  echo "park_mb=$mb $(python tools/ensi_c5.py | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms'],1), 'ms')")"
done
