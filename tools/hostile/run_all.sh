#!/bin/bash
# tools/hostile/run_all.sh [REPEATS=20] [OUT=gpurun_out/r06] [TAG=r06]: every hostile soak with LDS, registers and all call-to-call workspaces poisoned
# before each call (needs tools/hostile/build.sh; the oracle's answers come from build/hostile_cache when they were computed before)
cd "$(dirname "$0")/../.."
R=${1:-20}; OUT=${2:-gpurun_out/r06}; TAG=${3:-r06}; mkdir -p $OUT
rc=0
python tools/nbh_hostile_soak.py 0 160 $R poison        > $OUT/${TAG}_nbh_hostile_soak.txt 2>&1        || rc=1
python tools/ensi_multi_hostile_soak.py 0 150 $R poison > $OUT/${TAG}_ensi_multi_hostile_soak.txt 2>&1 || rc=1
python tools/ensi_hostile_soak.py 0 120 $R poison tile  > $OUT/${TAG}_ensi_tile_hostile_soak.txt 2>&1  || rc=1
python tools/ensi_hostile_soak.py 0 120 $R poison       > $OUT/${TAG}_ensi_hostile_soak.txt 2>&1       || rc=1
python tools/oi_hostile_soak.py 5000 5100 $R 62 poison  > $OUT/${TAG}_oi_hostile_soak_62.txt 2>&1      || rc=1
python tools/oi_hostile_soak.py 5000 5100 $R poison     > $OUT/${TAG}_oi_hostile_soak_32.txt 2>&1      || rc=1
python tools/oi_hostile_soak.py 5000 5100 $R poison reuse > $OUT/${TAG}_oi_hostile_soak_32_reuse.txt 2>&1 || rc=1
python tools/oi_hostile_soak.py 5000 5100 $R 62 poison reuse > $OUT/${TAG}_oi_hostile_soak_62_reuse.txt 2>&1 || rc=1
python tools/oi_hostile_soak.py 5000 5100 $R poison spatial > $OUT/${TAG}_oi_hostile_soak_32_spatial.txt 2>&1 || rc=1
python tools/oi_hostile_soak.py 5000 5100 $R poison pairs > $OUT/${TAG}_oi_hostile_soak_32_pairs.txt 2>&1 || rc=1
for f in $OUT/${TAG}_*hostile_soak*.txt; do echo "$f: $(grep -c '^pass' $f) passes, $(tail -n 30 $f | grep FAILURES)"; done
exit $rc
