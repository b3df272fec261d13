#!/bin/bash
# Counters of the first pass with one factorisation per tile (product) and per PAIR of tiles (GPP_OI_PAIR_TILES=1), one counter group per pass
# (as the MI355X guide prescribes), headline workload with the fields in HBM (tools/oi_stats_once.py: three calls) -> per-launch means on stdout.
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for mode in single pair; do
  [ $mode = pair ] && export GPP_OI_PAIR_TILES=1 || unset GPP_OI_PAIR_TILES
  echo "## $mode"
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
    rm -rf /tmp/pmcp
    rocprofv3 --pmc $grp --output-format csv -d /tmp/pmcp -- python $REPO/tools/oi_stats_once.py > /dev/null 2>&1
    f=$(find /tmp/pmcp -name "*counter_collection.csv" | head -1)
    test -n "$f" && python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if "k_oi_union" in r["Kernel_Name"] and ("false, 32" in r["Kernel_Name"] or "pair" in r["Kernel_Name"]):
        k = (r["Kernel_Name"].replace("void ", "")[:44], r["Counter_Name"])
        acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
for (kn, cn), (n, v) in sorted(acc.items()):
    print(f"{kn:46s} {cn:26s} launches {n:2d} per-launch {v / n:16.0f}")
PY
  done
done
