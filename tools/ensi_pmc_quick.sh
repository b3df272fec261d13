#!/bin/bash
# tools/ensi_pmc_quick.sh: what the three EnSI kernels of config 5 keep busy (two --pmc passes, per-kernel means) -> gpurun_out/r06/ensi_pmc_quick.txt
REPO=${GRAFT_REPO_ROOT:-$PWD}; O=$REPO/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/ensi_pmc_quick.txt
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA"; do
  rm -rf /tmp/pmcq && rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmcq -- python $REPO/tools/ensi_c5.py > /dev/null 2>&1
  f=$(find /tmp/pmcq -name "*counter_collection.csv" | head -1)
  python - "$f" >> $O/ensi_pmc_quick.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if "k_ensi" not in r["Kernel_Name"]: continue
    k = (r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])
    acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
for (kn, cn), (n, v) in sorted(acc.items()): print("%-42s %-28s launches %3d  total %.4g" % (kn, cn, n, v))
PY
done
cat $O/ensi_pmc_quick.txt
