#!/bin/bash
# PMC passes over the headline OI call (one counter group per pass, as the MI355X guide prescribes).  $1 = extra env
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_IFETCH" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_WAVES"; do
  rm -rf /tmp/pmc
  env $1 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc -- python /root/repo/tools/oi_stats.py > /dev/null 2>&1
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:40], r["Counter_Name"])
    acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
for (kn, cn), (n, v) in sorted(acc.items()):
    if "k_oi" in kn: print(f"{kn:42s} {cn:24s} per-launch {v / n:16.0f}")
PY
done
