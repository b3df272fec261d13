#!/bin/bash
# PMC passes (one counter group per pass) over any command: tools/pmc_cmd.sh "<kernel name filter>" <command...>
filter=$1; shift
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"; do
  rm -rf /tmp/pmc
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc -- "$@" > /dev/null 2>&1
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
  test -n "$f" && python - "$f" "$filter" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:48], r["Counter_Name"])
    acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
for (kn, cn), (n, v) in sorted(acc.items()):
    if sys.argv[2] in kn: print(f"{kn:50s} {cn:24s} launches {n:3d} per-launch {v / n:16.0f}")
PY
done
