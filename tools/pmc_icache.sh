#!/bin/bash
# Instruction-cache counters of any command (one pass): tools/pmc_icache.sh "<kernel name filter>" <command...>
filter=$1; shift
cd /tmp && export TMPDIR=/tmp
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
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
    if sys.argv[2] in kn: print(f"{kn:50s} {cn:28s} launches {n:3d} per-launch {v / n:16.0f}")
PY
done
