#!/bin/bash
# rocprofv3 --kernel-trace --stats of any command, top kernels only: tools/kstats.sh <command...>
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- "$@" > /dev/null 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
test -n "$f" && python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("%-70s calls %4s avg %12.1f us  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
