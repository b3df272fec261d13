#!/bin/bash
# tools/qf_box_prof.sh [q]: kernel times and vector instructions of one quantile_fast configuration (config 4 cube) under GPP_LIB variants
REPO=${GRAFT_REPO_ROOT:-$PWD}; Q=${1:-0.55}
cd /tmp && export TMPDIR=/tmp
for lib in "" var_nolazy; do
  if [ -n "$lib" ]; then export GPP_LIB=$REPO/gridpp_amd/lib/$lib.so; else unset GPP_LIB; fi
  echo "== ${lib:-product} q=$Q"
  rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $REPO/tools/qf_q_sweep.py $Q > /dev/null 2>&1
  f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
  python -c "
import csv, sys
for r in csv.DictReader(open('$f')):
    if 'k_qf_box<15, false>' in r['Name'] or 'k_qf_count' in r['Name']: print('   %-40s calls %s  avg %.1f us  min %.1f' % (r['Name'].split('(')[-2][-40:] if False else r['Name'][:48], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
"
  rm -rf /tmp/pm && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES --output-format csv -d /tmp/pm -- python $REPO/tools/qf_q_sweep.py $Q > /dev/null 2>&1
  f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if "k_qf_box<15, false>" not in r["Kernel_Name"]: continue
    acc[r["Counter_Name"]][0] += 1; acc[r["Counter_Name"]][1] += float(r["Counter_Value"])
print("   k_qf_box<15,false> per launch:", {k: "%.3g" % (v[1] / v[0]) for k, v in sorted(acc.items())})
PY
done
