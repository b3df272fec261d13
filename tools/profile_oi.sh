#!/bin/bash
# rocprofv3 evidence for the headline OI bench line (run on the GPU box from the repo root):
#   1. --kernel-trace --stats of `python bench.py --steps 5 --warmup 2 --no-cpu-baseline`
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ instruction mix) of the same command, as the MI355X guide prescribes
# Outputs land in gpurun_out/prof_oi/ ; the summaries are copied to profiles/ by hand (see profiles/README in DESIGN.md 6).
set -u
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/prof_oi
mkdir -p $OUT
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- $CMD > $OUT/bench_under_rocprof.json 2>/dev/null
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
$CMD > $OUT/bench_plain.json 2>/dev/null
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES"; do
  rm -rf /tmp/pmc
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc -- $CMD > /dev/null 2>&1
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  python - "$f" "$OUT/pmc_$tag.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"], r["Grid_Size"], r["Counter_Name"])
    acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
with open(sys.argv[2], "w") as f:
    f.write("kernel,grid_size,counter,launches,mean_per_launch\n")
    for (kn, gs, cn), (n, v) in sorted(acc.items()):
        if "k_oi" in kn or "k_pack" in kn: f.write('"%s",%s,%s,%d,%.1f\n' % (kn, gs, cn, n, v / n))
PY
done
cat $OUT/pmc_*.csv | grep -v "^kernel"
head -6 $OUT/kernel_stats.csv
cat $OUT/bench_plain.json | cut -c1-400
