#!/bin/bash
# Round-4 evidence (run on the GPU box from the repo root; outputs under gpurun_out/prof_r04/; copy them to profiles/ as r04_* and
# hbm_traffic.json):
#   1. python bench.py (default: the driver's N = 1 line with other_configs and the CPU baseline)
#   2. rocprofv3 --kernel-trace --stats of the OI bench, of C5 (EnSI) and of C4 (neighbourhood Mean / quantile_fast)
#   3. separate --pmc passes of the OI bench (FETCH_SIZE, WRITE_SIZE, SQ instruction mix), as the MI355X guide prescribes
#   4. --pmc FETCH_SIZE / WRITE_SIZE of C4 (quantile_fast traffic) and SQ_INSTS_VALU / SQ_INSTS_LDS / SQ_INSTS_VALU_MFMA_F64? of C5
set -u
ONLY=${PROFILE_ONLY:-all}
REPO=${GRAFT_REPO_ROOT:-$PWD}
TAG=${PROFILE_TAG:-r04}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ "$ONLY" = all ] && python $REPO/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
stats() {   # name, command...
  local name=$1; shift
  rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- "$@" > /dev/null 2>&1
  local f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
  test -n "$f" && python - "$f" "$OUT/${name}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as f:
    f.write("kernel,calls,total_ns,average_ns,percentage,min_ns,max_ns\n")
    for r in rows[:14]:
        n = r["Name"]
        if n.startswith("void at::") or "elementwise" in n: continue
        f.write('"%s",%s,%s,%s,%s,%s,%s\n' % (n[:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]))
PY
}
pmc() {     # name, filter, counters, command...
  local name=$1 filt=$2 ctr=$3; shift 3
  rm -rf /tmp/pmc && rocprofv3 --pmc $ctr --output-format csv -d /tmp/pmc -- "$@" > /dev/null 2>&1
  local f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
  test -n "$f" && python - "$f" "$OUT/${name}.csv" "$filt" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"], r["Grid_Size"], r["Counter_Name"])
    acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
with open(sys.argv[2], "w") as f:
    f.write("kernel,grid_size,counter,launches,mean_per_launch\n")
    for (kn, gs, cn), (n, v) in sorted(acc.items()):
        if any(t in kn for t in sys.argv[3].split("|")): f.write('"%s",%s,%s,%d,%.1f\n' % (kn[:90], gs, cn, n, v / n))
PY
}
# PROFILE_ONLY=oi: the OI passes only (kernel iteration); ensi: + the EnSI passes; noise: + the white-noise variant; nbh: the neighbourhood passes alone; default: everything

OI="python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs"
[ "$ONLY" != nbh ] && stats oi $OI
[ "$ONLY" = all -o "$ONLY" = ensi ] && stats ensi python $REPO/tools/ensi_c5.py
[ "$ONLY" = all -o "$ONLY" = nbh ] && stats nbh python $REPO/tools/bench_paths.py nb
[ "$ONLY" != nbh ] && pmc oi_pmc_fetch "k_oi" "FETCH_SIZE" $OI
[ "$ONLY" != nbh ] && pmc oi_pmc_write "k_oi" "WRITE_SIZE" $OI
[ "$ONLY" != nbh ] && pmc oi_pmc_sq "k_oi" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" $OI
[ "$ONLY" = all -o "$ONLY" = nbh ] && pmc nbh_pmc_fetch "k_member|k_qf|k_box" "FETCH_SIZE" python $REPO/tools/bench_paths.py nb
[ "$ONLY" = all -o "$ONLY" = nbh ] && pmc nbh_pmc_write "k_member|k_qf|k_box" "WRITE_SIZE" python $REPO/tools/bench_paths.py nb
[ "$ONLY" != nbh ] && pmc oi_pmc_fp64 "k_oi" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" $OI
# where the headline kernel's cycles go (round-3 verdict, item 3): busy / active / issue-stalled wave cycles, then what the waits are for
[ "$ONLY" != nbh ] && pmc oi_pmc_busy "k_oi" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" $OI
[ "$ONLY" != nbh ] && pmc oi_pmc_wait "k_oi" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" $OI
[ "$ONLY" != nbh ] && pmc oi_pmc_act "k_oi" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" $OI
[ "$ONLY" = all -o "$ONLY" = ensi ] && pmc ensi_pmc_sq "k_ensi" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F64" python $REPO/tools/ensi_c5.py
[ "$ONLY" = all -o "$ONLY" = ensi ] && pmc ensi_pmc_fp64 "k_ensi" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" python $REPO/tools/ensi_c5.py
[ "$ONLY" = all -o "$ONLY" = ensi ] && pmc ensi_pmc_busy "k_ensi" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" python $REPO/tools/ensi_c5.py
[ "$ONLY" = all -o "$ONLY" = ensi ] && pmc ensi_pmc_mfma "k_ensi" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" python $REPO/tools/ensi_c5.py
[ "$ONLY" = all -o "$ONLY" = nbh ] && pmc nbh_pmc_sq "k_qf|k_member" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" python $REPO/tools/prof_nb.py
[ "$ONLY" = all -o "$ONLY" = nbh ] && pmc nbh_pmc_busy "k_qf|k_member" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" python $REPO/tools/prof_nb.py
[ "$ONLY" = all ] && python $REPO/tools/oi_variants.py > $OUT/oi_variants.jsonl 2>/dev/null
# round 5: the white-noise terrain variant of config 3 (k_oi scans and parks, k_oi_pairs solves): kernel times and what the two kernels execute
if [ "$ONLY" = all -o "$ONLY" = noise ]; then
  stats noise python $REPO/tools/oi_noise.py 4000
  pmc noise_pmc_sq "k_oi" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" python $REPO/tools/oi_noise.py 4000
  pmc noise_pmc_fp64 "k_oi" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" python $REPO/tools/oi_noise.py 4000
  pmc noise_pmc_busy "k_oi" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" python $REPO/tools/oi_noise.py 4000
fi
python $REPO/tools/fold_profiles.py $OUT > $OUT/hbm_traffic.json
ls -la $OUT
cut -c1-300 $OUT/bench_n1.json | head -2
