#!/bin/bash
# Round-6 evidence on the final library, one GPU call per part (gpurun_out/r06/ -> profiles/r06_*):
#   tools/final_r06.sh suite     the GPU suite, plain and poisoned
#   tools/final_r06.sh hostile   tools/hostile/run_all.sh 20
#   tools/final_r06.sh soaks     the randomised soaks against the oracle, the large-n and ill-conditioned EnSI tables
#   tools/final_r06.sh bench     slice overheads, the reference's benchmark rows, the default bench line
cd "$(dirname "$0")/.."
O=gpurun_out/r06; mkdir -p $O
noids() { grep -v amdgpu.ids; }
case "$1" in
suite)
  (echo "# python -m pytest tests -m gpu -q (final library of round 6)"; python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error") > $O/r06_pytest_gpu.txt
  (echo "# GRIDPP_TEST_POISON=1 GPP_LIB=gridpp_amd/lib/var_poison.so python -m pytest tests -m gpu -q: LDS, registers and every call-to-call HBM workspace filled with 0xFF before every library call";
   GPP_LIB=$PWD/gridpp_amd/lib/var_poison.so GRIDPP_TEST_POISON=1 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error") > $O/r06_pytest_gpu_poisoned.txt
  cat $O/r06_pytest_gpu.txt $O/r06_pytest_gpu_poisoned.txt ;;
hostile)
  bash tools/hostile/run_all.sh 20 $O r06 ;;
soaks)
  python tools/oi_stress_more.py 6000 6600 2>&1 | noids > $O/r06_oi_stress_soak.txt
  python tools/oi_stress_more.py 6000 6300 62 2>&1 | noids > $O/r06_oi_stress_soak_62.txt
  python tools/oi_rough_soak.py 0 200 2>&1 | noids > $O/r06_oi_rough_soak.txt
  python tools/oi_parked_soak.py 0 60 2>&1 | noids > $O/r06_oi_parked_soak.txt
  python tools/ensi_soak.py 240 2>&1 | noids > $O/r06_ensi_soak.txt
  python tools/neighbourhood_soak.py 90 2>&1 | noids > $O/r06_neighbourhood_soak.txt
  python tools/qf_soak.py 120 2>&1 | noids > $O/r06_qf_soak.txt
  python tools/march_soak.py 90 2>&1 | noids > $O/r06_march_soak.txt
  python tools/fuzz_next_rows.py 120 2>&1 | noids > $O/r06_fuzz_next_rows.txt
  python tools/ensi_big_n.py 2>&1 | noids > $O/r06_ensi_big_n.txt
  python tools/ensi_illcond.py 2>&1 | noids > $O/r06_ensi_illcond.txt
  for f in $O/r06_*soak*.txt $O/r06_fuzz_next_rows.txt $O/r06_ensi_big_n.txt; do echo "== $f"; tail -n 3 $f | cut -c1-200; done ;;
bench)
  python tools/slice_overhead.py 2>&1 | noids > $O/r06_slice_overhead.txt
  python tools/slice_overhead_other.py 2>&1 | noids > $O/r06_slice_overhead_other.txt
  python tools/reference_benchmark_rows.py 2>/dev/null > $O/r06_reference_benchmark_rows.jsonl
  python tools/oi_variants.py 2>/dev/null > $O/r06_oi_variants.jsonl
  python bench.py > $O/r06_bench_n1_final.json 2> $O/bench.err
  cat $O/r06_slice_overhead.txt; cut -c1-600 $O/r06_bench_n1_final.json ;;
esac
