mkdir -p gpurun_out/r05
python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|error" > gpurun_out/r05/r05_pytest_gpu.txt
GPP_LIB=$PWD/gridpp_amd/lib/var_poison.so GRIDPP_TEST_POISON=1 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|error" > gpurun_out/r05/r05_pytest_gpu_poisoned.txt
GPP_LIB=$PWD/gridpp_amd/lib/var_poison.so python tools/nbh_hostile_soak.py 0 160 20 poison dump=/tmp/d > gpurun_out/r05/r05_nbh_hostile_soak.txt 2>&1
python tools/qf_soak.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/r05_qf_soak.txt
python tools/march_soak.py 150 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/r05_march_soak.txt
python tools/neighbourhood_soak.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/r05_neighbourhood_soak.txt
PROFILE_ONLY=nbh bash tools/profile_r05.sh > /dev/null 2>&1
for f in gpurun_out/r05/*.txt; do echo "== $f"; tail -n 2 $f; done
python tools/slice_overhead_other.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/r05_slice_overhead_other.txt
python bench.py > gpurun_out/r05/r05_bench_n1_final.json 2> gpurun_out/r05/bench.err
cut -c1-400 gpurun_out/r05/r05_bench_n1_final.json
