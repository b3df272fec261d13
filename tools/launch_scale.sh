#!/bin/bash
# The multi-GPU runs of one node, exactly as the driver launches them (one process per GPU, RCCL over xGMI).
#   tools/launch_scale.sh [oi|ensi|nbh] [steps] [warmup]      -> one JSON line per N in gpurun_out/scale_<case>.jsonl
# Every line carries n_gpus, n_ranks_seen (WORLD_SIZE and torch.distributed's world size) and env_overrides; the library
# refuses to run under GPP_* overrides.  The C++ equivalent (no torch): gridpp::multi in gridpp_amd/host/gridpp.hpp --
#   rank r: gpp_set_device(r); multi::init(rank, world, id128 from rank 0's gpp_comm_unique_id);
#           multi::optimal_interpolation(tile_grid, tile_background, points, obs, ...)   (tests/cpp/test_host_api.cpp)
set -u
cd "$(dirname "$0")/.."
CASE=${1:-oi}; STEPS=${2:-20}; WARMUP=${3:-3}
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
OUT=gpurun_out/scale_$CASE.jsonl
: > $OUT
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
# GPP_BENCH_SHARE_GPU=1 (with GPP_BENCH_BACKEND=gloo): every rank on GPU 0 -- the logic test of a one-GPU box (tests/test_gpu_bench_contract.py),
# never a measurement; GPP_BENCH_EXTRA: extra bench.py arguments for it (a smaller grid)
[ "${GPP_BENCH_SHARE_GPU:-0}" = 1 ] && NGPU=8
EXTRA=${GPP_BENCH_EXTRA:-}
for N in 1 2 4 8; do
  [ "$N" -le "$NGPU" ] || { echo "skipping N=$N: $NGPU GPUs visible" >&2; continue; }
  if [ "$N" -eq 1 ]; then
    python bench.py --gpus 1 --steps $STEPS --warmup $WARMUP --case $CASE --no-cpu-baseline --no-other-configs $EXTRA | grep '^{' | tail -1 >> $OUT
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) \
        bench.py --gpus $N --steps $STEPS --warmup $WARMUP --case $CASE --no-cpu-baseline --no-other-configs $EXTRA | grep '^{' | tail -1 >> $OUT
  fi
done
python - "$OUT" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
base = rows[0]["value"] if rows else 0
for r in rows:
    print("N=%d ranks_seen=%s ms/step=%.3f value=%.4g speedup=%.2f" % (r["n_gpus"], r.get("n_ranks_seen"), r["ms_per_step"], r["value"], r["value"] / base))
PY
