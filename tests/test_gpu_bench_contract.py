"""bench.py keeps its contract: one JSON line with the required keys at N = 1, and the N > 1 rank logic (row tiles, double-
buffered observation broadcast, max-over-ranks timing) runs end to end.  A one-GPU box cannot host two RCCL ranks, so the
N = 2 run puts both ranks on GPU 0 and exchanges over gloo (GPP_BENCH_SHARE_GPU / GPP_BENCH_BACKEND): it checks the logic,
not the speed."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline"}


def _json_line(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert len(lines) == 1, text[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-seconds", "1"],
                         capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = _json_line(out.stdout)
    assert KEYS <= set(r) and "cpu_baseline" in r
    assert r["n_gpus"] == 1 and r["steps"] == 3 and r["warmup"] == 1 and r["value"] > 1e8
    assert set(r["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(r["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample", "one_thread_value"}
    assert r["roofline_compute"]["bound"] == "valu_issue" and 0 < r["roofline_compute"]["frac"] < 1
    assert r["host_inclusive"]["ms_per_step"] > r["ms_per_step"]
    cases = " | ".join(c["case"] for c in r["other_configs"])
    for key in ("C1 OI", "C2 OI", "C4 neighbourhood Mean", "C4 quantile_fast", "C5 EnSI"):
        assert key in cases, cases
    for c in r["other_configs"]:
        assert c["ms"] > 0 and c["Mcells/s"] > 0 and 0 < c["frac_hbm"] < 1, c
    assert abs(r["value"] - 16e6 / (r["ms_per_step"] * 1e-3)) / r["value"] < 1e-6


def test_two_rank_logic_on_one_gpu():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, GPP_BENCH_SHARE_GPU="1", GPP_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    out = subprocess.run(cmd + ["--equal-tiles"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    r = _json_line(out.stdout)
    assert KEYS <= set(r) and r["n_gpus"] == 2 and r["scaling"] == "strong" and r["value"] > 0
    assert r["kernel"]["cells_per_launch"] == 8_000_000          # rank 0's row tile: half of the 4000 x 4000 grid
    assert "one analysis ahead" in r["config"]["calls"]
    # row tiles by measured kernel time (two ranks sharing one GPU differ by far more than the 3 % that trigger it) and blocking calls
    out = subprocess.run(cmd + ["--sync-calls"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    r = _json_line(out.stdout)
    note = r["config"]["row_tiles"]
    assert len(note["equal_tiles_kernel_ms"]) == 2 and note["spread"] >= 0
    if note["rebalanced"]:
        (a0, a1), (b0, b1) = note["row_tiles"]
        assert a0 == 0 and a1 == b0 and b1 == 4000 and r["kernel"]["cells_per_launch"] == (a1 - a0) * 4000
    assert "blocking" in r["config"]["calls"]


@pytest.mark.parametrize("case,extra", [("ensi", ["--ny", "256", "--nx", "256", "--obs", "500"]), ("nbh", ["--ny", "512", "--nx", "256"])])
def test_other_cases_two_rank_logic(case, extra):
    """--case ensi / nbh: row tiles, block broadcast resp. halo exchange, on two ranks sharing GPU 0 over gloo (logic only)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, GPP_BENCH_SHARE_GPU="1", GPP_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--case", case] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    r = _json_line(out.stdout)
    assert KEYS <= set(r) and r["n_gpus"] == 2 and r["value"] > 0 and r["roofline"]["achieved"] > 0


def _torchrun(n, args, env_extra=None, timeout=1500):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, GPP_BENCH_SHARE_GPU="1", GPP_BENCH_BACKEND="gloo", **(env_extra or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--no-other-configs"] + args
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    return _json_line(out.stdout)


@pytest.mark.parametrize("mode", ["equal-tiles", "rebalanced"])
def test_eight_rank_logic_on_one_gpu_oi(mode):
    """The rank logic of the 8-GPU run as it will run on the node -- eight processes, eight row tiles of 500 rows, the three-slot observation
    stream with one analysis ahead, max-over-ranks timing -- on ONE GPU over gloo (no RCCL: logic, not speed).  `rebalanced`: the row tiles
    cut again by measured kernel time (forced: ranks sharing a device are further apart than the 50 % that normally rules a workload out)."""
    args = ["--ny", "4000", "--nx", "512", "--obs", "2000"]
    if mode == "equal-tiles":
        r = _torchrun(8, args + ["--equal-tiles"])
        assert r["kernel"]["cells_per_launch"] == 500 * 512
    else:
        r = _torchrun(8, args, {"GPP_BENCH_FORCE_REBALANCE": "1"})
        note = r["config"]["row_tiles"]
        assert len(note["equal_tiles_kernel_ms"]) == 8 and note["rebalanced"]
        tiles = note["row_tiles"]
        assert tiles[0][0] == 0 and tiles[-1][1] == 4000 and all(tiles[k][1] == tiles[k + 1][0] for k in range(7)) and all(t1 - t0 >= 8 for t0, t1 in tiles)
        assert r["kernel"]["cells_per_launch"] == (tiles[0][1] - tiles[0][0]) * 512
    assert KEYS <= set(r) and r["n_gpus"] == 8 and r["scaling"] == "strong" and r["value"] > 0
    assert r["n_ranks_seen"] == {"env_WORLD_SIZE": 8, "torch_distributed": 8} or r["n_ranks_seen"]["torch_distributed"] == 8
    assert "one analysis ahead" in r["config"]["calls"]


@pytest.mark.parametrize("case,extra", [("ensi", ["--ny", "512", "--nx", "128", "--obs", "400"]), ("nbh", ["--ny", "4000", "--nx", "128"])])
def test_eight_rank_logic_on_one_gpu_other_cases(case, extra):
    """--case ensi (block broadcast) and nbh (4000 rows in eight tiles of 500 + 15 halo rows on either side: the 530-row tiles of the node)."""
    r = _torchrun(8, ["--case", case] + extra)
    assert KEYS <= set(r) and r["n_gpus"] == 8 and r["value"] > 0 and r["roofline"]["achieved"] > 0
    assert r["n_ranks_seen"]["torch_distributed"] == 8


def test_launch_scale_emits_one_line_per_rank_count():
    """tools/launch_scale.sh -- the command sequence of the driver's scaling run -- leaves exactly one JSON line per N = 1, 2, 4, 8, each from a
    run that saw N ranks (here all on GPU 0 over gloo, on a small grid)."""
    env = dict(os.environ, GPP_BENCH_SHARE_GPU="1", GPP_BENCH_BACKEND="gloo", GPP_BENCH_EXTRA="--ny 1600 --nx 512 --obs 1500 --equal-tiles")
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "launch_scale.sh"), "oi", "2", "1"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=2400)
    assert out.returncode == 0, out.stderr[-3000:]
    rows = [json.loads(l) for l in open(os.path.join(ROOT, "gpurun_out", "scale_oi.jsonl")) if l.strip()]
    assert [r["n_gpus"] for r in rows] == [1, 2, 4, 8], [r["n_gpus"] for r in rows]
    for r in rows:
        assert r["n_ranks_seen"]["torch_distributed"] == r["n_gpus"] and r["value"] > 0
        assert "GPP_BENCH_SHARE_GPU" in r["env_overrides"]       # (recorded: such a line can never pass for a measurement)
