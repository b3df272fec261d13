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
