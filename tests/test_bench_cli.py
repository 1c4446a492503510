"""bench.py's launcher contract: --gpus N must agree with the launcher's WORLD_SIZE; without a launcher it starts N
ranks itself (rehearsed on the one-GPU box with every rank on cuda:0 and gloo in place of RCCL)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def test_gpus_flag_disagreeing_with_world_size_is_an_error():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "--gpus 2 disagrees with WORLD_SIZE=4" in r.stderr
    assert r.stdout.strip() == ""
    env.pop("WORLD_SIZE")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "0"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus must be >= 1" in r.stderr


@pytest.mark.gpu
def test_gpus_2_starts_two_ranks(dev):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(IPX_BENCH_DEVICE="0", IPX_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4", "--cpu-sample", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["batch_per_gpu"] == 4 and j["value"] > 0
    assert j["scaling"] == "weak" and j["cpu_baseline"] is None
    assert set(j["latency_b1_us"]) >= {"gaussian5_bgr", "bgr2gray_equalize", "sobel3_gray", "step"}
    assert set(j["kernels"]) >= {"gaussian5_bgr", "bgr2gray_hist", "equalize_table_pass", "bgr2gray_equalize", "sobel3_gray"}
