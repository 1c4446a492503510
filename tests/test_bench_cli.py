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
    assert set(j["latency_b1_us"]) >= {"gaussian5_bgr+bgr2gray", "hist_tables_gray", "sobel3_gray+equalize_table_pass", "step"}
    assert set(j["kernels"]) >= {"gaussian5_bgr+bgr2gray", "hist_tables_gray", "sobel3_gray+equalize_table_pass"}
    assert j["config"]["launch_order"] == ["gaussian5_bgr+bgr2gray", "hist_tables_gray", "sobel3_gray+equalize_table_pass"]
    assert j["roofline"]["op"] == "gaussian5_bgr+bgr2gray"
    # the four-launch form of round 2 stays available
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2", "--cpu-sample", "0", "--order", "sobel_mid"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    j4 = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert set(j4["kernels"]) >= {"gaussian5_bgr", "bgr2gray_hist", "equalize_table_pass", "bgr2gray_equalize", "sobel3_gray"}


def test_a_rank_that_dies_before_the_rendezvous_ends_the_run_quickly():
    """the parent watches every child: rank 1 exits with code 3 before init_process_group; rank 0 (which would sit in the
    rendezvous) is stopped and bench.py returns non-zero within seconds, naming the rank."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(IPX_BENCH_FAIL_RANK="1", IPX_BENCH_BACKEND="gloo")
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--cpu-sample", "0"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert time.time() - t0 < 20.0
    assert "rank 1 exited with code 3" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_silent_ranks_are_stopped_after_the_rank_timeout():
    """--rank-timeout: with no GPU here every rank fails on its own, so make them hang instead (a test knob that sleeps)."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(IPX_BENCH_HANG="1")
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rank-timeout", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and time.time() - t0 < 20.0
    assert "--rank-timeout 2 s" in r.stderr and "ranks still running: [0, 1]" in r.stderr


@pytest.mark.gpu
def test_rowpart_workload_two_ranks_on_one_gpu(dev):
    """BASELINE configs[4] as a bench mode: 2 ranks (gloo, both on cuda:0) share one 2048 x 2048 gray image by rows;
    N = 1 runs the whole image and reports the CPU baseline."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(IPX_BENCH_DEVICE="0", IPX_BENCH_BACKEND="gloo")
    base = [sys.executable, BENCH, "--workload", "rowpart", "--rowpart-size", "2048", "--steps", "3", "--warmup", "1"]
    r = subprocess.run(base + ["--gpus", "2", "--cpu-sample", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and len(j["ms_per_step_per_rank"]) == 2
    assert j["scaling"] == "strong" and j["config"]["rows_per_rank"] == 1024 and j["value"] > 0
    assert set(j["kernels"]) == {"hist_tables_gray", "box3_gray+equalize_table_pass"} and j["roofline"]["frac"] > 0
    assert j["config"]["form"] == "fused"
    r = subprocess.run(base + ["--gpus", "2", "--cpu-sample", "0", "--rowpart-form", "separate"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert set(j["kernels"]) == {"equalize_hist_gray", "box3_gray"} and j["ranks_seen"] == 2
    r = subprocess.run(base + ["--gpus", "1", "--cpu-sample", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["config"]["rows_per_rank"] == 2048 and j["cpu_baseline"]["value"] > 0


@pytest.mark.gpu
def test_the_drivers_launch_form_torch_distributed_run(dev):
    """N > 1 as the driver starts it: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W` (one rank per GPU; here both ranks on cuda:0 with gloo in place of
    RCCL): rank 0 prints the one JSON line, every rank's step time is in it."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(IPX_BENCH_DEVICE="0", IPX_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4", "--cpu-sample", "0"],
                       env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and len(j["ms_per_step_per_rank"]) == 2 and j["steps"] == 3 and j["warmup"] == 1
    assert j["value"] > 0 and j["scaling"] == "weak" and j["roofline"]["frac"] > 0
