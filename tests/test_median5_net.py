"""The min / max networks of the 5 x 5 ... 11 x 11 median kernels: the committed ipx_median5_net.inc is what the generator emits, and
the emitted text selects the median (all 2^25 0-1 inputs and random byte windows, run on the CPU by tests/tools/median5_net_check.cpp)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_network_is_the_generated_one():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen_median5_net.py"), "--check"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "correct" in r.stdout and "up to date" in r.stdout


@pytest.mark.parametrize("k", [7, 9, 11])
def test_committed_larger_networks_are_the_generated_ones(k):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", f"gen_median{k}_net.py"), "--check"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "correct" in r.stdout and "up to date" in r.stdout


def test_emitted_network_selects_the_median(tmp_path):
    exe = tmp_path / "median5_net_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "image-processing_amd", "csrc"),
                    os.path.join(ROOT, "tests", "tools", "median5_net_check.cpp"), "-o", str(exe)], check=True, timeout=300)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr
