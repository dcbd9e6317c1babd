"""GPU parity fuzz: a compact run of tests/fuzz_parity.py (random shapes incl. > 1024 rows, integer /
degenerate / near-tie data, primal and dual loops, eager / graph / batched group execution), bitwise."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [1, 2])
def test_randomized_parity_sweep(gpu, oracle, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), str(seed), "150"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout


def test_randomized_group_sweep(gpu, oracle):
    """tests/fuzz_groups.py: lpx_multi_run on groups of up to 400 node LPs of mixed shapes and loops."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_groups.py"), "11", "10"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout


def test_randomized_revised_sweep(gpu, oracle):
    """tests/fuzz_revised.py: the revised path on ragged shapes (one of them mid-size), pivot sequences equal to the oracle's."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_revised.py"), "5", "10"],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout
