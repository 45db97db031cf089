"""GPU: a fresh slice of the randomised parity runs on every suite run.

scripts/fuzz_gpu.py (full evaluation: shapes, shards, weighted lanes, universes,
either DP kernel family) and scripts/fuzz_children.py (search mode: trees of
candidates scored from their parents) draw their cases from a seed; here the seed
is the clock's, so successive suite runs cover different cases.  A failure prints
the seed of the failing case; GFALIGN_FUZZ_SEED=<n> repeats a run.
"""
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gpu():
    from gfalign_amd import scorer
    if scorer.device_count() < 1:
        pytest.fail("these tests need an MI355X: the product path has no CPU fallback")


def _seed():
    return int(os.environ.get("GFALIGN_FUZZ_SEED", int(time.time()) % 1000000))


@pytest.mark.parametrize("script,cases,env", [("fuzz_gpu.py", 50, {}), ("fuzz_gpu.py", 40, {"GFAL_SCAN": "3"}),
                                              ("fuzz_children.py", 30, {})])
def test_fresh_random_cases(gpu, script, cases, env):
    """(GFAL_SCAN=3: k_tile / k_scan3 also for the small batches that k_scan would take)"""
    seed = _seed()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script), str(cases), str(seed)],
                       cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, "seed0 %d\n%s\n%s" % (seed, p.stdout[-3000:], p.stderr[-2000:])
    assert "OK:" in p.stdout
