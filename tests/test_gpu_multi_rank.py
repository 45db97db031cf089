"""GPU: bench.py's N-rank code path, as the round driver launches it.

One process per rank -- through `python -m torch.distributed.run`, and through
bench.py's own spawn (`python3 bench.py --gpus 2`, no launcher) -- every rank a
shard of the alignments made by gfal_scorer_create_sharded, the per-path
counters all-reduced inside the timed step.  A one-GPU box has no second device,
so GFALIGN_BENCH_REHEARSAL=1 puts both ranks on cuda:0 and sums through gloo;
everything else (process group, sharded scorers, barrier / max-over-ranks
timing, the JSON line) is the code the 2/4/8-GPU runs execute.  The summed
counters must equal the one-rank run bit for bit (`counter_checksum`).
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(extra_env, launcher, gpus=1):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):       # (a plain `python bench.py` starts without them)
        env.pop(k, None)
    cmd = launcher + [os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1",
                      "--workload", "config2", "--no-cpu-baseline", "--no-search-mode"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_two_ranks_sum_to_the_one_rank_counters():
    from gfalign_amd import scorer
    if scorer.device_count() < 1:
        pytest.fail("needs an MI355X")
    one = _bench({}, [sys.executable])
    two = _bench({"GFALIGN_BENCH_REHEARSAL": "1"},
                 [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                  "--master-addr", "127.0.0.1", "--master-port", str(_free_port())], gpus=2)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["counter_checksum"] == one["config"]["counter_checksum"]
    assert two["value"] > 0 and two["scaling"] == "strong"
    # exactly as a driver without a launcher would start it: `python3 bench.py --gpus 2`
    # spawns its own rank processes (before anything touches the GPU in the parent)
    own = _bench({"GFALIGN_BENCH_REHEARSAL": "1"}, [sys.executable], gpus=2)
    assert own["n_gpus"] == 2
    assert own["config"]["counter_checksum"] == one["config"]["counter_checksum"]


def test_a_failing_rank_fails_the_run():
    """--gpus 2 without the rehearsal switch on a one-GPU box: rank 1 has no device; the
    parent must end rank 0 (it would wait at the rendezvous) and exit non-zero."""
    from gfalign_amd import scorer
    if scorer.device_count() != 1:
        pytest.skip("needs a box with exactly one GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "GFALIGN_BENCH_REHEARSAL"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--workload", "config2", "--no-cpu-baseline", "--no-search-mode"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
