"""GPU: bench.py's N-rank code path, as the round driver launches it.

One process per rank through `python -m torch.distributed.run`, every rank a
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


def _bench(extra_env, launcher):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = launcher + [os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--workload", "config2",
                      "--no-cpu-baseline", "--no-search-mode"]
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
                  "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] )
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["counter_checksum"] == one["config"]["counter_checksum"]
    assert two["value"] > 0 and two["scaling"] == "strong"
