"""CPU: the synthetic-tangle generator is deterministic and well-formed."""
import numpy as np

from gfalign_amd import synth


def test_deterministic():
    a, b = synth.make("smoke"), synth.make("smoke")
    for name in ("T", "aln_off", "aln_steps", "path_off", "path_steps"):
        assert np.array_equal(getattr(a, name), getattr(b, name))
    c = synth.Tangle(V=60, n_T=50, N=2000, P=48, seed=8)
    assert not np.array_equal(a.aln_steps, c.aln_steps)


def test_shape_of_config2():
    t = synth.make("config2")
    assert t.N == 100_000 and t.P == 1_000
    assert t.T[0] == 0 and t.T[-1] == (t.V - 1) << 1
    assert 300 <= t.n_T <= 1000
    m = np.diff(t.aln_off)
    assert m.min() >= 2 and m.max() <= 32 and 4.0 < m.mean() < 5.0
    n = np.diff(t.path_off)
    assert n.min() >= 2 and n.max() <= t.n_T
    assert t.aln_steps.min() >= 0 and (t.aln_steps >> 1).max() < t.V
    # ~5 % of alignments touch a node outside the walk
    on = np.zeros(t.V, bool)
    on[t.walk_nodes] = True
    outside = ~on[t.aln_steps >> 1]
    frac = np.add.reduceat(outside, t.aln_off[:-1]).astype(bool).mean()
    assert 0.03 < frac < 0.07
    assert t.algorithmic_bytes() == t.P * (4 * t.S + 4 * (t.N + 1) + 12) + 4 * int(t.path_off[-1])
