"""Probe: scan time when alignments are long (ONT-like: m = 2 + Poisson(lam))."""
import os, sys, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gfalign_amd import synth
from gfalign_amd.scorer import Scorer
import oracle

lam = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
N, P = 200_000, 2_000
rng = synth.Rng(5)
T, nodes, mult = synth.make_truth_walk(2000, 900, rng)
# alignments with a heavier length distribution
cdf = synth._poisson_cdf(lam, 60)
m = 2 + np.minimum(np.searchsorted(cdf, rng.uniform(N)), 60).astype(np.int64)
starts = rng.below(N, len(T) - m + 1)
off, steps = synth._subwalks(T, starts, m)
flip = (rng.bits(N) & np.uint64(1)).astype(bool)
pos_in = np.arange(off[-1]) - np.repeat(off[:-1], m)
mirror = np.repeat(off[:-1] + m - 1, m) - pos_in
steps = np.where(np.repeat(flip, m), steps[mirror] ^ 1, steps).astype(np.int32)
sub = rng.uniform(N) < 0.1
at = off[:-1] + rng.below(N, m)
steps[at[sub]] = T[rng.below(N, len(T))][sub]
poff, pst = synth.make_candidates(T, P, rng)
dev = torch.device("cuda", 0)
sc = Scorer(off.astype(np.int32), steps, 2000)
d_off = torch.from_numpy(poff).to(dev); d_st = torch.from_numpy(pst).to(dev)
d_cnt = torch.zeros(3 * P, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream(dev)
mx = int(np.diff(poff).max())
for i in range(2):
    sc.score_device(d_off.data_ptr(), d_st.data_ptr(), P, int(poff[-1]), mx, True, d_cnt.data_ptr(), stream.cuda_stream)
torch.cuda.synchronize(); sc.set_profiling(True)
for i in range(5):
    sc.score_device(d_off.data_ptr(), d_st.data_ptr(), P, int(poff[-1]), mx, True, d_cnt.data_ptr(), stream.cuda_stream)
torch.cuda.synchronize(); info = sc.info()
pairs = N * P
print("lam %.0f: mean m %.1f  max m %d  scan %.2f ms dp %.2f ms -> %.1f Gpairs/s (config3 runs at %.0f)" % (
    lam, m.mean(), m.max(), info["scan_ms"], info["dp_ms"], pairs / info["scan_ms"] / 1e6, 1e10 / 12.7 / 1e6))
# parity on a sample
pick = np.linspace(0, P - 1, 6).astype(int)
paths = [pst[poff[k]:poff[k + 1]] for k in pick]
po = np.zeros(len(paths) + 1, np.int32); po[1:] = np.cumsum([len(p) for p in paths])
c = d_cnt.cpu().numpy().view(np.uint32)
eb, eg, eu = oracle.evaluate_paths(off[:20001].astype(np.int32), steps[:off[20000]], po, np.concatenate(paths), True)
with Scorer(off[:20001].astype(np.int32), steps[:off[20000]], 2000) as s2:
    gb, gg, gu = s2.evaluate_paths(po, np.concatenate(paths), True)
print("parity on sample:", np.array_equal(gb, eb) and np.array_equal(gg, eg) and np.array_equal(gu, eu))
