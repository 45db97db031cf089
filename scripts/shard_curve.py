"""What one rank of an N-GPU run does, emulated on one GPU: the config-3 batch
(10 000 paths, device-resident) and a search-sized batch (128 paths, blocking
call from host buffers) against 1/N of the alignments, N = 1, 2, 4, 8.  The
N-GPU step is this plus one all-reduce of uint32[3P] (120 KB / 1.5 KB)."""
import gc, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gfalign_amd import synth
from gfalign_amd.scorer import Scorer

# (the interpreter's cyclic collector off: one ~40 ms generation-2 collection used to land, at a
# deterministic allocation count, inside the 50-call loop of the last shard of the 4-way split --
# the "0.97 / 1.03 / 0.93 ms" outlier of the round-2 and mid-round-3 records; the device time of
# those calls was 0.21 ms like its neighbours', scripts/small_shard_probe.py)
gc.disable()
t = synth.make(sys.argv[1] if len(sys.argv) > 1 else "config3")
dev = torch.device("cuda", 0)
P = t.P
d_off = torch.from_numpy(t.path_off).to(dev); d_st = torch.from_numpy(t.path_steps).to(dev)
d_cnt = torch.zeros(3 * P, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream(dev)
mx = int(np.diff(t.path_off).max())
rng = np.random.default_rng(5)
pick = np.sort(rng.choice(P, 128, replace=False))
soff = [0]; sst = []
for p in pick:
    sst.append(t.path_steps[t.path_off[p]:t.path_off[p + 1]]); soff.append(soff[-1] + len(sst[-1]))
soff = np.asarray(soff, np.int32); sst = np.concatenate(sst).astype(np.int32)
base = {}
for world in (1, 2, 4, 8):
  worst = {"big": 0.0, "small": 0.0}
  for k in range(world):          # every shard in turn: the slowest one is the N-GPU step
    with Scorer(t.aln_off, t.aln_steps, t.V, shard=(k, world)) as sc:
        for _ in range(3):
            sc.score_device(d_off.data_ptr(), d_st.data_ptr(), P, int(t.path_off[-1]), mx, True, d_cnt.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            sc.score_device(d_off.data_ptr(), d_st.data_ptr(), P, int(t.path_off[-1]), mx, True, d_cnt.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        big = (time.perf_counter() - t0) / 10 * 1e3
        for _ in range(5):
            sc.evaluate_paths(soff, sst, True)
        t0 = time.perf_counter()
        for _ in range(50):
            sc.evaluate_paths(soff, sst, True)
        small = (time.perf_counter() - t0) / 50 * 1e3
    worst["big"] = max(worst["big"], big)
    worst["small"] = max(worst["small"], small)
  big, small = worst["big"], worst["small"]
  if world == 1:
      base = {"big": big, "small": small}
  print("1/%d of the alignments (slowest shard): 10 000-path step %.3f ms (%.0f %% of ideal 1/%d), 128-path blocking call %.3f ms (%.0f %%)" % (
      world, big, 100 * base["big"] / world / big, world, small, 100 * base["small"] / world / small), flush=True)
