"""gfalign_amd -- MI355X-native scorer for the `gfalign search` hot path.

The product is the C ABI in include/gfalign_scorer.h, implemented in
csrc/scorer.hip (hand-written HIP for gfx950).  This package holds the host
side around it: the ctypes binding (scorer.py), the synthetic-tangle generator
(synth.py), the build helper (build.py) and the multi-GPU sharding glue
(shard.py).
"""
from .scorer import Scorer, ScorerError, pack_step, device_count  # noqa: F401

__all__ = ["Scorer", "ScorerError", "pack_step", "device_count"]
