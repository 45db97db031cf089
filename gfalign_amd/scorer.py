"""Host-side binding of the C ABI in include/gfalign_scorer.h (ctypes).

Mirrors the one seam the reference has on this path -- ``evaluatePath``
(reference src/eval.cpp:67-108) -- as :meth:`Scorer.evaluate_paths`.  There is
no CPU fallback here: if ``libgfalign_scorer.so`` is missing or no HIP device
is usable, construction raises.
"""
import ctypes
import os

import numpy as np

from . import build as _build

GFAL_MAX_STEPS = 1000
GFAL_STEP_OTHER = 0x40000000

_i32p = ctypes.POINTER(ctypes.c_int32)
_u32p = ctypes.POINTER(ctypes.c_uint32)


class GfalInfo(ctypes.Structure):
    _fields_ = [("n_aln", ctypes.c_int64), ("n_steps", ctypes.c_int64),
                ("n_nodes", ctypes.c_int32), ("n_local_nodes", ctypes.c_int32),
                ("max_aln_len", ctypes.c_int32), ("tile_paths", ctypes.c_int32),
                ("n_workgroups", ctypes.c_int32), ("lds_bytes", ctypes.c_int32),
                ("dp_pairs", ctypes.c_int64), ("scan_ms", ctypes.c_float),
                ("dp_ms", ctypes.c_float), ("total_ms", ctypes.c_float),
                ("profiled_calls", ctypes.c_int32), ("n_lanes", ctypes.c_int64),
                ("n_score_calls", ctypes.c_int64), ("n_device_passes", ctypes.c_int64),
                ("n_overflow_reruns", ctypes.c_int64), ("wl_capacity", ctypes.c_int64),
                ("scan_kernel_ms", ctypes.c_float), ("reserved_", ctypes.c_int32)]


# every symbol include/gfalign_scorer.h declares
EXPORTS = {
    "gfal_abi_version": (ctypes.c_int, []),
    "gfal_build_id": (ctypes.c_char_p, []),
    "gfal_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "gfal_last_error": (ctypes.c_char_p, []),
    "gfal_device_count": (ctypes.c_int, []),
    "gfal_scorer_create": (ctypes.c_int, [_i32p, _i32p, ctypes.c_int64, ctypes.c_int32,
                                          ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "gfal_scorer_create_sharded": (ctypes.c_int, [_i32p, _i32p, ctypes.c_int64, ctypes.c_int32,
                                                  ctypes.c_int, _i32p, ctypes.c_int32, ctypes.c_int32,
                                                  ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]),
    "gfal_scorer_create_dedup": (ctypes.c_int, [_i32p, _i32p, ctypes.c_int64, ctypes.c_int32,
                                                ctypes.c_int, _i32p, ctypes.c_int32, ctypes.c_int32,
                                                ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]),
    "gfal_scorer_create_ex": (ctypes.c_int, [_i32p, _i32p, ctypes.c_int64, ctypes.c_int32,
                                             ctypes.c_int, _i32p, ctypes.c_int32,
                                             ctypes.POINTER(ctypes.c_void_p)]),
    "gfal_shard_owner": (ctypes.c_int, [_i32p, _i32p, ctypes.c_int64, ctypes.c_int32, _i32p,
                                        ctypes.c_int32, ctypes.c_int32, _i32p]),
    "gfal_scorer_destroy": (None, [ctypes.c_void_p]),
    "gfal_scorer_score": (ctypes.c_int, [ctypes.c_void_p, _i32p, _i32p, ctypes.c_int32,
                                         ctypes.c_int, _u32p, _u32p, _u32p]),
    "gfal_scorer_score_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_void_p, ctypes.c_int32,
                                                ctypes.c_int64, ctypes.c_int32,
                                                ctypes.c_int, ctypes.c_void_p,
                                                ctypes.c_void_p]),
    "gfal_group_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_void_p)]),
    "gfal_group_destroy": (None, [ctypes.c_void_p]),
    "gfal_group_uses_rccl": (ctypes.c_int, [ctypes.c_void_p]),
    "gfal_group_score": (ctypes.c_int, [ctypes.c_void_p, _i32p, _i32p, ctypes.c_int32, ctypes.c_int,
                                        _u32p, _u32p, _u32p]),
    "gfal_group_score_begin": (ctypes.c_int, [ctypes.c_void_p, _i32p, _i32p, ctypes.c_int32, ctypes.c_int]),
    "gfal_group_score_end": (ctypes.c_int, [ctypes.c_void_p, _u32p, _u32p, _u32p]),
    "gfal_group_score_poll": (ctypes.c_int, [ctypes.c_void_p]),
    "gfal_group_store_reserve": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64]),
    "gfal_group_score_store_begin": (ctypes.c_int, [ctypes.c_void_p, _i32p, _i32p, ctypes.c_int32, _i32p]),
    "gfal_group_score_children_begin": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, _i32p, _i32p, _i32p,
                                                       ctypes.c_int32]),
    "gfal_scorer_sync_status": (ctypes.c_int, [ctypes.c_void_p]),
    "gfal_scorer_pair_scores": (ctypes.c_int, [ctypes.c_void_p, _i32p, ctypes.c_int32,
                                               _i32p, _i32p]),
    "gfal_scorer_set_profiling": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "gfal_scorer_get_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(GfalInfo)]),
}

_lib = None


class ScorerError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        super().__init__(what)


def library_path():
    # GFALIGN_SCORER_SO: an experimental build of the same ABI (scripts/ only)
    return os.environ.get("GFALIGN_SCORER_SO") or _build.SCORER_SO


def load_library():
    """dlopen the in-tree shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise ScorerError(-3, "%s is missing: run `python -m gfalign_amd.build` "
                                  "(there is no CPU fallback)" % path)
        lib = ctypes.CDLL(path)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _check(code):
    if code != 0:
        lib = load_library()
        msg = lib.gfal_strerror(code).decode()
        detail = lib.gfal_last_error().decode()
        raise ScorerError(code, "%s%s" % (msg, (": " + detail) if detail else ""))


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _ptr(a, ty):
    return a.ctypes.data_as(ctypes.POINTER(ty))


def shard_owner(aln_off, aln_steps, n_nodes, n_shards, universe=None):
    """owner[k] = the shard of gfal_scorer_create_sharded that takes alignment k
    (host code of the library: works without a GPU)."""
    lib = load_library()
    aln_off, aln_steps = _i32(aln_off), _i32(aln_steps)
    n = len(aln_off) - 1
    owner = np.zeros(max(n, 1), np.int32)
    uni = None if universe is None else _i32(universe)
    _check(lib.gfal_shard_owner(_ptr(aln_off, ctypes.c_int32), _ptr(aln_steps, ctypes.c_int32), n,
                                int(n_nodes), None if uni is None else _ptr(uni, ctypes.c_int32),
                                0 if uni is None else len(uni), int(n_shards),
                                _ptr(owner, ctypes.c_int32)))
    return owner[:n]


def pack_step(node_id, orientation):
    """Reference Step{id, orientation} -> packed int32 (see the header)."""
    if orientation == "+":
        return node_id << 1
    if orientation == "-":
        return (node_id << 1) | 1
    return GFAL_STEP_OTHER | (node_id << 1)


class Scorer:
    """One shard of alignments resident on one MI355X."""

    def __init__(self, aln_off, aln_steps, n_nodes, device=0, universe=None, shard=None,
                 dedup=False):
        """shard = (index, count): this scorer keeps its share of the alignment
        set given in full (gfal_scorer_create_sharded).  dedup: identical
        alignments collapsed into weighted lanes (gfal_scorer_create_dedup)."""
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        aln_off, aln_steps = _i32(aln_off), _i32(aln_steps)
        self.n_aln = len(aln_off) - 1
        if shard is not None or dedup:
            shard = shard or (0, 1)
            uni = None if universe is None else _i32(universe)
            create = self._lib.gfal_scorer_create_dedup if dedup else self._lib.gfal_scorer_create_sharded
            _check(create(
                _ptr(aln_off, ctypes.c_int32), _ptr(aln_steps, ctypes.c_int32),
                self.n_aln, int(n_nodes), int(device),
                None if uni is None else _ptr(uni, ctypes.c_int32), 0 if uni is None else len(uni),
                int(shard[0]), int(shard[1]), ctypes.byref(self._h)))
        elif universe is None:
            _check(self._lib.gfal_scorer_create(
                _ptr(aln_off, ctypes.c_int32), _ptr(aln_steps, ctypes.c_int32),
                self.n_aln, int(n_nodes), int(device), ctypes.byref(self._h)))
        else:
            universe = _i32(universe)
            _check(self._lib.gfal_scorer_create_ex(
                _ptr(aln_off, ctypes.c_int32), _ptr(aln_steps, ctypes.c_int32),
                self.n_aln, int(n_nodes), int(device),
                _ptr(universe, ctypes.c_int32), len(universe), ctypes.byref(self._h)))

    def close(self):
        if self._h:
            self._lib.gfal_scorer_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def evaluate_paths(self, path_off, path_steps, filter=True, want_unaligned=True):
        """evaluatePath (src/eval.cpp:67-108) per path -> (bad, good, unaligned)."""
        path_off, path_steps = _i32(path_off), _i32(path_steps)
        P = len(path_off) - 1
        bad = np.zeros(P, np.uint32)
        good = np.zeros(P, np.uint32)
        una = np.zeros(P, np.uint32) if want_unaligned else None
        _check(self._lib.gfal_scorer_score(
            self._h, _ptr(path_off, ctypes.c_int32), _ptr(path_steps, ctypes.c_int32),
            P, int(bool(filter)), _ptr(bad, ctypes.c_uint32), _ptr(good, ctypes.c_uint32),
            _ptr(una, ctypes.c_uint32) if want_unaligned else None))
        return bad, good, una

    def score_device(self, d_path_off, d_path_steps, n_paths, total_steps,
                     max_path_len, filter, d_counts, stream=0):
        """Async form on raw device pointers (ints); see the header."""
        _check(self._lib.gfal_scorer_score_device(
            self._h, ctypes.c_void_p(d_path_off), ctypes.c_void_p(d_path_steps),
            int(n_paths), int(total_steps), int(max_path_len), int(bool(filter)),
            ctypes.c_void_p(d_counts), ctypes.c_void_p(stream)))

    def sync_status(self):
        _check(self._lib.gfal_scorer_sync_status(self._h))

    def pair_scores(self, path_steps, out=None):
        """(fw, rc) traceback scores of every alignment vs one path.  A shard
        writes the entries of its own alignments only: pass the same `out`
        arrays to every shard of a set to fill them."""
        path_steps = _i32(path_steps)
        if out is None:
            fw = np.zeros(self.n_aln, np.int32)
            rc = np.zeros(self.n_aln, np.int32)
        else:
            fw, rc = out
            assert fw.dtype == np.int32 and rc.dtype == np.int32 and len(fw) == len(rc) == self.n_aln
        _check(self._lib.gfal_scorer_pair_scores(
            self._h, _ptr(path_steps, ctypes.c_int32), len(path_steps),
            _ptr(fw, ctypes.c_int32), _ptr(rc, ctypes.c_int32)))
        return fw, rc

    def set_profiling(self, on=True):
        """True / 1: events around every phase of a call; 2: around the dominant scan
        kernel only (what bench.py's timed region uses); False: none."""
        _check(self._lib.gfal_scorer_set_profiling(self._h, 2 if on == 2 else int(bool(on))))

    def info(self):
        out = GfalInfo()
        _check(self._lib.gfal_scorer_get_info(self._h, ctypes.byref(out)))
        return {name: getattr(out, name) for name, _ in GfalInfo._fields_}


def device_count():
    n = load_library().gfal_device_count()
    return max(n, 0)


class Group:
    """The shards of one alignment set scored together (gfal_group_*): counters
    summed by one RCCL all-reduce on the devices, or on the host when RCCL cannot
    serve the group (`uses_rccl`)."""

    def __init__(self, scorers):
        self._lib = load_library()
        self._scorers = list(scorers)                 # keep them alive
        arr = (ctypes.c_void_p * len(self._scorers))(*[s._h for s in self._scorers])
        self._h = ctypes.c_void_p()
        _check(self._lib.gfal_group_create(arr, len(self._scorers), ctypes.byref(self._h)))

    @property
    def uses_rccl(self):
        return bool(self._lib.gfal_group_uses_rccl(self._h))

    def evaluate_paths(self, path_off, path_steps, filter=True):
        path_off, path_steps = _i32(path_off), _i32(path_steps)
        P = len(path_off) - 1
        bad, good, una = (np.zeros(P, np.uint32) for _ in range(3))
        _check(self._lib.gfal_group_score(
            self._h, _ptr(path_off, ctypes.c_int32), _ptr(path_steps, ctypes.c_int32), P, int(bool(filter)),
            _ptr(bad, ctypes.c_uint32), _ptr(good, ctypes.c_uint32), _ptr(una, ctypes.c_uint32)))
        return bad, good, una

    def begin(self, path_off, path_steps, filter=True):
        path_off, path_steps = _i32(path_off), _i32(path_steps)
        self._pending_paths = len(path_off) - 1
        _check(self._lib.gfal_group_score_begin(
            self._h, _ptr(path_off, ctypes.c_int32), _ptr(path_steps, ctypes.c_int32),
            self._pending_paths, int(bool(filter))))

    def end(self):
        P = self._pending_paths
        bad, good, una = (np.zeros(P, np.uint32) for _ in range(3))
        _check(self._lib.gfal_group_score_end(self._h, _ptr(bad, ctypes.c_uint32), _ptr(good, ctypes.c_uint32),
                                              _ptr(una, ctypes.c_uint32)))
        return bad, good, una

    # ---- search mode: paths kept on the devices, children scored from their parents ----
    def store_reserve(self, n_slots):
        _check(self._lib.gfal_group_store_reserve(self._h, int(n_slots)))

    def score_store(self, path_off, path_steps, slots):
        """evaluate_paths(filter=True) that also keeps path p in slot slots[p] (-1: not kept)."""
        path_off, path_steps, slots = _i32(path_off), _i32(path_steps), _i32(slots)
        self._pending_paths = len(path_off) - 1
        assert len(slots) == self._pending_paths
        _check(self._lib.gfal_group_score_store_begin(
            self._h, _ptr(path_off, ctypes.c_int32), _ptr(path_steps, ctypes.c_int32), self._pending_paths,
            _ptr(slots, ctypes.c_int32)))
        return self.end()

    def score_children(self, parent, step, slot, max_path_len):
        """child i = parent[i] + step[i] (parent >= 0: store slot; < 0: ~index of an
        earlier child of this batch), kept in slot[i] (-1: not kept)."""
        parent, step, slot = _i32(parent), _i32(step), _i32(slot)
        self._pending_paths = len(parent)
        assert len(step) == len(slot) == len(parent)
        _check(self._lib.gfal_group_score_children_begin(
            self._h, len(parent), _ptr(parent, ctypes.c_int32), _ptr(step, ctypes.c_int32),
            _ptr(slot, ctypes.c_int32), int(max_path_len)))
        return self.end()

    def close(self):
        if self._h:
            self._lib.gfal_group_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
