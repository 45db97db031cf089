"""CPU oracle for the gfalign path-scoring hot path -- TEST INFRASTRUCTURE.

Loads ``oracle/libgfalign_oracle.so`` (plain-C restatement of reference
src/eval.cpp:67-108 and src/alignments.cpp:499-561; see gfalign_oracle.h for
the parity status).  Only ``tests/``, ``bench.py``'s cpu_baseline leg and
``__graft_entry__.smoke()`` may import this package; the product package
``gfalign_amd`` never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgfalign_oracle.so")
_lib = None

INT32_MIN = -(2 ** 31)


def build(force=False):
    """Compile the oracle with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "gfalign_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s", "libgfalign_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        i32p = ctypes.POINTER(ctypes.c_int32)
        u32p = ctypes.POINTER(ctypes.c_uint32)
        L.gfo_evaluate_paths_packed.restype = ctypes.c_int
        L.gfo_evaluate_paths_packed.argtypes = [
            i32p, i32p, ctypes.c_int64, i32p, i32p, ctypes.c_int32,
            ctypes.c_int, u32p, u32p, u32p]
        L.gfo_pair_scores_packed.restype = ctypes.c_int
        L.gfo_pair_scores_packed.argtypes = [
            i32p, i32p, ctypes.c_int64, i32p, ctypes.c_int32, i32p, i32p]
        _lib = L
    return _lib


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, ty):
    return a.ctypes.data_as(ctypes.POINTER(ty))


def evaluate_paths(aln_off, aln_steps, path_off, path_steps, filter=True):
    """evaluatePath for every packed path -> (bad, good, unaligned) uint32 arrays."""
    aln_off, aln_steps = _i32(aln_off), _i32(aln_steps)
    path_off, path_steps = _i32(path_off), _i32(path_steps)
    n_aln, n_paths = len(aln_off) - 1, len(path_off) - 1
    bad = np.zeros(n_paths, np.uint32)
    good = np.zeros(n_paths, np.uint32)
    una = np.zeros(n_paths, np.uint32)
    rc = lib().gfo_evaluate_paths_packed(
        _p(aln_off, ctypes.c_int32), _p(aln_steps, ctypes.c_int32), n_aln,
        _p(path_off, ctypes.c_int32), _p(path_steps, ctypes.c_int32), n_paths,
        int(bool(filter)), _p(bad, ctypes.c_uint32), _p(good, ctypes.c_uint32),
        _p(una, ctypes.c_uint32))
    if rc:
        raise ValueError("oracle rejected the input (status %d)" % rc)
    return bad, good, una


def pair_scores(aln_off, aln_steps, path_steps):
    """Traceback scores (fw, rc) of every alignment against one packed path."""
    aln_off, aln_steps, path_steps = _i32(aln_off), _i32(aln_steps), _i32(path_steps)
    n_aln = len(aln_off) - 1
    fw = np.zeros(n_aln, np.int32)
    rv = np.zeros(n_aln, np.int32)
    rc = lib().gfo_pair_scores_packed(
        _p(aln_off, ctypes.c_int32), _p(aln_steps, ctypes.c_int32), n_aln,
        _p(path_steps, ctypes.c_int32), len(path_steps),
        _p(fw, ctypes.c_int32), _p(rv, ctypes.c_int32))
    if rc:
        raise ValueError("oracle rejected the input (status %d)" % rc)
    return fw, rv


# ---- the kernels' decision rule as a multi-threaded CPU program (gfalign_fast.c) ----
_FAST_PATH = os.path.join(_HERE, "libgfalign_fast.so")
_fast = None


def fast_lib():
    global _fast
    if _fast is None:
        src = os.path.join(_HERE, "gfalign_fast.c")
        if not os.path.exists(_FAST_PATH) or os.path.getmtime(_FAST_PATH) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-s", "libgfalign_fast.so"])
        L = ctypes.CDLL(_FAST_PATH)
        i32p = ctypes.POINTER(ctypes.c_int32)
        u32p = ctypes.POINTER(ctypes.c_uint32)
        L.gfo_fast_evaluate_paths_packed.restype = ctypes.c_int
        L.gfo_fast_evaluate_paths_packed.argtypes = [
            i32p, i32p, ctypes.c_int64, i32p, i32p, ctypes.c_int32, ctypes.c_int, ctypes.c_int,
            u32p, u32p, u32p]
        _fast = L
    return _fast


def fast_evaluate_paths(aln_off, aln_steps, path_off, path_steps, filter=True, threads=0):
    """Same result as evaluate_paths(), computed with the kernels' decision rule
    on `threads` host cores (0 = all)."""
    aln_off, aln_steps = _i32(aln_off), _i32(aln_steps)
    path_off, path_steps = _i32(path_off), _i32(path_steps)
    n_aln, n_paths = len(aln_off) - 1, len(path_off) - 1
    bad = np.zeros(n_paths, np.uint32)
    good = np.zeros(n_paths, np.uint32)
    una = np.zeros(n_paths, np.uint32)
    rc = fast_lib().gfo_fast_evaluate_paths_packed(
        _p(aln_off, ctypes.c_int32), _p(aln_steps, ctypes.c_int32), n_aln,
        _p(path_off, ctypes.c_int32), _p(path_steps, ctypes.c_int32), n_paths,
        int(bool(filter)), int(threads), _p(bad, ctypes.c_uint32), _p(good, ctypes.c_uint32),
        _p(una, ctypes.c_uint32))
    if rc:
        raise ValueError("fast CPU scorer rejected the input (status %d)" % rc)
    return bad, good, una
