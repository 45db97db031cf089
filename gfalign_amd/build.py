"""In-tree build of the native pieces (hipcc for gfx950, g++ for the host CLI).

``python -m gfalign_amd.build`` builds everything; ``__graft_entry__.build()``
calls :func:`build_all`.  Outputs stay next to the sources (git-ignored, but
they travel with gpurun snapshots).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
SCORER_SO = os.path.join(CSRC, "libgfalign_scorer.so")
CLI_BIN = os.path.join(CSRC, "gfalign")


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm is required to build the scorer)")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_scorer(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> csrc/libgfalign_scorer.so"""
    src = os.path.join(CSRC, "scorer.hip")
    hdr = os.path.join(INCLUDE, "gfalign_scorer.h")
    if not force and not _stale(SCORER_SO, [src, hdr]):
        return SCORER_SO
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-pthread", "-I", INCLUDE, "-o", SCORER_SO, src]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return SCORER_SO


def build_cli(force=False, verbose=False):
    """g++ host driver (search / evalPath / filter CLI) linked against the scorer."""
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))
            if f.endswith(".cpp")]
    if not srcs:
        return None
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(INCLUDE, "gfalign_scorer.h"))
    build_scorer(force=False, verbose=verbose)
    if not force and not _stale(CLI_BIN, srcs + hdrs + [SCORER_SO]):
        return CLI_BIN
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-I", INCLUDE, "-I", CSRC,
           "-o", CLI_BIN] + srcs + ["-L", CSRC, "-lgfalign_scorer",
                                    "-Wl,-rpath,$ORIGIN", "-pthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return CLI_BIN


def build_all(force=False, verbose=False):
    out = [build_scorer(force, verbose)]
    cli = build_cli(force, verbose)
    if cli:
        out.append(cli)
    return out


if __name__ == "__main__":
    for p in build_all(force="--force" in sys.argv, verbose=True):
        print(p)
