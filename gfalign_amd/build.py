"""In-tree build of the native pieces (hipcc for gfx950, g++ for the host CLI).

``python -m gfalign_amd.build`` builds everything; ``__graft_entry__.build()``
calls :func:`build_all`.  Outputs stay next to the sources (git-ignored, but
they travel with gpurun snapshots).

Staleness is decided by content, not by mtime: every artefact is compiled with
``-DGFAL_BUILD_ID="<sha256 of its sources + command line>"`` and carries that
string (``gfal_build_id()`` in the library, ``gfalign --build-id`` in the CLI);
the same hash is written to ``<artefact>.stamp``.  An artefact whose stamp
differs from the hash of the sources in the tree is rebuilt, so a checkout, a
copy or a snapshot can never leave a binary behind that does not match the
sources next to it (tests/test_abi.py and tests/test_gpu_parity.py compare the
embedded id with the tree).
"""
import hashlib
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

SCORER_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread"]
CLI_FLAGS = ["-O3", "-std=c++17", "-Wall", "-Wextra", "-pthread"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm is required to build the scorer)")


def source_hash(sources, flags):
    """sha256 over the flags and the (name, content) of every source file."""
    h = hashlib.sha256()
    h.update(" ".join(flags).encode())
    for path in sorted(sources, key=os.path.basename):
        h.update(b"\0" + os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:32]


def scorer_sources():
    return [os.path.join(CSRC, "scorer.hip"), os.path.join(INCLUDE, "gfalign_scorer.h")]


def cli_sources():
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".cpp")]
    hdrs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    hdrs.append(os.path.join(INCLUDE, "gfalign_scorer.h"))
    return srcs, hdrs


def scorer_build_id():
    return source_hash(scorer_sources(), SCORER_FLAGS)


def cli_build_id():
    srcs, hdrs = cli_sources()
    # the CLI links the scorer: a new library means a new CLI (gfal_info layout)
    return source_hash(srcs + hdrs, CLI_FLAGS + [scorer_build_id()])


def _stamp(target):
    try:
        with open(target + ".stamp") as f:
            return f.read().strip()
    except OSError:
        return None


def _stale(target, build_id):
    return not os.path.exists(target) or _stamp(target) != build_id


def _write_stamp(target, build_id):
    with open(target + ".stamp", "w") as f:
        f.write(build_id + "\n")


def build_scorer(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> csrc/libgfalign_scorer.so"""
    bid = scorer_build_id()
    if not force and not _stale(SCORER_SO, bid):
        return SCORER_SO
    src = scorer_sources()[0]
    cmd = [_hipcc()] + SCORER_FLAGS + ['-DGFAL_BUILD_ID="%s"' % bid, "-I", INCLUDE,
                                       "-o", SCORER_SO, src]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    _write_stamp(SCORER_SO, bid)
    return SCORER_SO


def build_cli(force=False, verbose=False):
    """g++ host driver (search / evalPath / filter CLI) linked against the scorer."""
    srcs, _ = cli_sources()
    if not srcs:
        return None
    build_scorer(force=False, verbose=verbose)
    bid = cli_build_id()
    if not force and not _stale(CLI_BIN, bid):
        return CLI_BIN
    cmd = (["g++"] + CLI_FLAGS + ['-DGFAL_BUILD_ID="%s"' % bid, "-I", INCLUDE, "-I", CSRC,
                                  "-o", CLI_BIN] + srcs +
           ["-L", CSRC, "-lgfalign_scorer", "-Wl,-rpath,$ORIGIN"])
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    _write_stamp(CLI_BIN, bid)
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
