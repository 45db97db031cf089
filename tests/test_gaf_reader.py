"""CPU: the parallel path-only GAF reader that `search` uses
(gfalign_amd/csrc/graph_io.h read_gaf_paths) against the record reader +
PackedAlignments::add, on the reference's own test files, a synthetic tangle
and malformed input (same acceptance, same error text)."""
import os
import subprocess

import pytest

from gfalign_amd import synth
from helpers import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gfalign_amd", "csrc")


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("native") / "check_gaf_reader")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", CSRC,
                           "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "check_gaf_reader.cpp"), "-o", exe])
    return exe


def run(checker, gfa, gaf, threads):
    p = subprocess.run([checker, gfa, gaf, str(threads)], capture_output=True, text=True, timeout=120)
    return p.returncode, p.stdout.strip()


@pytest.mark.parametrize("name", ["random1", "random2", "random3"])
@pytest.mark.parametrize("threads", [1, 3])
def test_reference_test_files(checker, name, threads):
    d = os.path.join(GOLDEN, "reference_testfiles")
    gaf = os.path.join(d, name + ".gaf")
    gfa = os.path.join(d, name + ".gfa")
    if not (os.path.exists(gaf) and os.path.exists(gfa)):
        pytest.skip("no such pair among the committed reference test files")
    rc, out = run(checker, gfa, gaf, threads)
    assert rc == 0 and out.startswith("SAME"), out


@pytest.mark.parametrize("threads", [1, 2, 7])
def test_synthetic_tangle(checker, tmp_path, threads):
    t = synth.make("smoke")
    gfa, gaf = str(tmp_path / "g.gfa"), str(tmp_path / "a.gaf")
    t.write_gfa(gfa)
    t.write_gaf(gaf)
    # the reader cuts at 1 MiB per thread at least: make the file big enough
    # for several pieces by repeating it
    body = open(gaf).read()
    with open(gaf, "w") as f:
        while f.tell() < (threads + 1) * (1 << 20):
            f.write(body)
    rc, out = run(checker, gfa, gaf, threads)
    assert rc == 0 and out.startswith("SAME"), out
    assert int(out.split()[1]) >= t.N


GOOD = "r1\t100\t0\t100\t+\t>a<b>unknown\t300\t5\t105\t90\t100\t60\ttp:A:P"


@pytest.mark.parametrize("line,same_error", [
    (GOOD, None),
    (GOOD.replace("\t60\ttp:A:P", ""), "fewer than 12 columns"),
    (GOOD.replace("\t300\t", "\tx300\t"), "malformed"),
    (GOOD.replace("\t300\t", "\t99999999999\t"), "malformed"),
    (GOOD.replace("\t300\t", "\t 300abc\t"), None),          # std::stoi takes the prefix
    (GOOD + "\n\n" + GOOD, "fewer than 12 columns"),          # an empty line is a short record
    (GOOD.replace(">a<b>unknown", "a"), None),                # no marker: first char is eaten
    (GOOD + "\r", None),
])
def test_acceptance_and_errors_match(checker, tmp_path, line, same_error):
    gfa, gaf = str(tmp_path / "g.gfa"), str(tmp_path / "a.gaf")
    with open(gfa, "w") as f:
        f.write("S\ta\t*\nS\tb\t*\nL\ta\t+\tb\t-\t0M\n")
    with open(gaf, "w") as f:
        f.write(line + "\n")
    rc, out = run(checker, gfa, gaf, 2)
    assert rc == 0, out
    if same_error:
        assert out.startswith("SAME error") and same_error in out, out
    else:
        assert out.startswith("SAME") and "error" not in out, out


def test_empty_file(checker, tmp_path):
    gfa, gaf = str(tmp_path / "g.gfa"), str(tmp_path / "a.gaf")
    open(gfa, "w").write("S\ta\t*\n")
    open(gaf, "w").close()
    rc, out = run(checker, gfa, gaf, 2)
    assert rc == 0 and out.startswith("SAME 0 records"), out
