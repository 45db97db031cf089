"""The `gfalign` command line (host driver around the C ABI).

CPU part: everything that does not score alignments -- the reference's own
golden tests test.6 (search without -g) and test.7 (filter), option handling,
the best-first order against the Python search oracle.
GPU part: search with alignments, evalPath (C.2), and
a synthetic tangle against the search oracle; speculation must not change a
byte of stdout.
"""
import os
import subprocess

import pytest

from gfalign_amd import build as gbuild
from gfalign_amd import synth
from helpers import REF_FILES, load_appendix_c
from oracle import search_oracle


@pytest.fixture(scope="module")
def cli():
    path = gbuild.build_cli()
    assert path and os.path.exists(path)
    return path


def run(cli, args, cwd=None, env=None):
    e = dict(os.environ)
    if env:
        e.update(env)
    p = subprocess.run([cli] + args, cwd=cwd, env=e, capture_output=True, text=True, timeout=600)
    return p.returncode, p.stdout, p.stderr


def tst(name):
    with open(os.path.join(REF_FILES, name)) as f:
        lines = f.read().split("\n")
    return lines[0].split()[1:], "\n".join(lines[2:])


def test_reference_test6_search_without_alignments(cli, tmp_path):
    args, expected = tst("test.6.tst")
    args = [a.replace("testFiles/", REF_FILES + "/") for a in args]
    rc, out, _ = run(cli, args)
    assert rc == 0 and out == expected


def test_reference_test7_filter(cli, tmp_path):
    args, expected = tst("test.7.tst")
    args = [a.replace("testFiles/", REF_FILES + "/") for a in args]
    rc, out, _ = run(cli, args, cwd=str(tmp_path))     # writes ./gaf like the reference
    assert rc == 0 and out == expected
    with open(os.path.join(REF_FILES, "test.7.output.gaf")) as f:
        assert (tmp_path / "gaf").read_text() == f.read()


@pytest.mark.parametrize("name", ["test.0.tst", "test.2.tst", "test.3.tst", "test.5.tst"])
def test_reference_evalgfa_summaries(cli, name):
    """validateFiles/test.{0,2,3,5}.tst: evalGFA alignment summary on random1 /
    random2, with and without --sort-alignment (test.1 / test.4 need the
    gfastats report of the absent gfalibs and are out of scope)."""
    args, expected = tst(name)
    args = [a.replace("testFiles/", REF_FILES + "/") for a in args]
    rc, out, _ = run(cli, args)
    assert rc == 0 and out == expected


def _edge_weights(gaf_paths, ids):
    """reference src/alignments.cpp:353-403 restated: adjacency lists of
    [from_orient, to, to_orient, weight]; returns them."""
    flip = {"+": "-", "-": "+"}
    adj = {v: [] for v in ids.values()}

    def find(lst, e):
        for x in lst:
            if x[:3] == e[:3]:
                return x
        return None

    for path in gaf_paths:
        nodes, i = [], 0
        while i < len(path):
            j = i + 1
            while j < len(path) and path[j] not in "<>":
                j += 1
            nodes.append((ids.get(path[i + 1:j], 0), "+" if path[i] == ">" else "-"))
            i = j
        for (a, ao), (b, bo) in zip(nodes, nodes[1:]):
            fw, rv = [ao, b, bo, 1], [flip[bo], a, flip[ao], 1]
            hit = find(adj[a], fw)
            if hit is None:
                adj[a].append(fw)
                if find(adj[b], rv) is None:
                    adj[b].append(rv)
            else:
                hit[3] += 1
                find(adj[b], rv)[3] += 1
    return adj


def test_evalgfa_tags_links_with_read_counts(cli, tmp_path):
    """evalGFA -o: RC:i tag per L line = how many alignments walk the link in
    either direction (src/eval.cpp:34-61); a link that is its own reverse
    (a+ -> a-) counts twice from the second sighting on, as in the reference.
    The GFA text around the tag is this build's (gfalibs' writer is absent)."""
    gfa = tmp_path / "g.gfa"
    gaf = tmp_path / "a.gaf"
    links = [("a", "+", "b", "+"), ("b", "+", "c", "-"), ("a", "+", "a", "-"), ("c", "+", "a", "+"),
             ("b", "-", "a", "-")]
    gfa.write_text("H\tVN:Z:1.0\n" + "".join("S\t%s\tACGT\n" % n for n in "abc") +
                   "".join("L\t%s\t%s\t%s\t%s\t0M\n" % l for l in links))
    paths = [">a>b<c", ">c<b<a", ">a<a", ">a<a", ">a<a>b", "<b<a", ">c", ">b>zzz"]
    gaf.write_text("".join("r%d\t10\t0\t10\t+\t%s\t30\t0\t10\t10\t10\t60\n" % (k, p)
                           for k, p in enumerate(paths)))
    out = tmp_path / "tagged.gfa"
    rc, stdout, err = run(cli, ["evalGFA", "-f", str(gfa), "-g", str(gaf), "-o", str(out)])
    assert rc == 0, err
    assert stdout.startswith("+++Alignment summary+++")
    ids = {"a": 0, "b": 1, "c": 2}
    adj = _edge_weights(paths, ids)
    got = [l.split("\t") for l in out.read_text().splitlines()]
    assert [l for l in got if l[0] != "L"] == [l.split("\t") for l in gfa.read_text().splitlines()
                                               if not l.startswith("L")]
    tagged = [l for l in got if l[0] == "L"]
    assert len(tagged) == len(links)
    for l, (a, ao, b, bo) in zip(tagged, links):
        hit = [e for e in adj[ids[a]] if e[:3] == [ao, ids[b], bo]]
        want = hit[0][3] if hit else 0
        assert l[:6] == ["L", a, ao, b, bo, "0M"] and l[6] == "RC:i:%d" % want, (l, want)
    # spot values: a+b+ walked by >a>b<c, >a<a>b and, backwards, by <b<a; the
    # self-reverse link a+a- three times = 1 + 2 + 2
    assert tagged[0][6] == "RC:i:3" and tagged[2][6] == "RC:i:5" and tagged[3][6] == "RC:i:0"
    # -o naming a format writes to stdout after the summary
    rc, stdout, err = run(cli, ["evalGFA", "-f", str(gfa), "-g", str(gaf), "-o", "gfa"])
    assert rc == 0 and stdout.splitlines()[-len(got):] == out.read_text().splitlines()


def test_filter_min_nodes(cli, tmp_path):
    rc, out, _ = run(cli, ["filter", "-g", REF_FILES + "/random3.gaf", "-n",
                           REF_FILES + "/random3.filter_nodelist.ls", "-o", "x.gaf",
                           "--min-nodes", "3"], cwd=str(tmp_path))
    assert rc == 0
    rows = (tmp_path / "x.gaf").read_text().splitlines()
    assert [r.split("\t")[5] for r in rows] == [">1>2>3>4", ">4>3>2>1"]
    assert "# alignments: 2" in out


def test_step_cap_message_and_atoi(cli):
    base = ["search", "-f", REF_FILES + "/random3.gfa", "-n",
            REF_FILES + "/random3.search_nodelist.tsv", "-s", "1", "-d", "4"]
    rc, out, _ = run(cli, base + ["-m", "2"])
    assert out.splitlines()[-1] == "Reached maximum number of steps (2)"
    # reference parses -m with atoi: "1e9" means 1 step (src/main.cpp:462-464)
    rc, out, _ = run(cli, base + ["-m", "1e9"])
    assert out.splitlines()[-1] == "Reached maximum number of steps (1)"


def test_unknown_mode_and_missing_file(cli):
    rc, _, err = run(cli, ["frobnicate"])
    assert rc != 0 and "mode 'frobnicate' does not exist. Terminating." in err
    rc, _, err = run(cli, ["search", "-f", "/nonexistent.gfa"])
    assert rc != 0


def _write_tangle(t, d):
    t.write_gfa(os.path.join(d, "g.gfa"))
    t.write_nodelist(os.path.join(d, "nodes.tsv"))
    t.write_gaf(os.path.join(d, "a.gaf"))


def test_search_order_matches_oracle_without_alignments(cli, tmp_path):
    """Best-first order, budgets, uniques, Hamiltonian flag, printing rule."""
    t = synth.Tangle(V=14, n_T=12, N=10, P=1, seed=21)
    _write_tangle(t, str(tmp_path))
    args = dict(gfa=str(tmp_path / "g.gfa"), node_file=str(tmp_path / "nodes.tsv"),
                source="utig4-0", destination="utig4-13", max_steps=400)
    for extra_cli, extra in ((["--return-all-paths"], dict(return_all_paths=True)),
                             (["--min-nodes", "5"], dict(min_nodes=5)), ([], {})):
        exp = search_oracle.search(**args, **extra)
        rc, out, _ = run(cli, ["search", "-f", args["gfa"], "-n", args["node_file"], "-s",
                               args["source"], "-d", args["destination"], "-m", "400"] + extra_cli)
        assert rc == 0 and out.splitlines() == exp
        assert len(exp) >= (2 if extra.get("return_all_paths") else 1)
        if not extra:
            assert exp[0].split("\t")[6] == "T"      # a Hamiltonian path is flagged


def test_search_needs_a_gpu_when_there_are_alignments(cli):
    from gfalign_amd import scorer
    if scorer.device_count() > 0:
        pytest.skip("a GPU is visible here")
    rc, out, err = run(cli, ["search", "-f", REF_FILES + "/random3.gfa", "-g",
                             REF_FILES + "/random3.gaf", "-n",
                             REF_FILES + "/random3.search_nodelist.tsv", "-s", "1", "-d", "4"])
    assert rc != 0 and "no usable HIP device" in err and out == ""


# ---------------------------------------------------------------- GPU ----

@pytest.mark.gpu
def test_search_with_alignments_appendix_c1(cli, gpu):
    gold = load_appendix_c()["search_with_gaf"]
    base = ["search", "-f", REF_FILES + "/random3.gfa", "-g", REF_FILES + "/random3.gaf",
            "-n", REF_FILES + "/random3.search_nodelist.tsv", "-s", "1", "-d", "4"]
    rc, out, err = run(cli, base)
    assert rc == 0, err
    assert out.splitlines() == gold["stdout"]
    rc, out, _ = run(cli, base + ["--return-all-paths"])
    assert out.splitlines() == gold["stdout"] + [gold["return_all_paths_extra_row"]]


@pytest.mark.gpu
def test_eval_path_appendix_c2(cli, gpu):
    gold = load_appendix_c()["eval_path"]
    rc, out, err = run(cli, ["evalPath", "-f", REF_FILES + "/random3.gfa", "-g",
                             REF_FILES + "/random3.gaf", "-p", gold["path"]])
    assert rc == 0, err
    assert out.splitlines() == gold["stdout"]
    rc, out, err = run(cli, ["evalPath", "-f", REF_FILES + "/random3.gfa", "-g",
                             REF_FILES + "/random3.gaf", "-p", gold["path"], "--devices", "2"],
                       env={"GFALIGN_SHARE_DEVICE": "1"})
    assert rc == 0, err
    assert out.splitlines() == gold["stdout"]
    rc, out, err = run(cli, ["evalPath", "-f", REF_FILES + "/random3.gfa", "-g",
                             REF_FILES + "/random3.gaf", "-p", gold["path"]],
                       env={"GFALIGN_DEDUP": "0"})      # (one lane per alignment; the default collapses identical ones)
    assert rc == 0, err
    assert out.splitlines() == gold["stdout"]


@pytest.mark.gpu
def test_search_on_a_synthetic_tangle_matches_oracle(cli, gpu, tmp_path):
    t = synth.Tangle(V=30, n_T=24, N=400, P=1, seed=22)
    _write_tangle(t, str(tmp_path))
    gfa, nodes, gaf = (str(tmp_path / n) for n in ("g.gfa", "nodes.tsv", "a.gaf"))
    exp = search_oracle.search(gfa, nodes, "utig4-0", "utig4-29", gaf=gaf, max_steps=150,
                               return_all_paths=True)
    base = ["search", "-f", gfa, "-g", gaf, "-n", nodes, "-s", "utig4-0", "-d", "utig4-29",
            "-m", "150", "--return-all-paths"]
    outs = []
    for spec in ("1", "7", "512"):
        rc, out, err = run(cli, base, env={"GFALIGN_SPECULATE": spec})
        assert rc == 0, err
        outs.append(out)
    assert outs[0] == outs[1] == outs[2]          # batching never changes the output
    # candidates scored from their parents on the device (the default) or all in full
    for spec in ("1", "128"):
        rc, out, err = run(cli, base, env={"GFALIGN_SPECULATE": spec, "GFALIGN_INCREMENTAL": "0"})
        assert rc == 0, err
        assert out == outs[0]
    rc, out, err = run(cli, base + ["--verbose"], env={"GFALIGN_SPECULATE": "128"})
    assert rc == 0 and out == outs[0], err
    import re
    m = re.search(r"scored (\d+) candidate paths in \d+ batches, (\d+) of them in full", err)
    assert m and int(m.group(2)) < int(m.group(1)), err       # the search did use the shortcut
    # nor does keeping a further batch in flight while the host pops
    for spec in ("7", "128"):
        rc, out, err = run(cli, base, env={"GFALIGN_SPECULATE": spec, "GFALIGN_PREFETCH": "1"})
        assert rc == 0, err
        assert out == outs[0]
    # nor does the speculation policy (best-first by the alignments' support of the edges)
    for spec in ("7", "128"):
        rc, out, err = run(cli, base, env={"GFALIGN_SPECULATE": spec, "GFALIGN_SPEC_POLICY": "best"})
        assert rc == 0, err
        assert out == outs[0]
    # alignments sharded over three scorers (all on this box's one GPU): same bytes
    rc, out, err = run(cli, base + ["--devices", "3"], env={"GFALIGN_SHARE_DEVICE": "1"})
    assert rc == 0, err
    assert out == outs[0]
    # one lane per alignment instead of identical alignments collapsed into weighted lanes
    # (the default), sharded or not: same bytes
    for extra in ([], ["--devices", "2"]):
        rc, out, err = run(cli, base + extra, env={"GFALIGN_SHARE_DEVICE": "1", "GFALIGN_DEDUP": "0"})
        assert rc == 0, err
        assert out == outs[0]
    assert outs[0].splitlines() == exp
    assert any(int(r.split("\t")[2]) > 0 for r in exp[:-1])   # non-zero good counters


@pytest.mark.gpu
def test_config5_flow_search_then_eval_path(cli, gpu, tmp_path):
    """BASELINE config 5 at reduced size: `search` with the step cap given as an
    integer literal (the reference parses -m with atoi, so "1e9" would mean one
    step: src/main.cpp:462-464), alignments sharded over two scorers, then
    `evalPath` on the best path the search printed (README.md:36-40 workflow).
    The search rows are checked against oracle/search_oracle.py, the evalPath
    rows and summary against oracle.pair_scores / oracle.evaluate_paths."""
    import numpy as np
    import oracle
    t = synth.Tangle(V=20, n_T=16, N=1500, P=1, seed=25)      # (the search runs until its queue is empty)
    _write_tangle(t, str(tmp_path))
    gfa, nodes, gaf = (str(tmp_path / n) for n in ("g.gfa", "nodes.tsv", "a.gaf"))
    exp = search_oracle.search(gfa, nodes, "utig4-0", "utig4-19", gaf=gaf, max_steps=1000000000)
    rc, out, err = run(cli, ["search", "-f", gfa, "-g", gaf, "-n", nodes, "-s", "utig4-0", "-d", "utig4-19",
                             "-m", "1000000000", "--devices", "2"], env={"GFALIGN_SHARE_DEVICE": "1"})
    assert rc == 0, err
    rows = out.splitlines()
    assert rows == exp and len(rows) >= 1 and "\t" in rows[-1]
    best = rows[-1].split("\t")
    path_text = best[7]
    # evalPath on that path: every alignment (filter off), one row each + summary
    rc, out, err = run(cli, ["evalPath", "-f", gfa, "-g", gaf, "-p", path_text, "--devices", "2"],
                       env={"GFALIGN_SHARE_DEVICE": "1"})
    assert rc == 0, err
    lines = out.splitlines()
    assert lines[0] == path_text and len(lines) == t.N + 2
    ids = {"utig4-%d" % k: k for k in range(t.V)}
    path = [(ids[c[:-1]] << 1) | int(c[-1] == "-") for c in path_text.split(",")]
    fw, rcs = oracle.pair_scores(t.aln_off, t.aln_steps, path)
    got_scores = [int(l.split("\t")[2]) for l in lines[1:-1]]
    assert got_scores == np.maximum(fw, rcs).tolist()
    bad, good, _ = oracle.evaluate_paths(t.aln_off, t.aln_steps, [0, len(path)], path, False)
    uniques = len(set(s >> 1 for s in path))
    assert lines[-1] == "%d\t%d\t%d\t%d\t%d" % (bad[0], good[0], int(bad[0]) - int(good[0]) - uniques,
                                                 len(path), uniques)
    # the search row (filter on) and the evalPath summary (filter off) differ only
    # by the alignments with a node off the path
    assert int(best[2]) <= int(good[0]) + int(bad[0])


@pytest.mark.gpu
def test_config5_flow_at_size(cli, gpu, tmp_path):
    """The config-5 shape on the box: a 5 000-node tangle, 300 000 alignments, `search -m
    1000000000` bounded by a node list of the walk's first nodes (the queue runs empty),
    alignments sharded over two scorers; the rows must not depend on how candidates are
    scored (from their parents, the default, or every one in full), and `evalPath` on the
    best path must print the oracle's score for every one of the 300 000 alignments."""
    import numpy as np
    import oracle
    t = synth.Tangle(V=5000, n_T=1000, N=300000, P=1, seed=33)
    d = str(tmp_path)
    t.write_gfa(d + "/g.gfa")
    t.write_gaf(d + "/a.gaf")
    K = 28                                     # the search may only walk the truth walk's first steps
    head = [int(x) >> 1 for x in t.T[:K]]
    with open(d + "/nodes.tsv", "w") as f:
        for node in dict.fromkeys(head):
            f.write("utig4-%d\t%d\n" % (node, head.count(node)))
    src, dst = "utig4-%d" % head[0], "utig4-%d" % head[-1]
    base = ["search", "-f", d + "/g.gfa", "-g", d + "/a.gaf", "-n", d + "/nodes.tsv", "-s", src, "-d", dst,
            "-m", "1000000000", "--devices", "2", "--verbose"]
    outs = []
    for env in ({"GFALIGN_SHARE_DEVICE": "1"}, {"GFALIGN_SHARE_DEVICE": "1", "GFALIGN_INCREMENTAL": "0"}):
        rc, out, err = run(cli, base, env=env)
        assert rc == 0, err[-2000:]
        outs.append(out)
        assert "Reached maximum" not in out
    assert outs[0] == outs[1] and outs[0].count("\n") >= 1
    best = outs[0].splitlines()[-1].split("\t")
    assert int(best[2]) > 0                    # non-zero good counter
    path_text = best[7]
    rc, out, err = run(cli, ["evalPath", "-f", d + "/g.gfa", "-g", d + "/a.gaf", "-p", path_text, "--devices", "2"],
                       env={"GFALIGN_SHARE_DEVICE": "1"})
    assert rc == 0, err[-2000:]
    lines = out.splitlines()
    assert lines[0] == path_text and len(lines) == t.N + 2
    ids = {"utig4-%d" % k: k for k in range(t.V)}
    path = [(ids[c[:-1]] << 1) | int(c[-1] == "-") for c in path_text.split(",")]
    fw, rcs = oracle.pair_scores(t.aln_off, t.aln_steps, path)
    got_scores = np.array([int(l.rsplit("\t", 1)[1]) for l in lines[1:-1]])
    assert np.array_equal(got_scores, np.maximum(fw, rcs))
    bad, good, _ = oracle.evaluate_paths(t.aln_off, t.aln_steps, [0, len(path)], path, False)
    uniques = len(set(x >> 1 for x in path))
    assert lines[-1] == "%d\t%d\t%d\t%d\t%d" % (bad[0], good[0], int(bad[0]) - int(good[0]) - uniques,
                                                 len(path), uniques)


# ---- CLI behaviours of the reference beyond its .tst files (CPU) ----

def test_graph_statistics_is_refused_not_ignored(cli):
    """validateFiles/test.1.tst / test.4.tst need gfalibs' gfastats report, which
    is not in the reference tree: the flag fails loudly instead of printing a
    summary without the '+++Assembly summary+++' block."""
    rc, out, err = run(cli, ["evalGFA", "-f", REF_FILES + "/random1.gfa", "-g",
                             REF_FILES + "/random1.gaf", "--graph-statistics"])
    assert rc != 0 and out == "" and "not part of this build" in err


def test_piped_gfa_input(cli):
    """`-f -` reads the graph from stdin (reference src/main.cpp:425-429, 443-449)."""
    args, expected = tst("test.6.tst")
    args = [a.replace("testFiles/", REF_FILES + "/") for a in args]
    gfa = args[args.index("-f") + 1]
    args[args.index("-f") + 1] = "-"
    with open(gfa, "rb") as f:
        p = subprocess.run([cli] + args, stdin=f, capture_output=True, timeout=120)
    assert p.returncode == 0 and p.stdout.decode() == expected


def test_unknown_source_aliases_node_zero(cli):
    """reference src/eval.cpp:127-128: headersToIds[source] default-inserts, so a
    name that is not in the graph stands for uId 0 (the first S line).  random3's
    first segment is '1': the search from 'nope' gives the rows of the search
    from '1' (path strings come from the uIds)."""
    args, expected = tst("test.6.tst")
    args = [a.replace("testFiles/", REF_FILES + "/") for a in args]
    args[args.index("-s") + 1] = "nope"
    rc, out, _ = run(cli, args)
    assert rc == 0 and out == expected


def test_threads_option_is_accepted(cli, tmp_path):
    """-j sets the reader threads (reference src/main.cpp:472-474); any value
    gives the same bytes."""
    args, expected = tst("test.7.tst")
    args = [a.replace("testFiles/", REF_FILES + "/") for a in args]
    for j in ("1", "3"):
        rc, out, _ = run(cli, args + ["-j", j], cwd=str(tmp_path))
        assert rc == 0 and out == expected
