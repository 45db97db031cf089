"""BASELINE config 5 end to end on ONE GPU box: 5 000-node tangle, 10 M
alignments: `gfalign search` with a step budget, then `gfalign evalPath` on the
best path found (the reference workflow of README.md:36-40).  Prints wall times."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfalign_amd import synth, build

cfg = sys.argv[1] if len(sys.argv) > 1 else "config5"
budget = sys.argv[2] if len(sys.argv) > 2 else "20000"
t = synth.make(cfg)
d = "/tmp/e2e_" + cfg
os.makedirs(d, exist_ok=True)
t0 = time.time()
t.write_gfa(d + "/g.gfa"); t.write_nodelist(d + "/nodes.tsv"); t.write_gaf(d + "/a.gaf")
print("wrote files in %.1f s (GAF %.1f MB, %d alignments)" % (
    time.time() - t0, os.path.getsize(d + "/a.gaf") / 1e6, t.N), flush=True)
cli = build.build_cli()
t0 = time.time()
p = subprocess.run([cli, "search", "-f", d + "/g.gfa", "-g", d + "/a.gaf", "-n", d + "/nodes.tsv",
                    "-s", "utig4-0", "-d", "utig4-%d" % (t.V - 1), "-m", budget, "--verbose"],
                   capture_output=True, text=True)
print("search -m %s: %.2f s, rc %d" % (budget, time.time() - t0, p.returncode))
for line in p.stderr.strip().splitlines()[-2:]:
    print("  " + line)
rows = [r for r in p.stdout.strip().splitlines() if "\t" in r]
print("  %d path rows; best: %s" % (len(rows), "\t".join(rows[-1].split("\t")[:7]) if rows else "-"), flush=True)
if rows:
    best = rows[-1].split("\t")[7]
    t0 = time.time()
    # the output is ~n x 11 bytes per alignment (94 GB for 10 M alignments and a
    # 858-step path): counted and checksummed on the fly, not stored
    q = subprocess.Popen([cli, "evalPath", "-f", d + "/g.gfa", "-g", d + "/a.gaf", "-p", best],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    n_bytes = n_lines = 0
    last = b""
    tail = b""
    while True:
        chunk = q.stdout.read(1 << 24)
        if not chunk:
            break
        n_bytes += len(chunk)
        n_lines += chunk.count(b"\n")
        tail = (tail + chunk)[-400:]
    err = q.stderr.read().decode()
    rc = q.wait()
    dt = time.time() - t0
    summary = tail.decode(errors="replace").rstrip("\n").split("\n")[-1]
    print("evalPath on the best path (%d steps): %.1f s, rc %d, %d lines, %.1f GB of rows, summary: %s %s" % (
        best.count(",") + 1, dt, rc, n_lines, n_bytes / 1e9, summary, err.strip()[-200:]))
    srow = rows[-1].split("\t")
    print("  search row said bad %s good %s (filter on); evalPath counts every alignment (filter off)" % (srow[1], srow[2]))
