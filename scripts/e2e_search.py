"""End-to-end `gfalign search` on a synthetic tangle (GPU box): wall time of the
CLI for a few step budgets and speculation widths."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfalign_amd import synth, build

cfg = sys.argv[1] if len(sys.argv) > 1 else "config2"
t = synth.make(cfg)
d = "/tmp/e2e_" + cfg
os.makedirs(d, exist_ok=True)
t0 = time.time()
t.write_gfa(d + "/g.gfa"); t.write_nodelist(d + "/nodes.tsv"); t.write_gaf(d + "/a.gaf")
print("wrote files in %.1f s (GAF %.1f MB)" % (time.time() - t0, os.path.getsize(d + "/a.gaf") / 1e6))
cli = build.build_cli()
base = [cli, "search", "-f", d + "/g.gfa", "-g", d + "/a.gaf", "-n", d + "/nodes.tsv",
        "-s", "utig4-0", "-d", "utig4-%d" % (t.V - 1), "--verbose"]
for steps in (int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["200", "2000"])):
    for spec in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["1", "64", "512", "4096"]):
        t0 = time.time()
        p = subprocess.run(base + ["-m", str(steps)], env=dict(os.environ, GFALIGN_SPECULATE=spec),
                           capture_output=True, text=True)
        dt = time.time() - t0
        rows = p.stdout.strip().splitlines()
        print("steps %5d spec %5s: %.2f s  rows %d  last: %s | %s" % (
            steps, spec, dt, len(rows), rows[-1][:60] if rows else "", " / ".join(p.stderr.strip().splitlines()[-2:])))
