"""Write the GFA / node list / GAF of a synthetic config under a directory."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gfalign_amd import synth
cfg, d = sys.argv[1], sys.argv[2]
os.makedirs(d, exist_ok=True)
t = synth.make(cfg)
t.write_gfa(d + "/g.gfa"); t.write_nodelist(d + "/nodes.tsv"); t.write_gaf(d + "/a.gaf")
print(t.V - 1)
