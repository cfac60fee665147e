"""How a fragment set looks to a FOREIGN reference shard: seed hits per fragment and the L1 class counts (one MI355X).
   python tools/probe_classes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastani_amd
from fastani_amd.api import DeviceGenomes, Sketch
e = fastani_amd.engine(0)
p = e.params()
N, L = 250, 5_000_000
words = (L + 15) // 16
buf = torch.empty(N * words + 64, dtype=torch.int32, device="cuda:0")
e.synth_packed(20260925, 0, N, L, buf.data_ptr())
shard = DeviceGenomes(buf.data_ptr(), N, L, first=0, count=125)
ptr, n, own = e.sketch_records_self(p, shard, 0)
import numpy as np
sk = Sketch(e, p, records=(ptr, n, np.full(125, L, dtype=np.int32), np.arange(126, dtype=np.int32)))
_, _, foreign = e.sketch_records_self(p, DeviceGenomes(buf.data_ptr(), N, L, first=125, count=125), 0)
for name, fs in (("own set", own), ("foreign set", foreign)):
    for rep in range(2):
        e.reset_counters()
        rows = sk.map_cgi_fragset(fs, 0)
        c = e.counters()
    print(name, {k: c[k] for k in ("queryFragments", "seedHits", "l1TinyFragments", "l1MidFragments", "l1BigFragments", "l1Candidates", "l1Probes")},
          {k: round(c[k], 3) for k in ("msL1", "msL1Probe", "msL1Main", "msL1Tiny", "msL2", "msReduce")}, "rows", len(rows), "frags", fs.info())
