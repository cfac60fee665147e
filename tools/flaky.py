"""Repeat one map_cgi_batch many times and count the runs whose rows differ from the first (GPU; diagnostics for races)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fastani_amd import _lib, api
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n, L = int(os.environ.get("FLAKY_N", 24)), 5_000_000
e = api.Engine(_lib.load(), 0)
words = (L + 15) // 16
buf = torch.zeros(n * words + 64, dtype=torch.int32, device="cuda:0")
torch.cuda.synchronize()
e.synth_packed(99, 0, n, L, buf.data_ptr())
dg = api.DeviceGenomes(buf.data_ptr(), n, L)
p = e.params()
sk = api.Sketch(e, p, dg)
KEYS = ("queryFragments", "querySketchHashes", "seedHits", "l1Candidates", "l2WindowEntries", "l2Steps", "l2QueryHashes", "l2FastCandidates", "l2SlowCandidates", "cgiRows")
e.reset_counters()
first = sk.map_cgi_batch(dg, 0)
c0 = {k: e.counters()[k] for k in KEYS}
print(c0, flush=True)
bad = 0
for i in range(reps):
    e.reset_counters()
    r = sk.map_cgi_batch(dg, 0)
    c1 = {k: e.counters()[k] for k in KEYS}
    if c1 != c0:
        print("run", i, "counters differ:", {k: (c0[k], c1[k]) for k in KEYS if c0[k] != c1[k]}, flush=True)
    if not np.array_equal(r, first):
        bad += 1
        if bad <= 3:
            m = min(len(r), len(first)); d = np.nonzero(r[:m] != first[:m])[0]
            print("run", i, "differs: len", len(r), len(first), "first diffs", [(tuple(first[j]), tuple(r[j])) for j in d[:3]], flush=True)
print("chunks", len(sk.chunks()), "rows", len(first), "bad", bad, "of", reps, {k: os.environ[k] for k in os.environ if k.startswith("ANI_")}, flush=True)
