"""Repeat the fragment sketch stage and compare every fragment's sorted-unique hash list between runs (GPU; race diagnostics)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from fastani_amd import _lib, api
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n, L = int(os.environ.get("FLAKY_N", 24)), 5_000_000
e = api.Engine(_lib.load(), 0)
words = (L + 15) // 16
buf = torch.zeros(n * words + 64, dtype=torch.int32, device="cuda:0")
torch.cuda.synchronize()
e.synth_packed(99, 0, n, L, buf.data_ptr())
dg = api.DeviceGenomes(buf.data_ptr(), n, L)
p = e.params()
first = e.query_sketch(p, dg)
bad = 0
for i in range(reps):
    fr = e.query_sketch(p, dg)
    assert len(fr) == len(first)
    diff = [f for f in range(len(fr)) if len(fr[f]) != len(first[f]) or not np.array_equal(fr[f], first[f])]
    if diff:
        bad += 1
        f = diff[0]
        a, b = set(first[f].tolist()), set(fr[f].tolist())
        print("run", i, "fragments differing:", len(diff), "first:", f, "len", len(first[f]), len(fr[f]), "only first", sorted(a - b)[:4], "only run", sorted(b - a)[:4], flush=True)
print("fragments", len(first), "bad runs", bad, "of", reps, flush=True)
# same for the reference-side minimizer stream
sk0 = api.Sketch(e, p, dg); m0 = sk0.minimizers(); sk0.close()
badm = 0
for i in range(max(4, reps // 4)):
    sk1 = api.Sketch(e, p, dg); m1 = sk1.minimizers(); sk1.close()
    if len(m1) != len(m0) or not np.array_equal(m1, m0):
        badm += 1; print("minimizer stream differs in run", i, len(m0), len(m1), flush=True)
print("minimizers", len(m0), "bad runs", badm, flush=True)
