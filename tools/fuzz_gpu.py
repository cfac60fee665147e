"""Fuzz on the GPU against the oracle (tests/parity_cases.py:fuzz).
    python tools/fuzz_gpu.py                     three seeds, 45 s each
    python tools/fuzz_gpu.py <seed> <iterations> one seed, fixed iteration count"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_cases as pc
from fastani_amd import _lib, api
e = api.Engine(_lib.load(), 0)
t0 = time.time()
if len(sys.argv) > 2:
    n = pc.fuzz(e, seed=int(sys.argv[1]), iterations=int(sys.argv[2]))
    print("seed", sys.argv[1], "iterations", n, "ok", {k: e.counters()[k] for k in ("queryFragments", "seedHits", "l1Candidates", "l2SlowCandidates")})
else:
    for seed in (101, 202, 303):
        n = pc.fuzz(e, seed=seed, seconds=45)
        print("seed", seed, "iterations", n, "ok", flush=True)
print("elapsed", round(time.time() - t0, 1))
