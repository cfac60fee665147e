"""Long random parity run on the GPU: python tools/gpu_fuzz.py <seed> <iterations>  (library through the C-ABI vs the oracle)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fastani_amd
import parity_cases as pc
e = fastani_amd.engine(0)
n = pc.fuzz(e, seed=int(sys.argv[1]), iterations=int(sys.argv[2]))
print("fuzz ok:", n, "iterations, seed", sys.argv[1], "counters", {k: e.counters()[k] for k in ("queryFragments", "seedHits", "l1Candidates", "l2SlowCandidates")})
