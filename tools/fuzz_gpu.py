"""Time-boxed fuzz on the GPU against the oracle (tests/parity_cases.py:fuzz): python tools/fuzz_gpu.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import parity_cases as pc
from fastani_amd import _lib, api
e = api.Engine(_lib.load(), 0)
t0 = time.time()
for seed in (101, 202, 303):
    n = pc.fuzz(e, seed=seed, seconds=45)
    print("seed", seed, "iterations", n, "ok", flush=True)
print("elapsed", round(time.time() - t0, 1))
