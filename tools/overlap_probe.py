#!/usr/bin/env python3
"""How much of the L1 stage (memory-paced: seed probe, hit gathers) hides beside the L2 stage (vector-issue-bound) of ANOTHER
sub-batch?  The 1000 x 1000 set is sketched once (fused pass, NS slices -> NS kept fragment sets, one index); then the NS sets are
mapped (a) one after the other on one context — what the library does today — and (b) by two host threads with a context each on
the same device, the second starting --stagger-ms after the first, so that one thread's L1 kernels meet the other's L2 kernels.
Rows must be identical.  Prints one line per variant (min / median of --reps).

    python tools/overlap_probe.py [--genomes 1000] [--slices 8] [--reps 5]
"""
import argparse
import ctypes as C
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genomes", type=int, default=1000)
    ap.add_argument("--genome-len", type=int, default=5_000_000)
    ap.add_argument("--slices", type=int, nargs="+", default=[2, 4, 8])
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--stagger-ms", type=float, nargs="+", default=[0.0, 8.0, 15.0])
    a = ap.parse_args()
    import numpy as np
    import torch
    import fastani_amd
    from fastani_amd.api import CGI_DT, DeviceGenomes, Sketch
    eA, eB = fastani_amd.engine(0), fastani_amd.engine(0)
    p = eA.params(16, 3000)
    N, L = a.genomes, a.genome_len
    words = (L + 15) // 16
    buf = torch.empty(N * words + 64, dtype=torch.int32, device="cuda:0")
    eA.synth_packed(20260925, 0, N, L, buf.data_ptr())
    contig_len = np.full(N, L, dtype=np.int32)
    gcs = np.arange(N + 1, dtype=np.int32)

    def map_one(e, sk, fr, first):
        pp, n = C.c_void_p(), C.c_size_t()
        e._chk(e.lib.ani_map_cgi_fragset(e.h, sk.h, fr.h, first, C.byref(pp), C.byref(n)))
        return e._take(pp, n.value, CGI_DT)

    for ns in a.slices:
        step = (N + ns - 1) // ns
        parts, sets, firsts = [], [], []
        for s0 in range(0, N, step):
            s1 = min(N, s0 + step)
            ptr, n, fr = eA.sketch_records_self(p, DeviceGenomes(buf.data_ptr(), N, L, first=s0, count=s1 - s0), s0)
            parts.append((ptr, n, s0)); sets.append(fr); firsts.append(s0)
        sk = Sketch(eA, p, record_parts=([x[0] or 0 for x in parts], [x[1] for x in parts], [x[2] for x in parts] + [N], contig_len, gcs), adopt=True)
        # warm both contexts (pools, LUTs)
        ref_rows = [map_one(eA, sk, fr, f0) for fr, f0 in zip(sets, firsts)]
        for fr, f0 in zip(sets[:2], firsts[:2]):
            map_one(eB, sk, fr, f0)
        torch.cuda.synchronize()
        ts = []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            rows = [map_one(eA, sk, fr, f0) for fr, f0 in zip(sets, firsts)]
            ts.append((time.perf_counter() - t0) * 1e3)
        print("slices %d  sequential, one context      : min %.1f ms  median %.1f ms" % (len(sets), min(ts), sorted(ts)[len(ts) // 2]), flush=True)
        t1 = []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            rows = sk.map_cgi_fragsets(sets, firsts)
            t1.append((time.perf_counter() - t0) * 1e3)
        print("slices %d  one ani_map_cgi_fragsets call: min %.1f ms  median %.1f ms" % (len(sets), min(t1), sorted(t1)[len(t1) // 2]), flush=True)
        for stagger in a.stagger_ms:
            tp = []
            ok = True
            for _ in range(a.reps):
                out = [None] * len(sets)

                def work(e, idx, delay):
                    if delay:
                        time.sleep(delay / 1e3)
                    for i in idx:
                        out[i] = map_one(e, sk, sets[i], firsts[i])
                thA = threading.Thread(target=work, args=(eA, list(range(0, len(sets), 2)), 0.0))
                thB = threading.Thread(target=work, args=(eB, list(range(1, len(sets), 2)), stagger))
                t0 = time.perf_counter()
                thA.start(); thB.start(); thA.join(); thB.join()
                tp.append((time.perf_counter() - t0) * 1e3)
                ok = ok and all(np.array_equal(x, y) for x, y in zip(out, ref_rows))
            print("slices %d  two contexts, stagger %4.1f ms : min %.1f ms  median %.1f ms  rows identical: %s" % (len(sets), stagger, min(tp), sorted(tp)[len(tp) // 2], ok), flush=True)
        for fr in sets:
            fr.close()
        sk.close()
    eA.close(); eB.close()


if __name__ == "__main__":
    main()
