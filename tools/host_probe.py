import sys, time, torch
sys.path.insert(0, '/root/repo')
import numpy as np, fastani_amd
from fastani_amd.api import DeviceGenomes, Sketch
e = fastani_amd.engine(0); p = e.params(16, 3000)
NG, L = 1000, 5000000; words = (L + 15) // 16
buf = torch.empty(NG * words + 64, dtype=torch.int32, device='cuda')
e.synth_packed(1234, 0, NG, L, buf.data_ptr(), variant=0)
refs = DeviceGenomes(buf.data_ptr(), NG, L)
for it in range(3):
    e.reset_counters(); torch.cuda.synchronize()
    t0 = time.perf_counter(); sk = Sketch(e, p, refs); t1 = time.perf_counter()
    rows = sk.map_cgi_batch(refs, 0); t2 = time.perf_counter()
    sk.close(); t3 = time.perf_counter()
    c = e.counters()
    print("build %.1f (stages %.1f) map %.1f (stages %.1f) close %.1f total %.1f" % ((t1-t0)*1e3, c['msSketch']+c['msIndex'], (t2-t1)*1e3, c['msFragSketch']+c['msL1']+c['msL2']+c['msReduce'], (t3-t2)*1e3, (t3-t0)*1e3))
