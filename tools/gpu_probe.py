"""Quick stage timing on the GPU box (development aid): N x N synthetic genomes generated on the device."""
import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastani_amd
from fastani_amd.api import DeviceGenomes, Sketch

def run(n, L=5_000_000, reps=2):
    e = fastani_amd.engine(0)
    p = e.params()
    words = (L + 15) // 16
    buf = torch.zeros(n * words + 64, dtype=torch.int32, device="cuda:0")
    t = time.time(); e.synth_packed(1, 0, n, L, buf.data_ptr()); torch.cuda.synchronize(); t_s = time.time() - t
    dg = DeviceGenomes(buf.data_ptr(), n, L)
    for r in range(reps):
        e.reset_counters()
        t0 = time.time(); sk = Sketch(e, p, dg); t1 = time.time()
        rows = sk.map_cgi_batch(dg, 0); t2 = time.time()
        c = e.counters()
        print(json.dumps(dict(n=n, rep=r, synth_s=round(t_s, 3), build_s=round(t1 - t0, 3), map_s=round(t2 - t1, 3),
                              pairs_per_s=round(n * n / (t2 - t0), 1), rows=len(rows),
                              **{k: (round(v, 2) if isinstance(v, float) else v) for k, v in c.items()})), flush=True)
        sk.close()
    e.close()

if __name__ == "__main__":
    for n in [int(x) for x in sys.argv[1:]] or [20]:
        run(n)
