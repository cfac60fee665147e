"""What the FIRST call of the library costs beyond the later ones (code objects loaded on first launch, first pool segment, page-locked staging, LUT uploads): 22 - 31 ms in the
first sketch + index call, 2.3 ms in the first mapping call (tiny input; profiles/r08s_first_call_probe.txt)."""
import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fastani_amd, orc
t0 = time.time(); e = fastani_amd.engine(0); print("engine (ani_init) %.1f ms" % ((time.time() - t0) * 1e3))
p = e.params(16, 3000)
genomes = [[orc.synth_genome(11, g, 90000)] for g in (0, 2, 9, 20)]
for rep in range(3):
    t0 = time.time(); sk = fastani_amd.Sketch(e, p, genomes); t1 = time.time(); rows = sk.map_cgi_batch(genomes, 0); t2 = time.time(); sk.close()
    print("rep %d: sketch+index %.1f ms, map %.1f ms, rows %d" % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, len(rows)))
