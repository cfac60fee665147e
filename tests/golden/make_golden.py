"""Generates the committed golden fixtures from the REFERENCE ITSELF (oracle/_ref/ref_dump, built from the untouched
sources under /root/reference by oracle/Makefile).  Run in the build container only:

    python tests/golden/make_golden.py

Outputs (small, committed):
  tests/golden/case_*.npz      inputs are regenerated from seeds by tests/golden_cases.py; the .npz hold the reference's
                               minimizers, fragment sketches, mapping records and CGI rows for each case
  tests/golden/stats_k16.npz   the reference's own LUTs: minimumHits(s), nucIdentity / upper bound for every (s, shared), s<=400
  tests/golden/windows.json    recommendedWindowSize for several (k, fragLen)
  tests/golden/visual_kat.json identity values and per-pair (count, mean) of the reference's own expected test outputs
                               (tests/data/*.visual, *-test.txt): known-answer vectors for the identity formula and the
                               reducer's mean that need no genomes
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc  # noqa: E402
import golden_cases  # noqa: E402

REF_DATA = "/root/reference/tests/data"


def main():
    assert orc.have_ref(), "build oracle/_ref first (make -C oracle)"
    only = set(sys.argv[1:])                       # optional: regenerate only the named cases
    for name, (refs, qrys, k, L) in golden_cases.cases().items():
        if only and name not in only:
            continue
        with tempfile.TemporaryDirectory() as td:
            d = orc.run_ref_dump(td, refs, qrys, k=k, frag_len=L)
        out = {"k": d["k"], "w": d["w"], "L": d["L"], "minimizers": d["minimizers"], "cgi": d["cgi"],
               "contigLen": d["contigLen"], "seqsByFile": d["seqsByFile"]}
        for qi in range(len(qrys)):
            out["maps%d" % qi] = d["maps"][qi]
            fr = d["frags"][qi]
            out["fragS%d" % qi] = np.array([len(x) for x in fr], dtype=np.int32)
            out["fragH%d" % qi] = np.concatenate(fr) if fr else np.zeros(0, dtype=np.uint32)
        np.savez_compressed(os.path.join(HERE, "case_%s.npz" % name), **out)
        print(name, len(d["minimizers"]), [len(m) for m in d["maps"]], len(d["cgi"]))
    if only:
        return
    # statistics LUTs from the reference's own functions
    with tempfile.TemporaryDirectory() as td:
        subprocess.check_call([orc.REF_DUMP, "--stats", "16", "400", os.path.join(td, "s")])
        raw = np.fromfile(os.path.join(td, "s.stats"), dtype="<i4")
        maxS = int(raw[0])
        minhits = raw[1:1 + maxS].copy()
        fl = raw[1 + maxS:].view("<f4").reshape(-1, 2)
        np.savez_compressed(os.path.join(HERE, "stats_k16.npz"), maxS=maxS, minHits=minhits, identity=fl[:, 0].copy(), upper=fl[:, 1].copy())
    wins = {}
    for k, L in [(16, 3000), (12, 3000), (16, 1000), (16, 5000), (8, 3000), (16, 500)]:
        wins["%d,%d" % (k, L)] = int(subprocess.check_output([orc.REF_DUMP, "--window", str(k), str(L)]).split()[0])
    json.dump(wins, open(os.path.join(HERE, "windows.json"), "w"), indent=1)
    # known-answer vectors from the reference's expected outputs
    kat = {"identities": [], "pairs": []}
    ids = set()
    for fn in sorted(os.listdir(REF_DATA)):
        if fn.endswith(".visual"):
            groups, groups_str = {}, {}
            for line in open(os.path.join(REF_DATA, fn)):
                p = line.rstrip("\n").split("\t")
                ids.add(p[2])
                groups.setdefault((p[0], p[1]), []).append(np.float32(p[2]))
                groups_str.setdefault((p[0], p[1]), []).append(p[2])
            exp = {}
            for line in open(os.path.join(REF_DATA, fn[:-len(".visual")])):
                q, r, ani, cnt, tot = line.split("\t")
                exp[(q, r)] = (ani, int(cnt))
            for key, vals in groups.items():
                if key in exp:
                    kat["pairs"].append({"file": fn, "n": len(vals), "expected_ani": exp[key][0], "expected_count": exp[key][1],
                                         "rows": groups_str[key]})
    kat["identities"] = sorted(ids)
    pos = {v: i for i, v in enumerate(kat["identities"])}
    for pr in kat["pairs"]:
        pr["rows"] = [pos[v] for v in pr["rows"]]
    json.dump(kat, open(os.path.join(HERE, "visual_kat.json"), "w"))
    print("visual KAT:", len(kat["identities"]), "distinct identity strings,", len(kat["pairs"]), "pairs")


if __name__ == "__main__":
    main()
