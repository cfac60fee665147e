"""Deterministic inputs of the golden fixtures (regenerated from seeds; only the reference's OUTPUTS are committed)."""
import numpy as np

import orc


def _rng(seed, n, alphabet=b"ACGT"):
    r = np.random.default_rng(seed)
    return np.frombuffer(alphabet, dtype=np.uint8)[r.integers(0, len(alphabet), n)]


def messy(seed, n):
    g = orc.synth_genome(seed, 3, n)
    g[1000:1300] = ord("N")
    g[5000] = ord("R")
    g[20000:26000] = np.frombuffer(g[20000:26000].tobytes().lower(), dtype=np.uint8)
    g[26000:26010] = ord("n")
    cut = [0, n // 2, n // 2 + 10, n // 2 + 30, n // 2 + 2000, n]
    return [g[cut[i]:cut[i + 1]] for i in range(len(cut) - 1)]


def repeat_at(na, n):
    """tests/gen_tests_data.py:33-55 of the reference in miniature: runs of `na` A's separated by one T"""
    return np.frombuffer((b"A" * na + b"T") * (n // (na + 1) + 1), dtype=np.uint8)[:n]


def cases():
    c = {}
    g = [orc.synth_genome(7, i, 60000) for i in (0, 3, 11, 19, 20)]
    c["cluster"] = ([[x] for x in g], [[g[0]], [g[2]], [g[4]]], 16, 3000)
    m = messy(5, 80000)
    c["messy"] = ([m, [orc.synth_genome(5, 0, 50000)], [_rng(1, 10)]], [m, [orc.synth_genome(5, 0, 50000)], [b"N" * 4000]], 16, 3000)
    g12 = [orc.synth_genome(3, i, 40000) for i in (0, 5, 12)]
    c["k12"] = ([[x] for x in g12], [[g12[1]]], 12, 3000)
    g1k = [orc.synth_genome(9, i, 30000) for i in (0, 7)]
    c["frag1000"] = ([[x] for x in g1k], [[g1k[0]], [g1k[1]]], 16, 1000)
    c["repeats"] = ([[repeat_at(64, 3500)], [repeat_at(8, 3400)], [repeat_at(128, 3300)]], [[repeat_at(128, 9000)], [repeat_at(64, 6000)], [np.full(6000, ord("A"), dtype=np.uint8)]], 16, 3000)
    unit = _rng(4, 700)
    rep = np.tile(unit, 30)
    r2 = np.random.default_rng(8)
    noise = rep.copy()
    mm = r2.random(len(noise)) < 0.03
    noise[mm] = np.frombuffer(b"ACGT", dtype=np.uint8)[r2.integers(0, 4, int(mm.sum()))]
    flank = _rng(6, 20000)
    ref = np.concatenate([flank[:10000], rep, flank[10000:]])
    qry = np.concatenate([flank[:10000], noise, flank[10000:]])
    c["tandem"] = ([[ref], [qry]], [[qry], [ref]], 16, 3000)
    fam = evolved_family(3, 60000, 6, seg=(800, 6000))
    c["evolved"] = (fam, [fam[0], fam[2], fam[4], fam[5]], 16, 3000)
    c["onehash"] = one_hash_sketch_case()
    c["manycontigs"] = many_short_contigs_case()
    return c


def _mutate(g, rate, seed):
    r = np.random.default_rng(seed)
    g = g.copy()
    m = r.random(len(g)) < rate
    g[m] = np.frombuffer(b"ACGT", dtype=np.uint8)[r.integers(0, 4, int(m.sum()))]
    return g


def one_hash_sketch_case():
    """query fragments whose sketch is a single hash (a period-23 tandem repeat) beside ordinary ones, against references of ordinary
    sequence that hold 40 bases of the repeat here and there: more than 127 distinct reference hashes fall into one gap of such a
    query sketch (round 3: the gap counters of the GPU's L2 simulation overflow on it, in the fill of the first super-window)"""
    r = np.random.default_rng(105)
    ref = _rng(10, 60000)
    rep = np.tile(_rng(14, 23), 400)[:9000]
    parts, pos = [], 0
    for i in range(12):
        step = int(r.integers(2500, 4500))
        parts += [ref[pos:pos + step], rep[:40]]
        pos += step
    g0 = np.concatenate(parts + [ref[pos:]])
    related = _mutate(ref[2000:32000], 0.06, 8)
    refs = [[g0], [_mutate(g0[:40000], 0.03, 77)], [orc.synth_genome(2, 0, 9000)]]
    qrys = [[np.concatenate([related[:6000], rep[:3000], related[6000:15000], rep[:3000], related[15000:30000]])], [rep[:6000]]]
    return (refs, qrys, 16, 3000)


def many_short_contigs_case():
    """one k-mer of the query's first fragment (chosen with a tiny hash: the minimizer of every window that holds it) on 2300 contigs of
    60 bases of one reference genome, two relatives beside it: thousands of isolated seed hits on consecutive contig ids"""
    r = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    while True:
        x = acgt[r.integers(0, 4, 16)]
        if orc.hash_kmer(x) < (1 << 32) // 20000:
            break
    base = _rng(31, 9000)
    base[1500:1516] = x
    contigs = []
    for i in range(2300):
        c = acgt[r.integers(0, 4, 60)]
        c[22:38] = x
        contigs.append(c)
    refs = [[_mutate(base, 0.04, 1)], contigs, [_mutate(base, 0.08, 2)], [orc.synth_genome(2, 0, 9000)]]
    for g in (refs[0], refs[2]):
        g[0][1500:1516] = x
    return (refs, [[base]], 16, 3000)


# ---------------------------------------------------------------------------------------------
# Evolved genomes: what real relatives differ by besides substitutions.  With substitutions only, fragment i of a query always
# lands at i * fragLen in its relative, candidate ranges are maximally regular and the reducer's bins never have two claimants
# (computeCoreIdentity.hpp:237-254); indels shift the frame, inversions give reverse-strand matches, duplications / translocations
# give several placements per fragment and several fragments per reference bin.
# ---------------------------------------------------------------------------------------------
_COMP = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTacgt", b"TGCAtgca"):
    _COMP[_a] = _b


def revcomp(a):
    return _COMP[np.asarray(a, dtype=np.uint8)[::-1]]


def evolve(g, seed, sub=0.03, indel=0.002, inversions=2, duplications=1, translocations=1, seg=(1000, 20000)):
    """g after `inversions` inversions (reverse complement in place), `duplications` segmental duplications (a copy of a segment
    inserted elsewhere), `translocations` (a segment cut out and re-inserted elsewhere) — segment lengths uniform in `seg`, capped
    at a quarter of the genome — then i.i.d. substitutions at rate `sub` and indel events at rate `indel` per base (half deletions,
    half insertions of random bases; lengths geometric with mean ~3, one event in twenty 50 times longer)."""
    r = np.random.default_rng(seed)
    g = np.asarray(g, dtype=np.uint8).copy()
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)

    def segment(n):
        hi = max(2, min(seg[1], n // 4))
        lo = max(1, min(seg[0], hi - 1))
        ln = int(r.integers(lo, hi + 1))
        a = int(r.integers(0, max(1, n - ln)))
        return a, min(n, a + ln)

    for _ in range(inversions):
        a, b = segment(len(g))
        g[a:b] = revcomp(g[a:b])
    for _ in range(duplications):
        a, b = segment(len(g))
        at = int(r.integers(0, len(g) + 1))
        g = np.concatenate([g[:at], g[a:b], g[at:]])
    for _ in range(translocations):
        a, b = segment(len(g))
        piece, rest = g[a:b].copy(), np.concatenate([g[:a], g[b:]])
        at = int(r.integers(0, len(rest) + 1))
        g = np.concatenate([rest[:at], piece, rest[at:]])
    if sub > 0:
        m = r.random(len(g)) < sub
        g[m] = acgt[r.integers(0, 4, int(m.sum()))]
    n_ev = int(r.binomial(len(g), indel)) if indel > 0 else 0
    if n_ev:
        pos = np.sort(r.choice(len(g), size=n_ev, replace=False))
        lens = r.geometric(0.3, n_ev) * np.where(r.random(n_ev) < 0.05, 50, 1)
        dele = r.random(n_ev) < 0.5
        out, last = [], 0
        for p, ln, d in zip(pos.tolist(), lens.tolist(), dele.tolist()):
            if p < last:
                continue                      # swallowed by the previous deletion
            out.append(g[last:p])
            if d:
                last = min(len(g), p + ln)
            else:
                out.append(acgt[r.integers(0, 4, ln)])
                last = p
        out.append(g[last:])
        g = np.concatenate(out)
    return g


def evolved_family(seed, n, members, contig_games=True, seg=(1000, 20000)):
    """`members` genomes derived from one random ancestor of n bases: member 0 is the ancestor; the others differ by growing amounts
    of substitutions, indels and rearrangements; with `contig_games` some come as several contigs in shuffled order, and some carry
    a second, plasmid-like contig that is a diverged copy of part of the chromosome."""
    r = np.random.default_rng(seed)
    anc = _rng(seed + 1, n)
    fam = [[anc]]
    for m in range(1, members):
        g = evolve(anc, 1000 * seed + m, sub=float(r.choice([0.0, 0.005, 0.02, 0.05, 0.08, 0.12, 0.16])), indel=float(r.choice([0.0005, 0.002, 0.005, 0.01])),
                   inversions=int(r.integers(0, 4)), duplications=int(r.integers(0, 3)), translocations=int(r.integers(0, 3)), seg=seg)
        contigs = [g]
        if contig_games and m % 3 == 1:
            cuts = [0] + sorted(r.integers(0, len(g), int(r.integers(1, 3))).tolist()) + [len(g)]
            contigs = [g[cuts[i]:cuts[i + 1]] for i in range(len(cuts) - 1)]
            contigs = [contigs[i] for i in r.permutation(len(contigs))]
        if contig_games and m % 3 == 2:
            a = int(r.integers(0, max(1, len(anc) - n // 8)))
            contigs.append(evolve(anc[a:a + n // 8], 7000 * seed + m, sub=0.04, indel=0.003, inversions=1, duplications=0, translocations=0, seg=seg))
        fam.append(contigs)
    return fam
