"""Parity cases shared by the GPU tests (-m gpu, product library) and the CPU-emulation tests (not gpu, tests/emu build).
Each case drives the engine through the C-ABI and compares with the CPU oracle (tests/orc.py) on the same inputs.
Bars: minimizers, fragment sketches, mapping records (incl. float fields) and CGI rows BIT-EXACT."""
import numpy as np

import golden_cases
import orc
from fastani_amd.api import DeviceGenomes, Sketch


def rng_genome(seed, n, alphabet=b"ACGT"):
    r = np.random.default_rng(seed)
    return np.frombuffer(alphabet, dtype=np.uint8)[r.integers(0, len(alphabet), n)]


def mutate(g, rate, seed):
    r = np.random.default_rng(seed)
    g = g.copy()
    m = r.random(len(g)) < rate
    g[m] = np.frombuffer(b"ACGT", dtype=np.uint8)[r.integers(0, 4, int(m.sum()))]
    return g


def messy_genome(seed, n):
    """multi-contig genome with N runs, IUPAC codes, lower case, a too-short contig and a contig shorter than fragLen"""
    g = orc.synth_genome(seed, 3, n)
    g[1000:1300] = ord("N")
    g[5000] = ord("R")
    g[20000:26000] = np.frombuffer(g[20000:26000].tobytes().lower(), dtype=np.uint8)
    g[26000:26010] = ord("n")
    cut = [0, n // 2, n // 2 + 10, n // 2 + 30, n // 2 + 2000, n]
    return [g[cut[i]:cut[i + 1]] for i in range(len(cut) - 1)]


def check_sketch(engine, genomes, k=16, frag_len=3000):
    p = engine.params(k, frag_len)
    assert p.windowSize == orc.recommended_window(k, frag_len)
    sk = Sketch(engine, p, genomes)
    osk = orc.Sketch(genomes, k, p.windowSize)
    got, exp = sk.minimizers(), osk.minimizers()
    assert len(got) == len(exp), (len(got), len(exp))
    assert np.array_equal(got, exp)
    st = sk.stats()
    assert st["minimizers"] == len(exp) and st["unique"] == osk.unique()
    return p, sk, osk


def check_queries(engine, p, sk, osk, queries, k=16, frag_len=3000):
    all_rows = []
    for qi, q in enumerate(queries):
        fr = engine.query_sketch(p, [q])
        exp_fr = []
        for c in q:
            c = orc.upper(c)
            if len(c) < max(p.windowSize, k, frag_len):
                continue
            for i in range(len(c) // frag_len):
                exp_fr.append(orc.fragment_sketch(c[i * frag_len:(i + 1) * frag_len], k, p.windowSize))
        assert len(fr) == len(exp_fr)
        for a, b in zip(fr, exp_fr):
            assert np.array_equal(a, b)
        maps, tot = sk.map_query(q)
        omaps, otot = osk.map_genome(q, frag_len)
        assert tot == otot
        assert len(maps) == len(omaps), (len(maps), len(omaps))
        assert np.array_equal(maps, omaps)
        cg = sk.compute_cgi(maps, tot, qi)
        ocg = osk.compute_cgi(omaps, otot, qi, frag_len)
        assert np.array_equal(cg, ocg)
        all_rows.append(ocg)
    rows = sk.map_cgi_batch(queries, 0)
    exp = np.concatenate(all_rows) if all_rows else np.zeros(0, dtype=rows.dtype)
    assert np.array_equal(rows, exp)
    return rows


def case_goldens(engine):
    """the HIP path against the committed golden fixtures themselves — outputs of the untouched reference (tests/golden/*.npz, written
    by tests/golden/make_golden.py from oracle/_ref/ref_dump), no oracle in between: minimizers, fragment sketches, the 44-byte mapping
    records and the CGI rows of every golden case, bit for bit"""
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for name, (refs, qrys, k, L) in sorted(golden_cases.cases().items()):
        g = np.load(os.path.join(gold, "case_%s.npz" % name))
        p = engine.params(k, L)
        assert p.windowSize == int(g["w"]), name
        sk = Sketch(engine, p, refs)
        assert np.array_equal(sk.minimizers(), g["minimizers"].view(orc.MINIMIZER_DT).reshape(-1)), name
        rows = []
        for qi, q in enumerate(qrys):
            fr = engine.query_sketch(p, [q])
            assert [len(x) for x in fr] == list(g["fragS%d" % qi]), (name, qi)
            assert np.array_equal(np.concatenate(fr) if len(fr) else np.zeros(0, np.uint32), g["fragH%d" % qi]), (name, qi)
            maps, tot = sk.map_query(q)
            assert np.array_equal(maps, g["maps%d" % qi].view(orc.MAPPING_DT).reshape(-1)), (name, qi)
            rows.append(sk.compute_cgi(maps, tot, qi))
        want = g["cgi"].view(orc.CGI_DT).reshape(-1)
        assert np.array_equal(np.concatenate(rows) if rows else want[:0], want), name
        assert np.array_equal(sk.map_cgi_batch(qrys, 0), want), name
        sk.close()


def case_synthetic_cluster(engine, n=120000):
    genomes = [[orc.synth_genome(7, g, n)] for g in (0, 1, 4, 11, 16, 19, 20)]
    p, sk, osk = check_sketch(engine, genomes)
    rows = check_queries(engine, p, sk, osk, [genomes[0], genomes[3], genomes[6]])
    assert len(rows) >= 8


def case_messy(engine):
    genomes = [messy_genome(5, 90000), [orc.synth_genome(5, 0, 60000)], [rng_genome(1, 10, b"ACGT")], messy_genome(5, 50000)]
    p, sk, osk = check_sketch(engine, genomes)
    check_queries(engine, p, sk, osk, [genomes[0], genomes[1], [b"NNNNNNNN" * 500], genomes[2]])


def case_kmer12(engine):
    genomes = [[orc.synth_genome(3, g, 50000)] for g in (0, 5, 12)]
    p, sk, osk = check_sketch(engine, genomes, k=12)
    check_queries(engine, p, sk, osk, [genomes[1]], k=12)


def case_fraglen1000(engine):
    genomes = [[orc.synth_genome(9, g, 40000)] for g in (0, 7)]
    p, sk, osk = check_sketch(engine, genomes, frag_len=1000)
    check_queries(engine, p, sk, osk, [genomes[0], genomes[1]], frag_len=1000)


def case_tandem_repeats(engine):
    """duplicate hashes inside one super-window (set semantics of the sliding map) and rightmost-tie winnowing"""
    unit = rng_genome(4, 700)
    rep = np.tile(unit, 30)                       # 21 kb of a 700-bp tandem repeat
    noise = mutate(rep, 0.03, 8)
    flank = rng_genome(6, 20000)
    ref = np.concatenate([flank[:10000], rep, flank[10000:]])
    qry = np.concatenate([flank[:10000], noise, flank[10000:]])
    genomes = [[ref], [qry]]
    p, sk, osk = check_sketch(engine, genomes)
    check_queries(engine, p, sk, osk, [[qry], [ref]])


def case_low_complexity(engine):
    """the reference's repeat_* fixtures in miniature: A-runs separated by a T (tests/gen_tests_data.py:33-55)"""
    def rep(na, n):
        return np.frombuffer((b"A" * na + b"T") * (n // (na + 1) + 1), dtype=np.uint8)[:n]
    genomes = [[rep(64, 3500)], [rep(128, 3300)]]      # stays inside the L1 fast-path limits (<= 4096 seed hits)
    p, sk, osk = check_sketch(engine, genomes)
    check_queries(engine, p, sk, osk, [[rep(128, 9000)], [rep(64, 6000)]])


def case_low_complexity_big(engine):
    """thousands of seed hits per fragment (> every LDS class): L1 global-memory path, L2 ranges beyond the fast-path limit.
    The reference's own repeat tests (tests/fastani_tests.cpp:302-416) guard such inputs with -s; without it they still map."""
    def rep(na, n):
        return np.frombuffer((b"A" * na + b"T") * (n // (na + 1) + 1), dtype=np.uint8)[:n]
    genomes = [[rep(64, 40000)], [rep(128, 21000)], [orc.synth_genome(2, 0, 9000)]]
    p, sk, osk = check_sketch(engine, genomes)
    engine.reset_counters()
    check_queries(engine, p, sk, osk, [[rep(128, 6000)], [np.full(3000, ord("A"), dtype=np.uint8)], genomes[2]])
    c = engine.counters()
    assert c["l2SlowCandidates"] > 0          # candidate ranges longer than 16384 entries went through the general kernel


def case_gap_counter_overflow(engine):
    """a fragment whose sketch is a single hash (a period-23 tandem repeat) against windows of ordinary sequence: more than 127
    distinct reference hashes fall into one gap of the query sketch, the 7-bit gap counters of k_l2_sim overflow — while the first
    super-window is still filling (k_l2_sim's fill phase) and later — and those candidates must come back from the general
    kernel, their lane neighbours (ordinary fragments of the same query, in the same wave) unharmed"""
    ref = rng_genome(5, 30000)
    rep = np.tile(rng_genome(9, 23), 400)[:9000]
    related = mutate(ref[14000:23000], 0.05, 3)
    genomes = [[np.concatenate([ref[:12000], rep[:300], ref[12000:]])],
               [np.concatenate([rng_genome(6, 7000), rep[:150], rng_genome(7, 9000), rep[:300], rng_genome(8, 5000)])],
               [orc.synth_genome(2, 0, 9000)]]
    p, sk, osk = check_sketch(engine, genomes)
    engine.reset_counters()
    check_queries(engine, p, sk, osk, [[np.concatenate([rep[:3000], related])], [rep[:6000]], [np.concatenate([related[:3000], rep[:3000], related[3000:6000]])]])
    c = engine.counters()
    assert c["l2SlowOverflow"] > 0 and c["l2FastCandidates"] > 0, c
    # the same with single occurrences of the repeat's hash in the references (40 bases of the repeat here and there: no same-hash
    # neighbours, so the overflow happens in the fill phase; an experimental fill phase that let an overflowing byte carry into the
    # next lane's field gave wrong rows on this input).
    r = np.random.default_rng(105)
    ref = rng_genome(10, 60000)
    rep = np.tile(rng_genome(14, 23), 400)[:9000]
    parts, pos = [], 0
    for i in range(12):
        step = int(r.integers(2500, 4500))
        parts += [ref[pos:pos + step], rep[:40]]
        pos += step
    g0 = np.concatenate(parts + [ref[pos:]])
    related = mutate(ref[2000:32000], 0.06, 8)
    genomes = [[g0], [mutate(g0[:40000], 0.03, 77)], [orc.synth_genome(2, 0, 9000)]]
    p, sk, osk = check_sketch(engine, genomes)
    engine.reset_counters()
    check_queries(engine, p, sk, osk, [[np.concatenate([related[:6000], rep[:3000], related[6000:15000], rep[:3000], related[15000:30000]])]])
    c = engine.counters()
    assert c["l2SlowOverflow"] > 0 and c["l2FastCandidates"] > 0, c


def case_l1_mid_noise(engine):
    """a fragment with 2048 < seed hits <= 4096 most of which are noise: one k-mer of the fragment (chosen with a tiny hash, so that
    it is the minimizer of every window that holds it) sits on 2300 short contigs of one reference genome — isolated hits on 2300
    consecutive contigs, which the noise filter of LDS class M must drop (its tile hash once mapped tile 0 of consecutive contigs
    onto half of its counters and kept 97 % of such hits), leaving the hits on the two relatives"""
    r = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    while True:
        x = acgt[r.integers(0, 4, 16)]
        if orc.hash_kmer(x) < (1 << 32) // 20000:
            break
    base = rng_genome(31, 9000)
    base[1500:1516] = x                                                   # in the first fragment
    contigs = []
    for i in range(2300):
        c = acgt[r.integers(0, 4, 60)]
        c[22:38] = x
        contigs.append(c)
    genomes = [[mutate(base, 0.04, 1)], contigs, [mutate(base, 0.08, 2)], [orc.synth_genome(2, 0, 9000)]]
    for g in (genomes[0], genomes[2]):
        g[0][1500:1516] = x
    p, sk, osk = check_sketch(engine, genomes)
    engine.reset_counters()
    rows = check_queries(engine, p, sk, osk, [[base]])
    c = engine.counters()
    assert len(rows) >= 2 and c["l1MidFragments"] >= 1, (len(rows), c["l1MidFragments"])


def case_sparse_hits(engine):
    """few, scattered seed hits per reference (10-20 % divergence, 60 references): most hits are isolated and the L1 noise
    filter drops them before the sort; the candidates must not change"""
    base = rng_genome(22, 6000)
    genomes = [[np.concatenate([rng_genome(1500 + i, 800), mutate(base, 0.2 if i % 3 else 0.1, 2500 + i), rng_genome(3500 + i, 800)])] for i in range(60)]
    p, sk, osk = check_sketch(engine, genomes)
    check_queries(engine, p, sk, osk, [[base], genomes[3]])


def case_l1_class_overflow(engine):
    """N near-identical copies of one 6-kb region: N x ~240 seed hits per query fragment, nearly all of them with neighbours, so the
    noise filter keeps them — 7 copies: class S, 14: class M (<= 4096 hits), 25 and 40: the batched global-memory path"""
    base = rng_genome(77, 6000)
    for copies in (7, 14, 25, 40):
        genomes = [[np.concatenate([rng_genome(900 + i, 300 + 7 * i), mutate(base, 0.002 * (i % 5), 4100 + i), rng_genome(1900 + i, 500)])] for i in range(copies)]
        p, sk, osk = check_sketch(engine, genomes)
        engine.reset_counters()
        check_queries(engine, p, sk, osk, [[base], genomes[copies // 2]])
        assert (engine.counters()["l1BigFragments"] > 0) == (copies >= 25), copies


def case_species_dense(engine, copies=48, n=15000):
    """a species-dense database in miniature: `copies` strains of one 15-kb genome (0-3 % divergence, small indels) as references,
    four of them as queries — EVERY fragment has more seed hits than the largest LDS class holds and takes the batched
    global-memory L1 path (l1.hpp: one gather, one device sort and one candidate kernel per group of fragments)"""
    base = rng_genome(91, n)
    genomes = [[golden_cases.evolve(base, 600 + i, sub=0.001 * (i % 30), indel=0.0005 * (i % 3), inversions=0, duplications=0, translocations=0)] for i in range(copies)]
    p, sk, osk = check_sketch(engine, genomes)
    engine.reset_counters()
    rows = check_queries(engine, p, sk, osk, [genomes[0], genomes[7], genomes[copies - 1], [base]])
    c = engine.counters()
    assert len(rows) == 4 * copies and c["l1BigFragments"] >= 10, (len(rows), c["l1BigFragments"])


def case_evolved(engine, n=60000, members=7):
    """relatives that differ by more than substitutions (golden_cases.evolve): indels shift the fragment frame, inversions map
    fragments to the reverse strand, duplications / translocations give a fragment several placements in one reference and several
    fragments one reference bin (the 2-way competition, computeCoreIdentity.hpp:237-254), contigs in shuffled order, a plasmid-like
    second contig that is a diverged copy of part of the chromosome"""
    fam = golden_cases.evolved_family(3, n, members, seg=(800, max(1000, n // 10)))
    other = golden_cases.evolved_family(8, n // 2, 2, seg=(800, max(1000, n // 20)))
    genomes = fam + other
    p, sk, osk = check_sketch(engine, genomes)
    rows = check_queries(engine, p, sk, osk, [genomes[0], genomes[2], genomes[4], genomes[5], genomes[-1]])
    assert len(rows) >= 3 * (members - 1)


def case_empty_and_short(engine):
    genomes = [[b"ACGT" * 3], [orc.synth_genome(2, 0, 30000)], [b""], [orc.synth_genome(2, 1, 2999)]]
    p, sk, osk = check_sketch(engine, genomes)
    check_queries(engine, p, sk, osk, [[b"ACGTACGT"], genomes[1], genomes[3], [b""]])


def case_device_synth(engine, alloc):
    """ani_synth_packed == oracle generator, and the DEVICE_PACKED2 input path == the host ASCII path"""
    n, L = 4, 33333
    words = (L + 15) // 16
    buf, ptr = alloc(n * words * 4)
    for cs, first in ((7, 3), (50, 68), (20, 18)):                 # cluster sizes other than 20 (bench.py --cluster-size), then the default
        engine.synth_packed(13, first, n, L, ptr, variant=3, cluster_size=cs)
        host = np.asarray(buf.cpu() if hasattr(buf, "cpu") else buf).view(np.uint32)[:n * words].reshape(n, words)
        for i in range(n):
            g = orc.synth_genome(13, first + i, L, variant=3, cluster_size=cs)
            code = np.zeros(256, dtype=np.uint32)
            code[ord("C")], code[ord("G")], code[ord("T")] = 1, 2, 3
            c = np.zeros(words * 16, dtype=np.uint32)
            c[:L] = code[g]
            exp = (c.reshape(words, 16) << (2 * np.arange(16, dtype=np.uint32))).sum(axis=1).astype(np.uint32)
            assert np.array_equal(host[i], exp), (cs, i)
    for i in range(n):
        g = orc.synth_genome(13, 18 + i, L, variant=3)
        code = np.zeros(256, dtype=np.uint32)
        code[ord("C")], code[ord("G")], code[ord("T")] = 1, 2, 3
        c = np.zeros(words * 16, dtype=np.uint32)
        c[:L] = code[g]
        exp = (c.reshape(words, 16) << (2 * np.arange(16, dtype=np.uint32))).sum(axis=1).astype(np.uint32)
        assert np.array_equal(host[i], exp)
    p = engine.params()
    dg = DeviceGenomes(ptr, n, L)
    sk_dev = Sketch(engine, p, dg)
    genomes = [[orc.synth_genome(13, 18 + i, L, variant=3)] for i in range(n)]
    sk_host = Sketch(engine, p, genomes)
    assert np.array_equal(sk_dev.minimizers(), sk_host.minimizers())
    r1 = sk_dev.map_cgi_batch(dg, 0)
    r2 = sk_host.map_cgi_batch(genomes, 0)
    assert np.array_equal(r1, r2) and len(r1) >= n


def case_chunked(engine, extra=True):
    """run on an engine created with a small ANI_MAX_INDEX_MINIMIZERS: the reference set is held as several index chunks cut at
    genome borders (the device-side form of the reference's database split, computeCoreIdentity.hpp:457-487); minimizers,
    the exact unique-hash count, mappings (global refSeqId, callback order), CGI rows and the fused batch path must not change"""
    genomes = [[orc.synth_genome(7, g, 45000)] for g in (0, 1, 4, 11, 16, 19, 20)] + [messy_genome(5, 50000)]
    p, sk, osk = check_sketch(engine, genomes)
    assert len(sk.chunks()) >= 3 and sk.chunks()[0] == 0, sk.chunks()
    rows = check_queries(engine, p, sk, osk, [genomes[0], genomes[7], genomes[6]])
    assert len(rows) >= 8
    # mappings handed to the reducer in arbitrary order (device sort) give the same rows
    maps, tot = sk.map_query(genomes[3])
    perm = np.random.default_rng(3).permutation(len(maps))
    assert np.array_equal(sk.compute_cgi(maps[perm], tot, 5), sk.compute_cgi(maps, tot, 5))
    case_messy(engine)
    if extra:
        case_tandem_repeats(engine)
        case_empty_and_short(engine)


def case_streamed(engine, tmpdir=None, n=45000, light=False):
    """run on an engine created with a small ANI_MAX_INDEX_MINIMIZERS and ANI_MAX_RESIDENT_CHUNKS=1 (or 2): the reference set is
    STREAMED — the sketch keeps its minimizer records and one chunk's index arrays at a time, every mapping call walks the set chunk
    by chunk (build, map all query sub-batches, drop) — the device-side form of running the reference once per database split
    (computeCoreIdentity.hpp:457-487, scripts/splitDatabase.sh).  Minimizers, the exact unique count, mappings, CGI rows through every
    entry point, several kept fragment sets in one call and the sketch file must not change."""
    from fastani_amd.api import FragmentSet
    genomes = [[orc.synth_genome(7, g, n)] for g in (0, 1, 4, 11, 16, 19, 20)] + [messy_genome(5, n + 5000)] + golden_cases.evolved_family(9, n - 5000, 3, seg=(800, 5000))
    engine.reset_counters()
    p, sk, osk = check_sketch(engine, genomes)
    r = sk.residency()
    n_chunks = len(sk.chunks())
    assert r["streaming"] and n_chunks >= 4 and r["max_resident"] < n_chunks and r["resident_now"] <= max(2, r["max_resident"]), (r, n_chunks)
    rows = check_queries(engine, p, sk, osk, [genomes[0], genomes[7], genomes[6], genomes[9]] if not light else [genomes[7], genomes[9]])
    assert len(rows) >= (10 if not light else 3)
    assert engine.counters()["indexChunkBuilds"] > n_chunks            # chunks were rebuilt
    if light:
        return
    # all-vs-all through kept fragment sets: two sets in ONE call build every chunk once
    exp = []
    for qi, g in enumerate(genomes):
        maps, tot = osk.map_genome(g)
        exp.append(osk.compute_cgi(maps, tot, qi))
    exp = np.concatenate(exp)
    fa, fb = engine.fragment_set(p, genomes[:5]), engine.fragment_set(p, genomes[5:])
    b0 = engine.counters()["indexChunkBuilds"]
    got = sk.map_cgi_fragsets([fa, fb], [0, 5])
    assert engine.counters()["indexChunkBuilds"] - b0 <= n_chunks
    assert np.array_equal(got, exp)
    assert np.array_equal(sk.map_cgi_fragset(fb, 5), exp[exp["qryGenomeId"] >= 5])
    fa.close(); fb.close()
    # the fused all-vs-all pass feeding a streamed set
    contig_len = np.array([len(c) for g in genomes for c in g], dtype=np.int32)
    gcs = np.cumsum([0] + [len(g) for g in genomes]).astype(np.int32)
    ptr, n, frags = engine.sketch_records_self(p, genomes, 0)
    sk2 = Sketch(engine, p, records=(ptr, n, contig_len, gcs))           # caller-owned records: the streamed sketch copies them
    engine.device_free(ptr)
    assert sk2.residency()["streaming"]
    assert np.array_equal(sk2.map_cgi_fragset(frags, 0), exp)
    assert np.array_equal(sk2.minimizers(), osk.minimizers())
    frags.close()
    if tmpdir is not None:
        import os
        path = os.path.join(str(tmpdir), "streamed.anisk")
        sk.save(path, ["g%d" % i for i in range(len(genomes))])
        sk3 = Sketch(engine, p, file=path)
        assert np.array_equal(sk3.minimizers(), osk.minimizers()) and sk3.stats() == sk.stats()
        assert np.array_equal(sk3.map_cgi_batch(genomes[:3], 0), exp[exp["qryGenomeId"] < 3])


def case_fragset_wire(engine, alloc):
    """a kept fragment set packed into one device buffer and viewed from it again (what travels between the GPUs of a
    reference-sharded run) maps to the same rows; damaged buffers are rejected"""
    from fastani_amd.api import AniError, FragmentSet
    genomes = [messy_genome(5, 50000), [orc.synth_genome(5, 0, 40000)], [rng_genome(1, 10, b"ACGT")], [orc.synth_genome(5, 3, 35001)], [b""]]
    p = engine.params()
    sk = Sketch(engine, p, genomes)
    rows0 = sk.map_cgi_batch(genomes, 0)
    f = engine.fragment_set(p, genomes)
    nb = f.packed_bytes()
    buf, ptr = alloc(nb + 1024)
    assert f.pack_into(ptr, nb + 1024) == nb
    info = f.info()
    f.close()
    v = FragmentSet.unpack(engine, ptr, nb, keepalive=buf)
    assert v.info() == info and info["genomes"] == len(genomes)
    assert np.array_equal(sk.map_cgi_fragset(v, 0), rows0)
    v.close()
    for bad_len in (100, nb - 256):
        try:
            FragmentSet.unpack(engine, ptr, bad_len)
            raise AssertionError("a truncated buffer must be rejected")
        except AniError as e:
            assert e.code == -1
    # several packed sets in the slots of one buffer (an all-gather's output) as ONE merged set: slot 1 left out (the rank's own), an
    # empty slot in between, query ids from the slots' bases — the rows of the included genomes, under those ids
    parts = [genomes[:2], genomes[2:3], [[b"ACGT"]], genomes[3:]]
    sets = [engine.fragment_set(p, g) for g in parts]
    pitch = (max(x.packed_bytes() for x in sets) + 255) // 256 * 256
    mbuf, mptr = alloc(pitch * len(sets) + 1024)
    mptr = (mptr + 255) // 256 * 256
    for i, x in enumerate(sets):
        x.pack_into(mptr + i * pitch, pitch)
        x.close()
    bases = [0, -1, 2, 3]                            # genomes 0,1 -> ids 0,1; slot 1 out; the 4-base genome -> id 2 (no fragments); genomes 3,4 -> ids 3,4
    for bad in ([0, -1, 50, 3], [3, -1, 2, 0], [0, -1, 1, 3]):    # query ids must ascend over the used slots, without overlap
        try:
            FragmentSet.unpack_merged(engine, mptr, pitch, bad)
            raise AssertionError("slot bases %r must be rejected" % bad)
        except AniError as e:
            assert e.code == -1
    mv = FragmentSet.unpack_merged(engine, mptr, pitch, bases, keepalive=mbuf)
    assert mv.info()["genomes"] == 5
    got = sk.map_cgi_fragset(mv, 0)
    want = rows0[rows0["qryGenomeId"] != 2]
    assert np.array_equal(got, want), (got, want)
    got100 = sk.map_cgi_fragset(mv, 100)
    assert np.array_equal(got100["qryGenomeId"], want["qryGenomeId"] + 100) and np.array_equal(got100["identity"], want["identity"])
    mv.close()
    try:
        FragmentSet.unpack_merged(engine, mptr + 256, pitch, bases)
        raise AssertionError("slots that do not start with a packed set must be rejected")
    except AniError as e:
        assert e.code == -1
    # an empty set travels too
    f0 = engine.fragment_set(p, [[b"ACGT"]])
    n0 = f0.pack_into(ptr, nb)
    v0 = FragmentSet.unpack(engine, ptr, n0, keepalive=buf)
    assert len(sk.map_cgi_fragset(v0, 7)) == 0
    v0.close(); f0.close()


def case_uploaded(engine):
    """ani_batch_upload: genomes packed and copied to the device once, then used as references AND as queries (the all-vs-all
    command line does that); both host layouts (flat buffer, per-contig pointers); sub-ranges through ani_sketch_records"""
    from fastani_amd.api import UploadedGenomes
    genomes = [messy_genome(5, 50000), [orc.synth_genome(5, 0, 40000)], [rng_genome(1, 10, b"ACGT")], [orc.synth_genome(5, 3, 35001)], [b""]]
    p = engine.params()
    sk0 = Sketch(engine, p, genomes)
    rows0 = sk0.map_cgi_batch(genomes, 0)
    # (third layout: ANI_SEQ_HOST_MIXED_PTRS — the contigs ani_pack_acgt accepts arrive as 2-bit codes made by the caller, the others raw;
    #  lower case packs, anything else does not: the messy genome's contigs with N / IUPAC stay raw bytes)
    for ptrs, mixed in ((False, False), (True, False), (False, True)):
        up = UploadedGenomes(engine, genomes, ptrs=ptrs, mixed=mixed)
        if mixed:
            assert 0 < up.packed_contigs < up.n_contigs
        sk = Sketch(engine, p, up)
        assert np.array_equal(sk.minimizers(), sk0.minimizers())
        assert np.array_equal(sk.map_cgi_batch(up, 0), rows0)
        assert np.array_equal(sk0.map_cgi_batch(up, 0), rows0)
        maps, tot = sk.map_query(genomes[1])
        maps0, tot0 = sk0.map_query(genomes[1])
        assert tot == tot0 and np.array_equal(maps, maps0)
        up.close()
    osk = orc.Sketch(genomes, 16, p.windowSize)
    assert np.array_equal(sk0.minimizers(), osk.minimizers())


def case_self(engine, combos=((16, 3000), (16, 1000), (12, 3000), (16, 3049), (16, 3050), (16, 5000))):
    """all-vs-all in one pass over the hashes (ani_sketch_records_self -> k_sketch_fused): reference records and fragment
    sketches of the same genomes, then ani_map_cgi_fragset; also kept fragment sets of plain queries (ani_fragset_build).
    Fragment lengths that fit a tile together with their lead-in are fused, longer ones take the two separate kernels."""
    genomes = [messy_genome(5, 50000), [orc.synth_genome(5, 0, 40000)], [rng_genome(1, 10, b"ACGT")], [orc.synth_genome(5, 3, 35001)], [b""],
               [orc.synth_genome(5, 1, 2999)], [orc.synth_genome(5, 2, 3000)], [orc.synth_genome(5, 4, 6047), orc.synth_genome(5, 20, 9100)]]
    contig_len = np.array([len(c) for g in genomes for c in g], dtype=np.int32)
    gcs = np.cumsum([0] + [len(g) for g in genomes]).astype(np.int32)
    for k, frag_len in combos:
        p = engine.params(k, frag_len)
        direct = Sketch(engine, p, genomes)
        rows0 = direct.map_cgi_batch(genomes, 0)
        ptr, n, frags = engine.sketch_records_self(p, genomes, 0)
        sk = Sketch(engine, p, records=(ptr, n, contig_len, gcs))
        assert np.array_equal(sk.minimizers(), direct.minimizers()), (k, frag_len)
        rows = sk.map_cgi_fragset(frags, 0)
        assert np.array_equal(rows, rows0), (k, frag_len)
        if n:
            engine.device_free(ptr)
        frags.close()
        f2 = engine.fragment_set(p, genomes[3:])
        assert np.array_equal(direct.map_cgi_fragset(f2, 3), rows0[rows0["qryGenomeId"] >= 3]), (k, frag_len)
        f2.close()
    # and against the oracle for the default parameters
    p = engine.params()
    osk = orc.Sketch(genomes, 16, p.windowSize)
    exp = []
    for qi, g in enumerate(genomes):
        maps, tot = osk.map_genome(g)
        exp.append(osk.compute_cgi(maps, tot, qi))
    ptr, n, frags = engine.sketch_records_self(p, genomes, 0)
    sk = Sketch(engine, p, records=(ptr, n, contig_len, gcs))
    assert np.array_equal(sk.map_cgi_fragset(frags, 0), np.concatenate(exp))


def case_sketch_file(engine, tmpdir):
    """persistent sketch file: save -> load gives the same minimizers, mappings and rows; a genome sub-range loads as a sketch of
    its own with ids renumbered from 0 (what a rank of a multi-GPU job reads); the header is readable without a device"""
    import os
    from fastani_amd.api import AniError, Params
    import ctypes as C
    genomes = [messy_genome(5, 50000), [orc.synth_genome(5, 0, 40000)], [rng_genome(1, 10, b"ACGT")], [orc.synth_genome(5, 3, 35001)], [b""],
               [orc.synth_genome(5, 4, 6047), orc.synth_genome(5, 20, 9100)]]
    names = ["genome_%d.fa" % i for i in range(len(genomes))]
    p = engine.params()
    sk = Sketch(engine, p, genomes)
    path = os.path.join(str(tmpdir), "refs.anisk")
    sk.save(path, names)
    hp, nc, ng, nm = Params(), C.c_int32(), C.c_int32(), C.c_uint64()
    engine._chk(engine.lib.ani_sketch_file_info(path.encode(), C.byref(hp), C.byref(nc), C.byref(ng), C.byref(nm)))
    assert (hp.kmerSize, hp.windowSize, hp.fragLen, ng.value, nm.value) == (16, p.windowSize, 3000, len(genomes), len(sk.minimizers()))
    sk2 = Sketch(engine, p, file=path)
    assert np.array_equal(sk2.minimizers(), sk.minimizers()) and sk2.stats() == sk.stats() and sk2.genome_names() == names
    assert np.array_equal(sk2.map_cgi_batch(genomes, 0), sk.map_cgi_batch(genomes, 0))
    m1, t1 = sk.map_query(genomes[1]); m2, t2 = sk2.map_query(genomes[1])
    assert t1 == t2 and np.array_equal(m1, m2)
    part = Sketch(engine, p, file=path, genome_range=(1, 4))
    ref = Sketch(engine, p, genomes[1:4])
    assert np.array_equal(part.minimizers(), ref.minimizers()) and part.genome_names() == names[1:4]
    assert np.array_equal(part.map_cgi_batch(genomes, 0), ref.map_cgi_batch(genomes, 0))
    # one file from several sketches (ani_sketch_writer_*: a reference set beyond the device memory is written block by block): the
    # same file contents as the one-sketch file, whatever the cut — the second and third sketch's contig and record numbers carry on
    from fastani_amd.api import SketchWriter
    for cuts in ((2,), (1, 4), (3, 5)):
        path2 = os.path.join(str(tmpdir), "refs_parts.anisk")
        w = SketchWriter(engine, path2)
        lo = 0
        for hi in list(cuts) + [len(genomes)]:
            blk = Sketch(engine, p, genomes[lo:hi])
            w.add(blk, names[lo:hi])
            blk.close()
            lo = hi
        w.close()
        sk3 = Sketch(engine, p, file=path2)
        assert np.array_equal(sk3.minimizers(), sk.minimizers()) and sk3.stats() == sk.stats() and sk3.genome_names() == names, cuts
        assert np.array_equal(sk3.map_cgi_batch(genomes, 0), sk.map_cgi_batch(genomes, 0)), cuts
        part3 = Sketch(engine, p, file=path2, genome_range=(1, 4))
        assert np.array_equal(part3.minimizers(), ref.minimizers()), cuts
        for x in (sk3, part3):
            x.close()
    w = SketchWriter(engine, os.path.join(str(tmpdir), "empty.anisk"))
    try:
        w.close()
        raise AssertionError("a sketch file without a sketch must be refused")
    except AniError as e:
        assert e.code == -1 and not os.path.exists(os.path.join(str(tmpdir), "empty.anisk"))
    # a refused add (other parameters) leaves the writer failed: close reports it and no file stays behind; so does a writer given up
    # inside a `with` block or dropped without close (ani_sketch_writer_abort)
    bad = os.path.join(str(tmpdir), "bad.anisk")
    w = SketchWriter(engine, bad)
    w.add(sk, names)
    other = Sketch(engine, engine.params(15, 3000), genomes[:2])
    try:
        w.add(other, names[:2])
        raise AssertionError("sketches with different parameters must be refused")
    except AniError as e:
        assert e.code == -1
    other.close()
    try:
        w.close()
        raise AssertionError("closing a writer after a failed add must fail")
    except AniError:
        assert not os.path.exists(bad)
    try:
        with SketchWriter(engine, bad) as w:
            w.add(sk, names)
            raise RuntimeError("given up")
    except RuntimeError:
        assert not os.path.exists(bad)
    w = SketchWriter(engine, bad)
    w.add(sk, names)
    del w
    assert not os.path.exists(bad)
    open(path, "r+b").write(b"XXXX")
    try:
        Sketch(engine, p, file=path)
        raise AssertionError("a damaged file must be rejected")
    except AniError as e:
        assert e.code == -1


def case_window_sizes(engine, windows=(2, 11, 12, 13, 25, 36, 47, 48, 49, 97)):
    """explicit window sizes around the borders of the winnowing kernel's two forms (prefix/suffix decomposition for
    kPer = 12 <= w <= 48, span doubling otherwise): reference minimizers, fragment sketches and the fused all-vs-all pass"""
    genomes = [messy_genome(6, 30000), [orc.synth_genome(6, 0, 20000)], [orc.synth_genome(6, 2, 9000), rng_genome(3, 7000, b"ACGTN")]]
    contig_len = np.array([len(c) for g in genomes for c in g], dtype=np.int32)
    gcs = np.cumsum([0] + [len(g) for g in genomes]).astype(np.int32)
    for w in windows:
        p = engine.params(16, 3000)
        p.windowSize = w
        sk = Sketch(engine, p, genomes)
        osk = orc.Sketch(genomes, 16, w)
        assert np.array_equal(sk.minimizers(), osk.minimizers()), w
        for q in genomes:
            fr = engine.query_sketch(p, [q])
            exp_fr = [orc.fragment_sketch(orc.upper(c)[i * 3000:(i + 1) * 3000], 16, w) for c in q if len(c) >= 3000 for i in range(len(c) // 3000)]
            assert len(fr) == len(exp_fr), w
            for a, b in zip(fr, exp_fr):
                assert np.array_equal(a, b), w
        ptr, n, frags = engine.sketch_records_self(p, genomes, 0)
        sk2 = Sketch(engine, p, records=(ptr, n, contig_len, gcs))
        assert np.array_equal(sk2.minimizers(), osk.minimizers()), w
        assert np.array_equal(sk2.map_cgi_fragset(frags, 0), sk.map_cgi_batch(genomes, 0)), w
        if n:
            engine.device_free(ptr)
        frags.close()


def case_limits(engine):
    """documented limits fail loudly with ANI_ERR_LIMIT (-4), never silently"""
    from fastani_amd.api import AniError
    p = engine.params(16, 3000)
    p.windowSize = 1                                   # every position is a minimizer: > 4096 per 3-kb... 2985 only; use a long fragment
    p.fragLen = 6000
    g = [[rng_genome(17, 20000)]]
    sk = Sketch(engine, p, g)
    try:
        sk.map_cgi_batch(g, 0)
        raise AssertionError("a fragment with > 4096 minimizers must be rejected")
    except AniError as e:
        assert e.code == -4 and "minimizers" in str(e), str(e)
    p2 = engine.params(16, 3000)
    p2.windowSize = 100000
    try:
        Sketch(engine, p2, g)
        raise AssertionError("window beyond the tile limit must be rejected")
    except AniError as e:
        assert e.code == -4


def case_cand_pool_retry(engine_plain, engine_limited):
    """the L1 candidate pool starts too small (ANI_TEST_CAND_POOL_MIN=1: one candidate per stripe) and the L1 kernels are repeated with the
    size the first attempt asked for — same rows as the oracle; and when the batch ALSO holds a fragment beyond the seed-hit limit
    (ANI_TEST_L1_HIT_LIMIT lowered: k_l1_probe marks it, and k_l1_probe runs on the first attempt only) the call must still fail with
    ANI_ERR_LIMIT after the retry instead of silently dropping the fragment's mappings (the retry used to erase the marker)"""
    from fastani_amd.api import AniError
    base = rng_genome(191, 9000)
    genomes = [[mutate(base, 0.003 * (i % 7), 7100 + i)] for i in range(12)] + [[rng_genome(2900 + i, 7000)] for i in range(3)]
    p, sk, osk = check_sketch(engine_plain, genomes)
    engine_plain.reset_counters()
    check_queries(engine_plain, p, sk, osk, [genomes[0], genomes[5], genomes[13]])
    sk2 = Sketch(engine_limited, p, genomes)
    try:
        sk2.map_cgi_batch([genomes[0], genomes[13]], 0)
        raise AssertionError("a fragment beyond the seed-hit limit must fail the call, also when the candidate pool is retried")
    except AniError as e:
        assert e.code == -4 and "seed hits" in str(e), str(e)
    rows = sk2.map_cgi_batch([genomes[13]], 0)           # the unrelated genome alone stays below the limit: the engine still works
    assert len(rows) == 1 and rows["refGenomeId"][0] == 13
    sk2.close()


def case_ref_id_base(engine):
    """ani_sketch_set_ref_id_base: a sketch that holds a shard / block of a larger set reports refGenomeId + base in the CGI rows of the
    batch entry points (≙ cgi::correctRefGenomeIds, computeCoreIdentity.hpp:480-487, done where the rows are made); mapping records and
    ani_compute_cgi keep the sketch's own ids"""
    genomes = [[orc.synth_genome(21, g, 60000)] for g in (0, 1, 3, 7, 20)]
    p = engine.params()
    sk = Sketch(engine, p, genomes)
    rows0 = sk.map_cgi_batch(genomes, 5)
    maps0, tot0 = sk.map_query(genomes[1])
    sk.set_ref_id_base(1000)
    rows1 = sk.map_cgi_batch(genomes, 5)
    fr = engine.fragment_set(p, genomes)
    rows2 = sk.map_cgi_fragset(fr, 5)
    fr.close()
    maps1, tot1 = sk.map_query(genomes[1])
    exp = rows0.copy()
    exp["refGenomeId"] += 1000
    assert len(rows0) >= 5 and np.array_equal(rows1, exp) and np.array_equal(rows2, exp)
    assert tot0 == tot1 and np.array_equal(maps0, maps1)
    assert np.array_equal(sk.compute_cgi(maps1, tot1, 6), sk.compute_cgi(maps0, tot0, 6))
    sk.set_ref_id_base(0)
    assert np.array_equal(sk.map_cgi_batch(genomes, 5), rows0)
    sk.close()


ALL_CASES = [case_ref_id_base, case_goldens, case_synthetic_cluster, case_messy, case_kmer12, case_fraglen1000, case_tandem_repeats, case_low_complexity,
             case_low_complexity_big, case_gap_counter_overflow, case_l1_mid_noise, case_sparse_hits, case_l1_class_overflow, case_species_dense, case_evolved, case_empty_and_short, case_uploaded, case_self, case_window_sizes]


def fuzz(engine, seed, seconds=None, iterations=None):
    """random genomes (alphabets with N / IUPAC / lower case, tandem repeats, A-rich), relatives by substitution or by
    golden_cases.evolve (indels, inversions, duplications, translocations), random contig splits, k in {8..16},
    fragLen in {500..3000}: sketch, fragment sketches, mappings and CGI rows all bit-exact against the oracle"""
    import time
    rng = np.random.default_rng(seed)

    def rand_genome(n):
        kind = rng.integers(0, 6)
        if kind == 0:
            return rng_genome(int(rng.integers(1e9)), n)
        if kind == 1:
            return orc.synth_genome(int(rng.integers(100)), int(rng.integers(0, 40)), n)
        if kind == 2:
            unit = rng_genome(int(rng.integers(1e9)), int(rng.integers(5, 400)))
            return np.tile(unit, n // len(unit) + 1)[:n].copy()
        if kind == 3:
            return rng_genome(int(rng.integers(1e9)), n, b"ACGTN")
        if kind == 4:
            return rng_genome(int(rng.integers(1e9)), n, b"AAAAAAAT")
        return rng_genome(int(rng.integers(1e9)), n, b"acgtACGTRYKM")

    def contigs(g):
        nc = int(rng.integers(1, 4))
        cuts = [0] + sorted(rng.integers(0, len(g) + 1, nc - 1).tolist()) + [len(g)]
        return [g[cuts[i]:cuts[i + 1]] for i in range(nc)]

    t0, it = time.time(), 0
    while (seconds is not None and time.time() - t0 < seconds) or (iterations is not None and it < iterations):
        it += 1
        k = int(rng.choice([16, 16, 16, 12, 14, 8]))
        L = int(rng.choice([3000, 3000, 1000, 500, 2000]))
        base = rand_genome(int(rng.integers(L, 12 * L)))
        genomes = []
        if rng.random() < 0.2:                       # many diverged copies: hundreds of scattered seed hits per fragment (L1 noise filter)
            base = base[:4 * L]
            for _ in range(int(rng.integers(20, 70))):
                flank = rng_genome(int(rng.integers(1e9)), int(rng.integers(0, 1500)))
                genomes.append([np.concatenate([flank, mutate(base, float(rng.choice([0.05, 0.1, 0.15, 0.2, 0.25])), int(rng.integers(1e9)))])])
        def relative(rate):
            if rng.random() < 0.5:
                return mutate(base, rate, int(rng.integers(1e9)))
            return golden_cases.evolve(base, int(rng.integers(1e9)), sub=rate, indel=float(rng.choice([0.0, 0.001, 0.005, 0.02])), inversions=int(rng.integers(0, 3)),
                                       duplications=int(rng.integers(0, 3)), translocations=int(rng.integers(0, 3)), seg=(max(50, L // 4), 3 * L))

        if rng.random() < 0.12 and len(base) >= L:  # one minimizer of the first fragment on hundreds or thousands of short contigs: isolated seed hits (L1 classes M / big, noise filter)
            w = orc.recommended_window(k, L)
            mins = orc.winnow(orc.upper(base[:L]), k, w)
            x = None
            if len(mins):
                best = mins[int(np.argmin(mins["hash"]))]              # wpos is the window's start (winSketch.hpp:120-155): the k-mer lies in [wpos, wpos + w)
                ub = orc.upper(base[:L])
                for pos in range(int(best["wpos"]), min(int(best["wpos"]) + w, L - k + 1)):
                    if orc.hash_kmer(ub[pos:pos + k]) == int(best["hash"]):
                        x = ub[pos:pos + k]
                        break
            if x is not None:
                acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
                many = []
                for _ in range(int(rng.choice([300, 1200, 2300, 3000]))):
                    c = acgt[rng.integers(0, 4, int(rng.integers(3 * k + w, 6 * k + w)))]
                    o = int(rng.integers(0, len(c) - k + 1))
                    c[o:o + k] = x
                    many.append(c)
                genomes.append(many)
        for _ in range(int(rng.integers(1, 5))):
            if rng.random() < 0.6:
                g = relative(float(rng.choice([0, 0.01, 0.05, 0.1, 0.2])))
            else:
                g = rand_genome(int(rng.integers(10, 10 * L)))
            genomes.append(contigs(g))
        qs = [genomes[int(rng.integers(len(genomes)))] for _ in range(2)] + [contigs(relative(0.03))]
        if rng.random() < 0.15:                      # a query whose fragments alternate between a short-period repeat (sketches of one or two hashes) and ordinary sequence
            unit = rng_genome(int(rng.integers(1e9)), int(rng.integers(5, 40)))
            rep = np.tile(unit, L // len(unit) + 1)[:L]
            rel = relative(0.04)
            qs.append([np.concatenate([rep, rel[:2 * L], rep, rel[2 * L:4 * L]])])
        p, sk, osk = check_sketch(engine, genomes, k=k, frag_len=L)
        check_queries(engine, p, sk, osk, qs, k=k, frag_len=L)
    return it
