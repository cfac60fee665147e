"""The fastANI-compatible command line against the untouched reference binary (oracle/_ref/fastANI_ref): same argv, the output
file, .matrix and .visual files must be identical after line sorting (the reference's own tests compare that way,
tests/fastani_tests.cpp:22-31,:66-71).  CPU: the CLI linked against the tests/emu build; GPU (-m gpu): fastani_amd/fastANI."""
import os
import subprocess

import numpy as np
import pytest

import golden_cases
import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lines(path):
    return sorted(open(path).read().splitlines())


def _run_both(binary, tmp):
    gs = [[orc.synth_genome(7, i, 45000)] for i in (0, 3, 11, 20)] + [golden_cases.messy(5, 40000)] + [[orc.synth_genome(7, 3, 45000)]]   # the last = genome 1 again: rows with equal identity
    paths = []
    for i, g in enumerate(gs):
        p = os.path.join(tmp, "g%d.fa" % i)
        orc.write_fasta(p, g, names=["c%d_%d some comment" % (i, j) for j in range(len(g))])
        paths.append(p)
    subprocess.check_call(["gzip", "-k", paths[1]])
    paths[1] += ".gz"                                            # one gzipped input
    lst = os.path.join(tmp, "l.txt")
    open(lst, "w").write("\n".join(paths) + "\n\n")
    for extra in ([], ["--matrix"], ["-t", "2", "-s", "--matrix"], ["--fragLen", "1000", "-k", "14", "--minFraction", "0.5"]):
        ra = subprocess.run([orc.REF_BIN, "--ql", lst, "--rl", lst, "-o", os.path.join(tmp, "ref.out")] + extra, capture_output=True)
        rb = subprocess.run([binary, "--ql", lst, "--rl", lst, "-o", os.path.join(tmp, "new.out")] + extra, capture_output=True)
        assert ra.returncode == 0 and rb.returncode == 0, rb.stderr.decode()[-2000:]
        assert _lines(os.path.join(tmp, "ref.out")) == _lines(os.path.join(tmp, "new.out")), extra
        assert len(_lines(os.path.join(tmp, "ref.out"))) > 0
        if "-t" not in extra:      # one thread: the reference's row order is deterministic, ties included (std::sort on a fixed sequence)
            assert open(os.path.join(tmp, "ref.out")).read() == open(os.path.join(tmp, "new.out")).read(), extra
        if "--matrix" in extra:
            assert open(os.path.join(tmp, "ref.out.matrix")).read() == open(os.path.join(tmp, "new.out.matrix")).read()
    # one-to-one with --visualize
    ra = subprocess.run([orc.REF_BIN, "-q", paths[2], "-r", paths[0], "--visualize", "-o", os.path.join(tmp, "v_ref.out")], capture_output=True)
    rb = subprocess.run([binary, "-q", paths[2], "-r", paths[0], "--visualize", "-o", os.path.join(tmp, "v_new.out")], capture_output=True)
    assert ra.returncode == 0 and rb.returncode == 0
    for suf in ("", ".visual"):
        assert _lines(os.path.join(tmp, "v_ref.out" + suf)) == _lines(os.path.join(tmp, "v_new.out" + suf)), suf
    # error behaviour (parseCmdArgs.hpp:200-210, :62-66)
    for args, msg in ((["-q", paths[0]], b"Provide reference file (s)"), (["-r", paths[0]], b"Provide query file (s)"),
                      (["-q", paths[0], "-r", "/nonexistent.fa", "-o", "/dev/null"], b"Could not open /nonexistent.fa")):
        ra = subprocess.run([orc.REF_BIN] + args, capture_output=True)
        rb = subprocess.run([binary] + args, capture_output=True)
        assert ra.returncode == rb.returncode == 1 and msg in ra.stderr and msg in rb.stderr


def _run_evolved(binary, tmp, n=45000, extras=(["--matrix"], ["-t", "2", "--visualize"], ["-t", "3", "--visualize", "--matrix", "--fragLen", "1500"])):
    """relatives with indels / inversions / duplications / translocations / shuffled contigs / a plasmid-like copy: all-vs-all with
    --matrix, and many-to-many --visualize on two threads (the .visual rows come from the 2-way selection, the only place where two
    fragments compete for one reference bin)"""
    fam = golden_cases.evolved_family(5, n, 6, seg=(800, max(1000, n // 8))) + golden_cases.evolved_family(6, n, 2, seg=(800, 4000))
    paths = []
    for i, g in enumerate(fam):
        p = os.path.join(tmp, "e%d.fa" % i)
        orc.write_fasta(p, g, names=["e%d_%d" % (i, j) for j in range(len(g))])
        paths.append(p)
    lst = os.path.join(tmp, "e.txt")
    open(lst, "w").write("\n".join(paths) + "\n")
    for extra in extras:
        ra = subprocess.run([orc.REF_BIN, "--ql", lst, "--rl", lst, "-o", os.path.join(tmp, "eref.out")] + extra, capture_output=True)
        rb = subprocess.run([binary, "--ql", lst, "--rl", lst, "-o", os.path.join(tmp, "enew.out")] + extra, capture_output=True)
        assert ra.returncode == 0 and rb.returncode == 0, rb.stderr.decode()[-2000:]
        assert _lines(os.path.join(tmp, "eref.out")) == _lines(os.path.join(tmp, "enew.out")), extra
        assert len(_lines(os.path.join(tmp, "eref.out"))) >= 20
        if "-t" not in extra:
            assert open(os.path.join(tmp, "eref.out")).read() == open(os.path.join(tmp, "enew.out")).read(), extra
        if "--matrix" in extra:
            assert open(os.path.join(tmp, "eref.out.matrix")).read() == open(os.path.join(tmp, "enew.out.matrix")).read()
        if "--visualize" in extra:
            # byte for byte, not just as a set of lines: which of several equal-identity mappings of a reference bin is reported,
            # and in which order (thread 0's queries first), is reproduced
            assert open(os.path.join(tmp, "eref.out.visual")).read() == open(os.path.join(tmp, "enew.out.visual")).read(), extra
            os.unlink(os.path.join(tmp, "enew.out.visual")); os.unlink(os.path.join(tmp, "eref.out.visual"))


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built")
def test_cli_evolved_cpu_build(tmp_path):
    emu = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-s", "-C", emu, "all"])
    _run_evolved(os.path.join(emu, "fastANI_emu"), str(tmp_path), n=30000)


@pytest.mark.gpu
@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not shipped")
def test_cli_evolved_gpu(tmp_path):
    _run_evolved(os.path.join(ROOT, "fastani_amd", "fastANI"), str(tmp_path))
    _run_evolved(os.path.join(ROOT, "fastani_amd", "fastANI"), str(tmp_path), n=400000)


def _run_streaming_variants(binary, tmp, full=True):
    """the streaming paths of the command line: queries != references, several file slices, several index chunks, and two
    device contexts (here: the same device twice) — every variant must print what the reference binary prints"""
    gs = [[orc.synth_genome(7, i, 36000)] for i in (0, 3, 11, 20, 21, 40)] + [golden_cases.messy(5, 40000)]
    paths = []
    for i, g in enumerate(gs):
        p = os.path.join(tmp, "s%d.fa" % i)
        orc.write_fasta(p, g, names=["c%d_%d" % (i, j) for j in range(len(g))])
        paths.append(p)
    rl, ql = os.path.join(tmp, "rl.txt"), os.path.join(tmp, "ql.txt")
    open(rl, "w").write("\n".join(paths[:5]) + "\n")
    open(ql, "w").write("\n".join(paths[3:]) + "\n")
    al = os.path.join(tmp, "al.txt")
    open(al, "w").write("\n".join(paths) + "\n")
    ref = {}
    for name, args in (("qr", ["--ql", ql, "--rl", rl]), ("all", ["--ql", al, "--rl", al, "--matrix"])):
        out = os.path.join(tmp, "ref_%s.out" % name)
        ra = subprocess.run([orc.REF_BIN] + args + ["-t", "2", "-o", out], capture_output=True)
        assert ra.returncode == 0
        ref[name] = (args, out)
    envs = [({}, []), ({"ANI_SLICE_BYTES": "40000"}, []), ({"ANI_SLICE_BYTES": "40000", "ANI_MAX_INDEX_MINIMIZERS": "7000"}, []),
            ({"ANI_SLICE_BYTES": "40000"}, ["--devices", "0,0"]), ({"ANI_SLICE_BYTES": "80000", "ANI_MAX_INDEX_MINIMIZERS": "7000"}, ["--devices", "0,0,0"]),
            # reference set streamed through the device: one index chunk resident at a time (what a set beyond the HBM gets)
            ({"ANI_SLICE_BYTES": "40000", "ANI_MAX_INDEX_MINIMIZERS": "7000", "ANI_MAX_RESIDENT_CHUNKS": "1"}, []),
            ({"ANI_SLICE_BYTES": "40000", "ANI_MAX_INDEX_MINIMIZERS": "5000", "ANI_MAX_RESIDENT_CHUNKS": "1"}, ["--devices", "0,0"]),
            # the query slices in waves (fragment sketches of one wave made, mapped against every shard, freed): one slice per wave
            # over two contexts, two slices per wave against a streamed set
            ({"ANI_SLICE_BYTES": "40000", "ANI_CLI_QUERY_WAVE_BYTES": "1"}, ["--devices", "0,0"]),
            ({"ANI_SLICE_BYTES": "40000", "ANI_MAX_INDEX_MINIMIZERS": "7000", "ANI_MAX_RESIDENT_CHUNKS": "1", "ANI_CLI_QUERY_WAVE_BYTES": "80000"}, []),
            # the index arrays allocated ahead of the build on a side thread (ani_pool_prewarm_index; forced: these inputs are small)
            ({"ANI_CLI_PREWARM": "force", "ANI_SLICE_BYTES": "40000"}, [])]
    if not full:                    # CPU build: the two streamed variants (one of them chunked over two contexts) and the prewarmed one
        envs = [envs[5], envs[6], envs[-1]]
    for env, extra in envs:
        for name, (args, rout) in ref.items():
            out = os.path.join(tmp, "new_%s.out" % name)
            rb = subprocess.run([binary] + args + ["-t", "3", "-o", out] + extra, capture_output=True, env=dict(os.environ, **env))
            assert rb.returncode == 0, rb.stderr.decode()[-2000:]
            assert _lines(rout) == _lines(out), (env, extra, name)
            assert len(_lines(out)) > 0
            if "--matrix" in args:
                assert open(rout + ".matrix").read() == open(out + ".matrix").read(), (env, extra)
            err = rb.stderr.decode()
            assert "Time spent sketching the reference" in err and "Time spent mapping fragments in query" in err and "Time spent post mapping" in err
            if "ANI_CLI_QUERY_WAVE_BYTES" in env and name == "qr":
                assert "query genomes mapped in" in err and "waves" in err, err[-800:]
    # persistent reference sketch: written by one run, used instead of --rl by the next (same rows, file names from the sketch)
    skf = os.path.join(tmp, "refs.anisk")
    args, rout = ref["qr"]
    out = os.path.join(tmp, "sk1.out")
    rb = subprocess.run([binary] + args + ["-o", out, "--saveSketch", skf], capture_output=True, env=dict(os.environ, ANI_MAX_INDEX_MINIMIZERS="7000"))
    assert rb.returncode == 0 and os.path.getsize(skf) > 4096 and _lines(rout) == _lines(out), rb.stderr.decode()[-1500:]
    for extra in ([], ["--devices", "0,0"]):
        out = os.path.join(tmp, "sk2.out")
        rb = subprocess.run([binary, "--ql", ql, "--refSketch", skf, "-t", "2", "-o", out, "--matrix"] + extra, capture_output=True)
        assert rb.returncode == 0, rb.stderr.decode()[-1500:]
        assert _lines(rout) == _lines(out), extra
    # a sketch file whose records do not fit the device at once is taken in BLOCKS of genomes (load a genome range, map every query,
    # drop it): same rows, whatever the block size, on one context and on two, with the queries in one wave or in several
    for env, extra in (({"ANI_CLI_REF_BLOCK_BYTES": "60000"}, []), ({"ANI_CLI_REF_BLOCK_BYTES": "25000", "ANI_SLICE_BYTES": "40000", "ANI_CLI_QUERY_WAVE_BYTES": "80000"}, []),
                       ({"ANI_CLI_REF_BLOCK_BYTES": "30000", "ANI_MAX_INDEX_MINIMIZERS": "5000", "ANI_MAX_RESIDENT_CHUNKS": "1"}, ["--devices", "0,0"])):
        out = os.path.join(tmp, "sk3.out")
        rb = subprocess.run([binary, "--ql", ql, "--refSketch", skf, "-t", "2", "-o", out, "--matrix"] + extra, capture_output=True, env=dict(os.environ, **env))
        assert rb.returncode == 0, rb.stderr.decode()[-1500:]
        assert _lines(rout) == _lines(out), (env, extra)
        assert b"blocks of genomes per device" in rb.stderr and b"reference block 2 of" in rb.stderr, rb.stderr.decode()[-1500:]
        assert open(os.path.join(tmp, "sk2.out.matrix")).read() == open(out + ".matrix").read(), (env, extra)
    rb = subprocess.run([binary, "--ql", ql, "--refSketch", skf, "-k", "14", "-o", out], capture_output=True)
    assert rb.returncode == 1 and b"sketch file was built with" in rb.stderr
    # --saveSketch on several devices is refused when the options are read, not after the whole reference set has been sketched
    rb = subprocess.run([binary] + args + ["-o", out, "--saveSketch", skf + ".2", "--devices", "0,0"], capture_output=True)
    assert rb.returncode == 1 and b"--saveSketch writes the sketch of one device" in rb.stderr and b"Time spent sketching" not in rb.stderr


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built")
def test_cli_streaming_variants_cpu_build(tmp_path):
    emu = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-s", "-C", emu, "all"])
    _run_streaming_variants(os.path.join(emu, "fastANI_emu"), str(tmp_path))


@pytest.mark.gpu
@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not shipped")
def test_cli_streaming_variants_gpu(tmp_path):
    _run_streaming_variants(os.path.join(ROOT, "fastani_amd", "fastANI"), str(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [20000, 90000])
def test_cli_matrix_at_scale_gpu(tmp_path, n):
    """--matrix for 20 000 and for 90 000 genomes (BASELINE configs[4]: the paper's 90k-prokaryote scale): the reference fills a dense
    N x N float matrix (1.6 GB / 32 GB, computeCoreIdentity.hpp:353-448); the streamed writer must stay far below that and print
    the same values"""
    import resource
    rng = np.random.default_rng(5)
    base = orc.synth_genome(3, 0, 6400)
    paths = []
    for i in range(n):
        p = os.path.join(str(tmp_path), "m%05d.fa" % i)
        if i < 6:
            g = base.copy()
            m = rng.random(len(g)) < 0.01 * i
            g[m] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(m.sum()))]
        else:
            g = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 3300)]
        with open(p, "wb") as f:
            f.write(b">m%d\n" % i + g.tobytes() + b"\n")
        paths.append(p)
    lst = os.path.join(str(tmp_path), "l.txt")
    open(lst, "w").write("\n".join(paths) + "\n")
    out = os.path.join(str(tmp_path), "m.out")
    before = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
    rb = subprocess.run([os.path.join(ROOT, "fastani_amd", "fastANI"), "--ql", lst, "--rl", lst, "--matrix", "-t", "16", "-o", out], capture_output=True)
    assert rb.returncode == 0, rb.stderr.decode()[-2000:]
    rss_mb = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 1024.0
    lines = open(out + ".matrix").read().split("\n")
    assert lines[0] == str(n) and len(lines) == n + 2
    for i in (0, 1, 5, 6, n - 1):
        f = lines[1 + i].split("\t")
        assert f[0] == paths[i] and len(f) == 1 + i
    row5 = lines[1 + 5].split("\t")[1:]
    assert all(v != "NA" and 90.0 < float(v) <= 100.0 for v in row5)          # the six related genomes
    assert set(lines[1 + n - 1].split("\t")[1:]) == {"NA"}
    rows = _lines(out)
    assert len(rows) >= 36 + 1000                                             # 6 x 6 related pairs + the self pairs whose 3.3-kb contig lets the window slide at all
    cap = 1200 if n <= 20000 else 4000        # O(files + results): names, contig tables and rows of 90 000 genomes; the dense matrix would be 32 GB
    assert rss_mb < cap or before / 1024.0 >= cap, "max RSS %.0f MB: the dense matrix alone would be %.0f MB" % (rss_mb, n * n * 4 / 2**20)


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built")
def test_cli_matches_reference_cpu_build(tmp_path):
    emu = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-s", "-C", emu, "all"])
    _run_both(os.path.join(emu, "fastANI_emu"), str(tmp_path))


@pytest.mark.gpu
@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not shipped")
def test_cli_matches_reference_gpu(tmp_path):
    binary = os.path.join(ROOT, "fastani_amd", "fastANI")
    assert os.path.exists(binary), "build the CLI with __graft_entry__.build()"
    _run_both(binary, str(tmp_path))
