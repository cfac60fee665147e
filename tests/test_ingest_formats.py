"""Ingest parity: the command line's block reader restates the vendored kseq.h the reference reads every
input with (src/common/kseq.h:176-214, called from src/map/include/winSketch.hpp:135-170 and computeMap.hpp:132-190, and a third
time for the genome lengths, src/cgi/include/computeCoreIdentity.hpp:48-92).  Here the same genomes are written as FASTA and FASTQ,
LF and CRLF, wrapped and unwrapped, plain and gzipped, with blank lines, junk before the first header, truncated quality strings,
truncated / corrupt gzip members and empty files — and the output of the untouched reference binary is the expectation, byte for
byte (one thread: the reference's row order is deterministic).  CPU: the CLI linked against the tests/emu build; -m gpu:
fastani_amd/fastANI."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import golden_cases
import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _wrap(b, width, eol):
    return b"".join(b[o:o + width] + eol for o in range(0, len(b), width)) if width else b + eol


def _qual(rng, n):
    # quality strings full of the characters that open records ('@', '>', '+'): kseq reads qualities by LENGTH, not by content
    return bytes(rng.choice(np.frombuffer(b"@>+IIIIFFFF#5!~", dtype=np.uint8), n))


def _fasta(contigs, names, width=80, eol=b"\n", lead=b"", blank_every=0, final_eol=True):
    out = [lead]
    for i, (c, nm) in enumerate(zip(contigs, names)):
        out.append(b">" + nm + eol)
        body = _wrap(orc.as_bytes(c).tobytes(), width, eol)
        if blank_every:
            lines = body.split(eol)[:-1]
            body = b"".join(ln + eol + (eol if (j + 1) % blank_every == 0 else b"") for j, ln in enumerate(lines))
        out.append(body)
    s = b"".join(out)
    return s if final_eol else s[:-len(eol)]


def _fastq(rng, contigs, names, width=0, eol=b"\n", repeat_name=False, cut_last_quality=0):
    out = []
    for i, (c, nm) in enumerate(zip(contigs, names)):
        b = orc.as_bytes(c).tobytes()
        q = _qual(rng, len(b))
        if cut_last_quality and i == len(contigs) - 1:
            q = q[:-cut_last_quality]
        out.append(b"@" + nm + eol + _wrap(b, width, eol) + b"+" + (nm if repeat_name else b"") + eol + _wrap(q, width, eol))
    return b"".join(out)


def make_inputs(tmp, n=24000):
    """returns (paths, note per path).  Relatives of one synthetic cluster (every pair has rows) in every container format."""
    rng = np.random.default_rng(11)
    G = lambda m, ln=n: orc.synth_genome(7, m, ln)                       # noqa: E731  cluster 0 of seed 7: member m
    two = lambda m: [G(m)[:n // 2 + 777], G(m)[n // 2 + 777:]]           # noqa: E731  the same genome as two contigs
    files = []

    def put(name, data, note):
        p = os.path.join(tmp, name)
        with open(p, "wb") as f:
            f.write(data)
        files.append((p, note))
        return p
    put("a_plain.fa", _fasta([G(0)], [b"a0 plain"]), "FASTA, 80 columns, LF")
    put("b_crlf.fa", _fasta(two(1), [b"b0 crlf comment", b"b1"], width=70, eol=b"\r\n"), "FASTA, CRLF, header comments")
    put("c_4line.fq", _fastq(rng, two(2), [b"c0", b"c1 x"]), "FASTQ, four lines per record")
    put("d_multi_crlf.fastq", _fastq(rng, two(3), [b"d0", b"d1"], width=60, eol=b"\r\n", repeat_name=True), "FASTQ, wrapped sequence and quality, CRLF, '+name' lines")
    with gzip.open(os.path.join(tmp, "e_gz.fq.gz"), "wb") as f:
        f.write(_fastq(rng, [G(4)], [b"e0"], width=100))
    files.append((os.path.join(tmp, "e_gz.fq.gz"), ".fq.gz"))
    put("f_messy.fa", _fasta(two(5), [b"f0\tTAB comment", b"f1"], width=61, lead=b"junk before the first header\n\n# more junk\n", blank_every=7, final_eol=False),
        "FASTA with leading junk, blank lines, no final newline")
    lower = [np.frombuffer(orc.as_bytes(c).tobytes().lower(), dtype=np.uint8) for c in two(6)]
    put("g_lower_crlf_blank.fa", _fasta(lower, [b"g0", b"g1"], width=50, eol=b"\r\n", blank_every=5), "lower case, CRLF with blank CRLF lines (a lone '\\r' becomes a sequence byte only as the first byte)")
    put("h_corrupt.fa.gz", b"\x1f\x8b\x08\x00" + bytes(rng.integers(0, 256, 4000, dtype=np.uint8)), "gzip magic + garbage: contributes nothing")
    whole = gzip.compress(_fasta(two(7), [b"h0", b"h1"]), 6)
    put("i_truncated.fa.gz", whole[:len(whole) * 2 // 3], "truncated gzip member")
    put("j_cutqual.fq", _fastq(rng, [G(8)[:n // 2], G(8)[n // 2:]], [b"j0", b"j1"], width=80, cut_last_quality=17), "FASTQ whose last quality string is short: kseq_read returns -2, the record is dropped")
    put("k_mixed.fa", _fasta([G(9)[:n // 2]], [b"k0"]) + _fastq(rng, [G(9)[n // 2:]], [b"k1"]) + _fasta([G(10)[:9000]], [b"k2"], width=0), "FASTA record, FASTQ record, unwrapped FASTA record in one file")
    put("l_empty.fa", b"", "empty file")
    put("m_header_only.fa", b">m0 nothing follows\n", "a header without sequence")
    put("n_cr_tail.fa", _fasta([G(11)], [b"n0"], width=80, eol=b"\r\n") + b"\r", "CRLF file that ends in a lone '\\r' (kseq keeps it: the strip needs a following byte or line end)")
    put("o_messy_bytes.fa", _fasta(golden_cases.messy(5, 30000), [b"o%d" % i for i in range(len(golden_cases.messy(5, 30000)))], width=77), "N runs, IUPAC, lower case")
    return files


def run_formats(binary, tmp):
    files = make_inputs(tmp)
    paths = [p for p, _ in files]
    lst = os.path.join(tmp, "l.txt")
    open(lst, "w").write("\n".join(paths) + "\n")
    for extra in (["--fragLen", "1000"], ["--fragLen", "1000", "--matrix", "--minFraction", "0.05"], ["--fragLen", "3000", "-k", "14"]):
        ref_out, new_out = os.path.join(tmp, "ref.out"), os.path.join(tmp, "new.out")
        ra = subprocess.run([orc.REF_BIN, "--ql", lst, "--rl", lst, "-o", ref_out] + extra, capture_output=True)
        rb = subprocess.run([binary, "--ql", lst, "--rl", lst, "-o", new_out] + extra, capture_output=True)
        assert ra.returncode == 0 and rb.returncode == 0, rb.stderr.decode()[-2000:]
        a, b = open(ref_out).read(), open(new_out).read()
        assert a == b, (extra, [ln for ln in a.splitlines() if ln not in set(b.splitlines())][:5], [ln for ln in b.splitlines() if ln not in set(a.splitlines())][:5])
        rows = a.splitlines()
        assert len(rows) >= 100, len(rows)
        # every readable container took part, the unreadable / empty ones did not
        seen = set(ln.split("\t")[0] for ln in rows)
        for p, note in files:
            unread = any(t in os.path.basename(p) for t in ("h_corrupt", "l_empty", "m_header_only"))
            assert (p in seen) != unread, (p, note)
        if "--matrix" in extra:
            assert open(ref_out + ".matrix").read() == open(new_out + ".matrix").read()
    # one-to-one, FASTQ query against CRLF FASTA reference, with --visualize
    for q, r in ((paths[3], paths[1]), (paths[9], paths[13])):
        ra = subprocess.run([orc.REF_BIN, "-q", q, "-r", r, "--fragLen", "1000", "--visualize", "-o", os.path.join(tmp, "v_ref.out")], capture_output=True)
        rb = subprocess.run([binary, "-q", q, "-r", r, "--fragLen", "1000", "--visualize", "-o", os.path.join(tmp, "v_new.out")], capture_output=True)
        assert ra.returncode == 0 and rb.returncode == 0, rb.stderr.decode()[-2000:]
        for suf in ("", ".visual"):
            assert open(os.path.join(tmp, "v_ref.out" + suf)).read() == open(os.path.join(tmp, "v_new.out" + suf)).read(), (q, r, suf)


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built")
def test_ingest_formats_cpu_build(tmp_path):
    emu = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-s", "-C", emu, "all"])
    run_formats(os.path.join(emu, "fastANI_emu"), str(tmp_path))


@pytest.mark.gpu
@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not shipped")
def test_ingest_formats_gpu(tmp_path):
    run_formats(os.path.join(ROOT, "fastani_amd", "fastANI"), str(tmp_path))
