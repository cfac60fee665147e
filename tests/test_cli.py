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
    gs = [[orc.synth_genome(7, i, 45000)] for i in (0, 3, 11, 20)] + [golden_cases.messy(5, 40000)]
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
