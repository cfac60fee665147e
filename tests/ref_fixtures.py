"""The reference's own runnable test inputs, regenerated.

/root/reference/tests/data/repeat_*_2048.fa (2 Mbp each) are the inputs of the only part of the reference's test suite whose data is
in its tree (tests/fastani_tests.cpp:302-416, the 8 "Repeat A2048 and <n>AT" cases).  They are the output of a generator the
reference ships (tests/gen_tests_data.py:33-55: runs of `nas` A's followed by `nts` T's until `nlength` bases are reached, wrapped
at 80 columns, one header line).  The GPU box has no /root/reference, and reference files are not copied into this repository:
this module restates the recipe, and tests/golden/ref_fixture_sha256.json pins its output to the reference's files byte for byte
(checked against the files themselves wherever /root/reference is present, tests/test_ref_fixtures.py).
"""
import hashlib
import json
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fixture_sha256.json")

# file name -> (header line, A's per period, T's per period, minimum length); headers as in the reference's files (typos included)
FIXTURES = {
    "repeat_8ats_2048.fa": (">Repat8AT2048", 8, 1, 2048000),
    "repeat_12ats_2048.fa": (">Repeat12AT2048", 12, 1, 2048000),
    "repeat_16ats_2048.fa": (">Repeat16AT2048", 16, 1, 2048000),
    "repeat_20ats_2048.fa": (">Repeat20AT2048", 20, 1, 2048000),
    "repeat_24ats_2048.fa": (">Repeat24AT2048", 24, 1, 2048000),
    "repeat_32ats_2048.fa": (">Repeat32AT2048", 32, 1, 2048000),
    "repeat_64ats_2048.fa": (">Repeat32AT2048", 64, 1, 2048000),
    "repeat_128ats_2048.fa": (">Repeat32AT2048", 128, 1, 2048000),
    "repeat_as_2048.fa": (">RepatAs2048", 32, 0, 2048000),
}
# the 8 cases of tests/fastani_tests.cpp:302-416: (test name, reference file); the query is always repeat_as_2048.fa
CASES = [("repeat-A2048-8AT", "repeat_8ats_2048.fa"), ("repeat-A2048-12AT", "repeat_12ats_2048.fa"), ("repeat-A2048-16AT", "repeat_16ats_2048.fa"),
         ("repeat-A2048-20AT", "repeat_20ats_2048.fa"), ("repeat-A2048-24AT", "repeat_24ats_2048.fa"), ("repeat-A2048-32AT", "repeat_32ats_2048.fa"),
         ("repeat-A2048-64AT", "repeat_64ats_2048.fa"), ("repeat-A2048-128AT", "repeat_128ats_2048.fa")]
QUERY = "repeat_as_2048.fa"


def fixture_bytes(name):
    header, na, nt, nlength = FIXTURES[name]
    period = b"A" * na + b"T" * nt
    reps = -(-nlength // len(period))              # whole periods until the length is reached (the last one is not cut)
    seq = period * reps
    lines = [seq[i:i + 80] for i in range(0, len(seq), 80)]
    return header.encode() + b"\n" + b"\n".join(lines) + b"\n"


def write_all(directory):
    """writes the nine files into `directory` (as data/<name> of the reference's test working directory); returns {name: path}"""
    os.makedirs(directory, exist_ok=True)
    out = {}
    for name in FIXTURES:
        p = os.path.join(directory, name)
        with open(p, "wb") as f:
            f.write(fixture_bytes(name))
        out[name] = p
    return out


def sha256_table():
    return {name: hashlib.sha256(fixture_bytes(name)).hexdigest() for name in FIXTURES}


def pinned():
    return json.load(open(GOLD))
