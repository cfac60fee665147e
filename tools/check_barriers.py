"""Device-assembly check: every s_barrier of the library must be preceded by an `s_waitcnt lgkmcnt(0)` with no LDS write in between
(ROCm 7.2 was seen to drop the wait a release fence needs at a loop header; block_barrier() spells it out).
usage: python tools/check_barriers.py [dev.s ...]   (without argument: compiles every unit of fastani_amd/csrc to assembly first)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    if len(sys.argv) > 1:
        paths = sys.argv[1:]
    else:
        td = tempfile.mkdtemp()
        paths = []
        for u in ("engine_core", "engine_ingest", "engine_sketch", "engine_index", "engine_map", "sort_device"):
            path = os.path.join(td, u + ".s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", path,
                                   os.path.join(ROOT, "fastani_amd", "csrc", u + ".hip")], stderr=subprocess.DEVNULL)
            paths.append(path)
    lines = []
    for path in paths:
        lines += open(path).read().split("\n")
    lds_write = re.compile(r"\s*ds_(write|store|add|sub|or|and|xor|max|min|inc|dec|cmpst|wrxchg|append|consume)")
    nb = bad = 0
    for i, l in enumerate(lines):
        if not re.match(r"\s*s_barrier", l):
            continue
        nb += 1
        j, ok = i - 1, False
        while j >= 0 and not re.match(r"^[A-Za-z_.$][\w.$]*:", lines[j]):     # back to the start of the basic block at most
            t = lines[j].strip()
            if t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
                ok = True
                break
            if lds_write.match(lines[j]):
                break
            j -= 1
        if not ok:
            bad += 1
            print("barrier at line %d: no lgkmcnt(0) wait after the last LDS write" % (i + 1))
    print("%d barriers, %d without their wait" % (nb, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
