"""Static view of one kernel's device assembly: basic blocks with their instruction mix (VALU / SALU / LDS / VMEM / multiplies).
usage: python tools/isa_blocks.py <dev.s> <kernel name substring> [min instructions]
(dev.s: hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only engine_map.hip — or whichever unit launches the kernel).  The integer kernels of this library are
instruction-issue-bound, so the instruction count of the hot blocks is the figure of merit that can be read
without a GPU."""
import collections
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(pat), l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.amdhsa_kernel") or lines[i].strip().startswith(".Lfunc_end"))
    blocks, cur, name = [], [], "entry"
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            blocks.append((name, cur)); cur, name = [], m.group(1)
            continue
        t = l.strip()
        if t and not t.startswith((";", ".")):
            cur.append(t)
    blocks.append((name, cur))
    tot = collections.Counter()
    print("%-14s %6s %6s %6s %5s %5s %5s  branch" % ("block", "instr", "valu", "salu", "lds", "vmem", "mul"))
    for name, ins in blocks:
        c = collections.Counter()
        for t in ins:
            op = t.split()[0]
            k = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem"
            c[k] += 1
            if re.match(r"v_(mul_lo|mul_hi|mad_u64|mad_i64|mul_u32)", op):
                c["mul"] += 1
        tot.update(c)
        if len(ins) >= min_n:
            br = [t for t in ins if t.startswith(("s_cbranch", "s_branch"))]
            print("%-14s %6d %6d %6d %5d %5d %5d  %s" % (name, len(ins), c["valu"], c["salu"], c["lds"], c["vmem"], c["mul"], " | ".join(b.split()[-1] for b in br[-2:])))
    print("total: %d instructions in %d blocks: %s" % (sum(len(i) for _, i in blocks), len(blocks), dict(tot)))


if __name__ == "__main__":
    main()
