"""End-to-end probe of the command line on a GPU box: writes the 1000 x 5 Mbp FASTA set once (80 columns), then runs
`fastANI --ql all --rl all` under several environments and prints, for each, the launcher's wall clock, the command line's own phase
marks and what lies outside main() (process start -> first mark, last mark -> exit).
usage: python tools/e2e_probe.py [n_genomes] [variant ...]     variant = NAME=VALUE[,NAME=VALUE...] or `default`"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import bench
    import orc
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    variants = sys.argv[2:] or ["default"]
    L = 5000000
    td = tempfile.mkdtemp(prefix="ani_e2e_", dir="/tmp")
    t0 = time.time()
    paths = bench.write_fasta_set(orc, 20260925, list(range(n)), L, td, 32)
    print("wrote %d genomes in %.1f s" % (n, time.time() - t0), flush=True)
    lst = os.path.join(td, "all.txt")
    open(lst, "w").write("\n".join(paths) + "\n")
    cli = os.path.join(ROOT, "fastani_amd", "fastANI")
    for v in variants:
        env = dict(os.environ, ANI_CLI_TRACE="1")
        threads = "64"
        if v != "default":
            for kv in v.split(","):
                k, val = kv.split("=")
                if k == "T":
                    threads = val
                else:
                    env[k] = val
        for rep in range(int(os.environ.get("E2E_REPS", "2"))):
            # a run that starts right behind another device process pays for that one's release (the runtime's start takes 0.16 - 0.23 s
            # instead of 0.05, tools/ubench/init_probe.hip): E2E_PAUSE seconds between runs measure a user's single run
            time.sleep(float(os.environ.get("E2E_PAUSE", "0")))
            t0 = time.time()
            pr = subprocess.Popen([cli, "--ql", lst, "--rl", lst, "-t", threads, "-o", os.path.join(td, "out.txt")], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
            marks = []
            exit_note = ""
            keep = open(os.path.join(os.environ["E2E_STDERR_DIR"], "stderr_%s_rep%d.txt" % (v.replace("=", "-").replace(",", "_"), rep)), "w") if os.environ.get("E2E_STDERR_DIR") else None
            for raw in pr.stderr:
                ln = raw.decode(errors="replace").rstrip("\n")
                if keep:
                    keep.write("%.4f %s\n" % (time.time() - t0, ln))
                if ln.startswith("[fastANI trace]"):
                    marks.append((time.time() - t0, float(ln.split()[2]), " ".join(ln.split()[4:])))
                if ln.startswith("[fastANI exit]"):
                    exit_note = ln
            pr.wait()
            wall = time.time() - t0
            print("%-40s rep %d  wall %.3f s  before main %.3f  after last mark %.3f %s | %s" % (
                v, rep, wall, marks[0][0] - marks[0][1] if marks else -1, wall - marks[-1][0] if marks else -1, exit_note,
                "; ".join("%.3f %s" % (m[1], m[2]) for m in marks)), flush=True)
    subprocess.call(["rm", "-rf", td])


if __name__ == "__main__":
    main()
