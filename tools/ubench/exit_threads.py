"""What a caller in wait() sees of a process's end, by the number of threads parked at exit_group() — no GPU involved.
tools/ubench/exit_threads <GB of touched anonymous memory> <parked threads> prints its threads' ids and a time stamp and calls _exit(0);
this script measures stamp -> wait() returned, and follows MemFree afterwards (the address space is torn down behind the caller when
enough threads were parked).   usage: g++ -O2 -o tools/ubench/exit_threads tools/ubench/exit_threads.cpp -lpthread; python tools/ubench/exit_threads.py"""
import os
import subprocess
import time

exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "exit_threads")


def free_gb():
    for ln in open("/proc/meminfo"):
        if ln.startswith("MemFree"):
            return int(ln.split()[1]) / 1048576


print(os.uname().release, os.cpu_count(), "cpus")
for gb, th in ((2, 0), (2, 1), (2, 2), (2, 4), (2, 6), (2, 8), (2, 16), (2, 64)):
    ts, tr = [], []
    for rep in range(6):
        p = subprocess.Popen([exe, str(gb), str(th)], stdout=subprocess.PIPE)
        p.stdout.readline()
        stamp = float(p.stdout.readline().decode().strip())
        p.wait()
        ts.append(time.time() - stamp)
        if rep == 0:
            for _ in range(6):
                tr.append("%.2f s: %.1f GB" % (time.time() - stamp, free_gb()))
                time.sleep(0.04)
        time.sleep(0.25)
    print("%d GB touched, %2d parked threads: wait() returns after %s s; MemFree after it: %s" % (gb, th, " ".join("%.3f" % t for t in ts), ", ".join(tr)), flush=True)
