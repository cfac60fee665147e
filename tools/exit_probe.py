"""stamp -> process gone for tools/ubench/exit_probe under several loads (device GB, pinned MB, parked threads)"""
import subprocess, sys, time, os
exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "exit_probe")
for gb, pin, th in ((0, 0, 0), (0, 0, 96), (0, 600, 0), (26, 0, 0), (26, 600, 0), (26, 600, 1), (26, 600, 4), (26, 600, 16), (26, 600, 96)):
    ts = []
    for rep in range(3):
        t0 = time.time()
        p = subprocess.Popen([exe, str(gb), str(pin), str(th)], stdout=subprocess.PIPE)
        stamp = float(p.stdout.readline().decode().strip())
        p.wait()
        ts.append((time.time() - stamp, stamp - t0))
    print("device %2d GB, pinned %3d MB, %2d threads: exit takes %s s (start -> stamp %s s)" % (gb, pin, th, " ".join("%.3f" % a for a, _ in ts), " ".join("%.3f" % b for _, b in ts)), flush=True)
