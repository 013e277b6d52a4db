import os, sys, time, threading, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pyoracle as po
from modimizer_amd import synth
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cpu.max", e)
n = 2000; L = 20000
bases = synth.iid_bases(n * L, 5); off = (np.arange(n + 1, dtype=np.int64) * L)
lib = po.lib()
for T in (1, 2, 4, 8, 16, 32):
    hs = [po.Hasher(21, 64, 17) for _ in range(T)]; ms = [po.Modset(hs[t], 22) for t in range(T)]
    b = [n * t // T for t in range(T + 1)]
    def work(t):
        lo, hi = b[t], b[t + 1]
        sub = np.ascontiguousarray(off[lo:hi + 1] - off[lo]); base = bases[int(off[lo]):int(off[hi])]
        lib.orcScanMany(C.byref(hs[t].c), base.ctypes.data, sub.ctypes.data, hi - lo, ms[t].p)
    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t0
    print("T=%d  %.3f Gbp/s" % (T, n * L / dt / 1e9))
