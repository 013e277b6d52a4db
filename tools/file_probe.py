#!/usr/bin/env python3
"""dev probe: FASTA file -> modset end to end (parse threads + pack/H2D + scan + insert), Gbp/s.
usage: file_probe.py [Gbp] [line width]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import modimizer_amd as mg

gbp = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
width = int(sys.argv[2]) if len(sys.argv) > 2 else 0
path = "/dev/shm/probe.fa"
rng = np.random.default_rng(1)
n = int(gbp * 1e9)
letters = np.frombuffer(b"ACGT", np.uint8)
with open(path, "wb") as f:
    done, i = 0, 0
    while done < n:
        L = min(int(rng.lognormal(np.log(20000) - 0.36, 0.6)), n - done) or 1
        s = letters[rng.integers(0, 4, L)]
        f.write(b">r%d\n" % i)
        if width:
            pad = (-L) % width
            t = np.concatenate([s, np.zeros(pad, np.uint8)]).reshape(-1, width)
            t = np.concatenate([t, np.full((len(t), 1), 10, np.uint8)], axis=1).ravel()
            t = t[t != 0]
            f.write(t.tobytes())
        else:
            f.write(s.tobytes() + b"\n")
        done += L; i += 1
size = os.path.getsize(path)
L = mg.lib()
for threads in (16,):
    os.environ["MODGPU_PARSE_THREADS"] = str(threads)
    mg.lib().mgReloadKnobs()
    t0 = time.time()
    r = L.mgSeqOpen(path.encode()); b = mg.MgSeqBatch(); tot = 0
    while L.mgSeqNextBatch(r, 2_000_000_000, C.byref(b)):
        tot += b.total; L.mgSeqBatchFree(C.byref(b))
    L.mgSeqClose(r)
    dt = time.time() - t0
    print("parse only, %2d threads: %.2f s  %.2f GB/s text  %.2f Gbp/s" % (threads, dt, size / dt / 1e9, tot / dt / 1e9))
del os.environ["MODGPU_PARSE_THREADS"]
mg.lib().mgReloadKnobs()
if L.mgDeviceCount() > 0:
    sh = mg.seqhashCreate(21, 64, 17); ms = mg.modsetCreate(sh, 28)
    for host in ("1", "0", "0", "1", "0"):
        os.environ["MODGPU_TEXT_HOST"] = host                      # 1: the host parser; 0: plain FASTA parsed on the device (read per call)
        mg.lib().mgReloadKnobs()
        L.mgModsetClear(ms, None)
        t0 = time.time()
        with mg.CFile("/dev/null", "w") as f:
            assert L.mgAddSequenceFile(ms, path.encode(), f) == 0
        dt = time.time() - t0
        print("file -> modset (%s parser): %.3f s  %.2f Gbp/s  (max %d)" % ("host" if host == "1" else "device", dt, n / dt / 1e9, ms.contents.max))
os.remove(path)
