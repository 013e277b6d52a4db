#!/usr/bin/env python3
"""dev probe: does the VALU-bound scan co-run with a pure memory stream at all?  Stream A loops the scan of 5 Gbp, stream B
loops a device-to-device copy (torch) of COPY_GB; each alone, then both at once.  If this pair overlaps (time ~ max) and
scan || build does not (tools/corun_probe.py), the build kernels contend for more than memory."""
import ctypes as C, sys, os, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import modimizer_amd as mg
from modimizer_amd import synth
L = mg.lib(); dev = torch.device("cuda", 0)
total = 5_000_000_000; G = total // 30
st0 = C.c_void_p(torch.cuda.current_stream().cuda_stream)
starts, offs, strands = synth.ont_read_plan(total, G, 1000)
g = torch.empty(L.mgPackedWords(G), dtype=torch.int32, device=dev); mg.check(L.mgSynthGenome(g.data_ptr(), G, 12345, st0))
ds = torch.from_numpy(starts.view(np.int64)).to(dev); do = torch.from_numpy(offs.view(np.int64)).to(dev); dst = torch.from_numpy(strands).to(dev)
r = torch.empty(L.mgPackedWords(total), dtype=torch.int32, device=dev)
mg.check(L.mgSynthReads(g.data_ptr(), G, ds.data_ptr(), do.data_ptr(), dst.data_ptr(), len(starts), total, 0.05, 777, r.data_ptr(), st0))
sh = mg.seqhashCreate(21, 64, 17)
cap = int(total / 64 * 1.3) + 65536
km = torch.empty(cap, dtype=torch.int64, device=dev); cnt = torch.zeros(4, dtype=torch.int64, device=dev)
work = torch.empty(L.mgScanWorkBytes(total, len(starts), cap), dtype=torch.uint8, device=dev)
gb = float(os.environ.get("COPY_GB", "4"))
src = torch.empty(int(gb * 1e9) // 8, dtype=torch.int64, device=dev).zero_(); dstb = torch.empty_like(src)
sA = torch.cuda.Stream(); sB = torch.cuda.Stream()
A = C.c_void_p(sA.cuda_stream)
def scan_loop(k):
    for _ in range(k):
        mg.check(L.seqhashScanBatchDevice(sh, r.data_ptr(), total, do.data_ptr(), len(starts), km.data_ptr(), None, None, cap, cnt.data_ptr(), work.data_ptr(), A))
    mg.check(L.mgStreamSynchronize(A))
def copy_loop(k):
    with torch.cuda.stream(sB):
        for _ in range(k):
            dstb.copy_(src)
    sB.synchronize()
def timed(fs, k=6):
    th = [threading.Thread(target=f, args=(k,)) for f in fs]
    torch.cuda.synchronize(); t = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / k * 1e3
scan_loop(1); copy_loop(1)
a = timed([scan_loop]); b = timed([copy_loop]); c = timed([scan_loop, copy_loop]); d = timed([scan_loop, scan_loop]); e = timed([copy_loop, copy_loop])
print("scan alone %.2f ms, copy of %.1f GB alone %.2f ms, both at once %.2f ms per pair (sum %.2f, max %.2f); two scans at once %.2f per pair; two copies at once %.2f per pair" % (a, gb, b, c, a + b, max(a, b), d, e))
