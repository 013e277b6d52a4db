#!/usr/bin/env python3
"""dev probe: does the VALU-bound scan co-run with the memory-bound modset build?  Stream A loops the scan of 5 Gbp,
stream B loops the build (modsetAddBatchDevice) of a pre-scanned k-mer array into another modset; each alone, then
both at once from two host threads."""
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
km2 = torch.empty(cap, dtype=torch.int64, device=dev); cnt2 = torch.zeros(4, dtype=torch.int64, device=dev)
work2 = torch.empty(L.mgScanWorkBytes(total, len(starts), cap), dtype=torch.uint8, device=dev)
mg.check(L.seqhashScanBatchDevice(sh, r.data_ptr(), total, do.data_ptr(), len(starts), km.data_ptr(), None, None, cap, cnt.data_ptr(), work.data_ptr(), st0))
torch.cuda.synchronize()
n = int(cnt[0].item()); print("modimizers", n)
ms = mg.modsetCreate(sh, 29)
pri = int(os.environ.get("CORUN_BUILD_PRIORITY", "0"))           # -1: the build's stream gets the higher priority
sA = torch.cuda.Stream(priority=0 if pri else 0); sB = torch.cuda.Stream(priority=pri)
A = C.c_void_p(sA.cuda_stream); B = C.c_void_p(sB.cuda_stream)
def scan_loop(k):
    for _ in range(k):
        mg.check(L.seqhashScanBatchDevice(sh, r.data_ptr(), total, do.data_ptr(), len(starts), km2.data_ptr(), None, None, cap, cnt2.data_ptr(), work2.data_ptr(), A))
    mg.check(L.mgStreamSynchronize(A))
def build_loop(k):
    for _ in range(k):
        mg.check(L.mgModsetClear(ms, B))
        mg.check(L.modsetAddBatchDevice(ms, km.data_ptr(), n, None, 1, B))
    mg.check(L.mgStreamSynchronize(B))
def timed(fs):
    th = [threading.Thread(target=f, args=(6,)) for f in fs]
    torch.cuda.synchronize(); t = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / 6 * 1e3
scan_loop(1); build_loop(1)
a = timed([scan_loop]); b = timed([build_loop]); c = timed([scan_loop, build_loop])
print("scan alone %.2f ms, build alone %.2f ms, both at once %.2f ms per pair (sum %.2f, max %.2f)" % (a, b, c, a + b, max(a, b)))
