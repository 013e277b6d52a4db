#!/usr/bin/env python3
"""dev probe: does the instruction-bound scan co-run with the random-access-bound table lookups (config 3's two halves)?
Stream A loops the scan (k-mer + pos + read) of a 5 Gbp batch, stream B loops modsetFindBatchDevice of a pre-scanned batch's
k-mers in a 3 Gbp-reference-sized modset; each alone, then both at once from two host threads."""
import ctypes as C, sys, os, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import modimizer_amd as mg
from modimizer_amd import synth
L = mg.lib(); dev = torch.device("cuda", 0)
total = 5_000_000_000; G = 1_500_000_000
st0 = C.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.empty(L.mgPackedWords(G), dtype=torch.int32, device=dev); mg.check(L.mgSynthGenome(g.data_ptr(), G, 333, st0))
sh = mg.seqhashCreate(21, 64, 17); ms = mg.modsetCreate(sh, 28)
ref_off = torch.arange(0, 13, dtype=torch.int64, device=dev) * (G // 12)
nh = C.c_uint64()
mg.check(L.mgAddReadsDevice(ms, g.data_ptr(), G, ref_off.data_ptr(), 12, C.byref(nh), st0))
starts, offs, strands = synth.ont_read_plan(total, G, 1000)
ds = torch.from_numpy(starts.view(np.int64)).to(dev); do = torch.from_numpy(offs.view(np.int64)).to(dev); dst = torch.from_numpy(strands).to(dev)
r = torch.empty(L.mgPackedWords(total), dtype=torch.int32, device=dev)
mg.check(L.mgSynthReads(g.data_ptr(), G, ds.data_ptr(), do.data_ptr(), dst.data_ptr(), len(starts), total, 0.05, 777, r.data_ptr(), st0))
cap = int(total / 64 * 1.3) + 65536
def bufs():
    return (torch.empty(cap, dtype=torch.int64, device=dev), torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(cap, dtype=torch.int32, device=dev),
            torch.zeros(4, dtype=torch.int64, device=dev), torch.empty(L.mgScanWorkBytes(total, len(starts), cap), dtype=torch.uint8, device=dev))
km, ps, rd, cnt, work = bufs(); km2, ps2, rd2, cnt2, work2 = bufs()
mg.check(L.seqhashScanBatchDevice(sh, r.data_ptr(), total, do.data_ptr(), len(starts), km.data_ptr(), ps.data_ptr(), rd.data_ptr(), cap, cnt.data_ptr(), work.data_ptr(), st0))
torch.cuda.synchronize()
n = int(cnt[0].item()); print("modimizers", n, "reference entries", ms.contents.max)
ix = torch.empty(cap, dtype=torch.int32, device=dev)
sA = torch.cuda.Stream(); sB = torch.cuda.Stream()
A = C.c_void_p(sA.cuda_stream); B = C.c_void_p(sB.cuda_stream)
def scan_loop(k):
    for _ in range(k):
        mg.check(L.seqhashScanBatchDevice(sh, r.data_ptr(), total, do.data_ptr(), len(starts), km2.data_ptr(), ps2.data_ptr(), rd2.data_ptr(), cap, cnt2.data_ptr(), work2.data_ptr(), A))
    mg.check(L.mgStreamSynchronize(A))
def find_loop(k):
    for _ in range(k):
        mg.check(L.modsetFindBatchDevice(ms, km.data_ptr(), n, ix.data_ptr(), B))
    mg.check(L.mgStreamSynchronize(B))
def timed(fs):
    th = [threading.Thread(target=f, args=(6,)) for f in fs]
    torch.cuda.synchronize(); t = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / 6 * 1e3
scan_loop(1); find_loop(1)
a = timed([scan_loop]); b = timed([find_loop]); c = timed([scan_loop, find_loop])
print("scan alone %.2f ms, find alone %.2f ms, both at once %.2f ms per pair (sum %.2f, max %.2f)" % (a, b, c, a + b, max(a, b)))
