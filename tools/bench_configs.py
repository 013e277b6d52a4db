#!/usr/bin/env python3
"""dev helper: the other BASELINE configs at reduced size (not the headline bench).
   C5: 150 b reads, k=31 d=4 (short-read / dense stress)   C3: modmap-style query of reads against a reference modset"""
import ctypes as C, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import modimizer_amd as mg
from modimizer_amd import synth
L = mg.lib(); dev = torch.device("cuda", 0); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

def gen(total, G, plan, err, seed):
    starts, offs, strands = plan
    g = torch.empty(L.mgPackedWords(G), dtype=torch.int32, device=dev)
    mg.check(L.mgSynthGenome(g.data_ptr(), G, 12345, st))
    ds = torch.from_numpy(starts.view(np.int64)).to(dev); do = torch.from_numpy(offs.view(np.int64)).to(dev); dst = torch.from_numpy(strands).to(dev)
    r = torch.empty(L.mgPackedWords(total), dtype=torch.int32, device=dev)
    mg.check(L.mgSynthReads(g.data_ptr(), G, ds.data_ptr(), do.data_ptr(), dst.data_ptr(), len(starts), total, err, seed, r.data_ptr(), st))
    torch.cuda.synchronize()
    return r, do, g

def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n

# ---- C5: 50x of a 20 Mbp genome, 150 b reads, k=31 d=4, B=28
G = 20_000_000; nreads = 6_666_667; total = nreads * 150
plan = synth.fixed_read_plan(nreads, 150, G, 5)
reads, doff, _ = gen(total, G, plan, 0.005, 9)
sh = mg.seqhashCreate(31, 4, 17); ms = mg.modsetCreate(sh, 28); nh = C.c_uint64()
def c5():
    mg.check(L.mgModsetClear(ms, st)); mg.check(L.mgAddReadsDevice(ms, reads.data_ptr(), total, doff.data_ptr(), nreads, C.byref(nh), st))
L.mgProfileEnable(1); L.mgProfileReset()
dt = timeit(c5)
print("C5 1 Gbp 150b k31 d4: %.1f ms  %.1f Gbp/s  hashes %d entries %d" % (dt * 1e3, total / dt / 1e9, nh.value, ms.contents.max))
for i in range(L.mgProfileKernels()):
    nm = C.c_char_p(); ms_ = C.c_double(); n = C.c_uint64(); L.mgProfileGet(i, C.byref(nm), C.byref(ms_), C.byref(n))
    if n.value: print("    %-28s %8.3f ms/launch x%d" % (nm.value.decode(), ms_.value / n.value, n.value))
L.modsetDestroy(ms); del reads

# ---- C3 (scaled): reference 300 Mbp in 24 sequences -> modset; queries 3 Gbp of reads from it
G = 300_000_000
ref_offs = (np.arange(25, dtype=np.uint64) * np.uint64(G // 24)); ref_offs[-1] = G
gw = torch.empty(L.mgPackedWords(G), dtype=torch.int32, device=dev)
mg.check(L.mgSynthGenome(gw.data_ptr(), G, 12345, st))
d_ro = torch.from_numpy(ref_offs.view(np.int64)).to(dev)
sh = mg.seqhashCreate(21, 64, 17); ms = mg.modsetCreate(sh, 26)
mg.check(L.mgAddReadsDevice(ms, gw.data_ptr(), G, d_ro.data_ptr(), 24, C.byref(nh), st))
print("C3 reference modset: %d hashes, %d entries" % (nh.value, ms.contents.max))
total = 3_000_000_000
plan = synth.ont_read_plan(total, G, 3)
reads, doff, _ = gen(total, G, plan, 0.05, 9)
cap = int(total / 64 * 1.2)
six = torch.empty(cap, dtype=torch.int32, device=dev); spos = torch.empty(cap, dtype=torch.int32, device=dev); srd = torch.empty(cap, dtype=torch.int32, device=dev)
ns = C.c_uint64()
def c3():
    mg.check(L.mgQueryReadsDevice(ms, reads.data_ptr(), total, doff.data_ptr(), len(plan[0]), six.data_ptr(), spos.data_ptr(), srd.data_ptr(), cap, C.byref(ns), st))
L.mgProfileReset()
dt = timeit(c3)
hit = int((six[:ns.value] != 0).sum().item())
print("C3 query 3 Gbp vs 300 Mbp ref: %.1f ms  %.1f Gbp/s  seeds %d hit %.3f" % (dt * 1e3, total / dt / 1e9, ns.value, hit / ns.value))
for i in range(L.mgProfileKernels()):
    nm = C.c_char_p(); ms_ = C.c_double(); n = C.c_uint64(); L.mgProfileGet(i, C.byref(nm), C.byref(ms_), C.byref(n))
    if n.value: print("    %-28s %8.3f ms/launch x%d" % (nm.value.decode(), ms_.value / n.value, n.value))
