#!/usr/bin/env python3
"""dev probe: do two independent scan+build pipelines on two streams overlap usefully on one MI355X?"""
import ctypes as C, sys, os, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import modimizer_amd as mg
from modimizer_amd import synth
L = mg.lib(); dev = torch.device("cuda", 0)

def make(total, seed):
    G = total // 30
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    starts, offs, strands = synth.ont_read_plan(total, G, seed)
    g = torch.empty(L.mgPackedWords(G), dtype=torch.int32, device=dev)
    mg.check(L.mgSynthGenome(g.data_ptr(), G, 12345, st))
    ds = torch.from_numpy(starts.view(np.int64)).to(dev); do = torch.from_numpy(offs.view(np.int64)).to(dev); dst = torch.from_numpy(strands).to(dev)
    r = torch.empty(L.mgPackedWords(total), dtype=torch.int32, device=dev)
    mg.check(L.mgSynthReads(g.data_ptr(), G, ds.data_ptr(), do.data_ptr(), dst.data_ptr(), len(starts), total, 0.05, seed, r.data_ptr(), st))
    torch.cuda.synchronize()
    sh = mg.seqhashCreate(21, 64, 17); ms = mg.modsetCreate(sh, 30)
    return dict(reads=r, off=do, n=len(starts), total=total, ms=ms, keep=(ds, dst))

def step(w, stream):
    nh = C.c_uint64()
    mg.check(L.mgModsetClear(w["ms"], stream))
    mg.check(L.mgAddReadsDevice(w["ms"], w["reads"].data_ptr(), w["total"], w["off"].data_ptr(), w["n"], C.byref(nh), stream))

def run(ws, streams, iters=4):
    def loop(w, s):
        for _ in range(iters): step(w, s)
        mg.check(L.mgStreamSynchronize(s))
    for w, s in zip(ws, streams): step(w, s)
    torch.cuda.synchronize()
    t = time.perf_counter()
    th = [threading.Thread(target=loop, args=(w, s)) for w, s in zip(ws, streams)]
    for x in th: x.start()
    for x in th: x.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    return sum(w["total"] for w in ws) * iters / dt / 1e9

one = make(10_000_000_000, 1000)
s0 = torch.cuda.Stream(); s1 = torch.cuda.Stream()
print("single 10 Gbp   : %.1f Gbp/s" % run([one], [C.c_void_p(s0.cuda_stream)]))
del one; torch.cuda.empty_cache()
a = make(5_000_000_000, 1000); b = make(5_000_000_000, 2000)
print("single 5 Gbp    : %.1f Gbp/s" % run([a], [C.c_void_p(s0.cuda_stream)]))
print("two x 5 Gbp, two streams/threads: %.1f Gbp/s aggregate" % run([a, b], [C.c_void_p(s0.cuda_stream), C.c_void_p(s1.cuda_stream)]))
