#!/usr/bin/env python3
"""dev probe: minimizer batch (seqhash.c:83-152 semantics) throughput on device-resident ONT-like reads"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import modimizer_amd as mg
from modimizer_amd import synth
L = mg.lib(); dev = torch.device("cuda", 0); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
total = int(float(sys.argv[1]) * 1e9) if len(sys.argv) > 1 else 2_000_000_000
G = total // 30
starts, offs, strands = synth.ont_read_plan(total, G, 3)
g = torch.empty(L.mgPackedWords(G), dtype=torch.int32, device=dev); mg.check(L.mgSynthGenome(g.data_ptr(), G, 12345, st))
ds = torch.from_numpy(starts.view(np.int64)).to(dev); do = torch.from_numpy(offs.view(np.int64)).to(dev); dst = torch.from_numpy(strands).to(dev)
r = torch.empty(L.mgPackedWords(total), dtype=torch.int32, device=dev)
mg.check(L.mgSynthReads(g.data_ptr(), G, ds.data_ptr(), do.data_ptr(), dst.data_ptr(), len(starts), total, 0.05, 9, r.data_ptr(), st))
torch.cuda.synchronize()
for k, w in ((21, 31), (21, 64), (15, 10)):
    sh = mg.seqhashCreate(k, w, 17)
    cap = int(total / (w / 2 + 1) * 1.3) + len(starts)
    dh = torch.empty(cap, dtype=torch.int64, device=dev); dq = torch.empty(cap, dtype=torch.int32, device=dev)
    dstart = torch.empty(len(starts) + 1, dtype=torch.int64, device=dev); n = C.c_uint64()
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mg.check(L.seqhashMinimizerBatchDevice(sh, r.data_ptr(), total, do.data_ptr(), len(starts), dh.data_ptr(), dq.data_ptr(), dstart.data_ptr(), cap, C.byref(n), st))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("k=%d w=%d: %d reads, %d minimizers (1 per %.1f bases), %.1f ms, %.1f Gbp/s" % (k, w, len(starts), n.value, total / n.value, dt * 1e3, total / dt / 1e9))
