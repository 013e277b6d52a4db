#!/usr/bin/env python3
"""dev probe: PCIe-inclusive rate of the host-buffer entry point (bytes 0..3 in host memory -> modset)"""
import ctypes as C, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import modimizer_amd as mg
from modimizer_amd import synth
L = mg.lib(); dev = torch.device("cuda", 0); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
total = int(float(os.environ.get("E2E_GBP", "4")) * 1e9); G = total // 30
starts, offs, strands = synth.ont_read_plan(total, G, 1000)
g = torch.empty(L.mgPackedWords(G), dtype=torch.int32, device=dev); mg.check(L.mgSynthGenome(g.data_ptr(), G, 12345, st))
ds = torch.from_numpy(starts.view(np.int64)).to(dev); do = torch.from_numpy(offs.view(np.int64)).to(dev); dst = torch.from_numpy(strands).to(dev)
r = torch.empty(L.mgPackedWords(total), dtype=torch.int32, device=dev)
mg.check(L.mgSynthReads(g.data_ptr(), G, ds.data_ptr(), do.data_ptr(), dst.data_ptr(), len(starts), total, 0.05, 777, r.data_ptr(), st))
b = torch.empty(total, dtype=torch.uint8, device=dev); mg.check(L.mgUnpackDevice(r.data_ptr(), total, b.data_ptr(), st)); torch.cuda.synchronize()
hb = b.cpu().numpy(); del b, r, g
o64 = offs.astype(np.int64)
sh = mg.seqhashCreate(21, 64, 17); ms = mg.modsetCreate(sh, 28)
for it in range(3):
    mg.check(L.mgModsetClear(ms, st))
    t = time.perf_counter()
    n = L.mgAddSequenceBatch(ms, hb.ctypes.data, o64.ctypes.data, len(starts))
    dt = time.perf_counter() - t
    print("mgAddSequenceBatch host bytes: %.3f s  %.2f Gbp/s  hashes %d" % (dt, total / dt / 1e9, n))
w = np.zeros(L.mgPackedWords(total), np.uint32)
t = time.perf_counter(); L.mgPackHost(hb.ctypes.data, total, w.ctypes.data); dt = time.perf_counter() - t
print("mgPackHost alone: %.3f s  %.2f GB/s of bases" % (dt, total / dt / 1e9))
