#!/usr/bin/env python3
"""dev probe: scan kernel time versus read length (how much the read-boundary tiles cost)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import modimizer_amd as mg
from modimizer_amd import synth
L = mg.lib(); dev = torch.device("cuda", 0); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
total = 4_000_000_000
r = torch.empty(L.mgPackedWords(total), dtype=torch.int32, device=dev)
mg.check(L.mgSynthGenome(r.data_ptr(), total, 12345, st))         # random bases; only the offsets differ below
sh = mg.seqhashCreate(21, 64, 17)
cap = int(total / 64 * 1.3)
dk = torch.empty(cap, dtype=torch.int64, device=dev); dcount = torch.zeros(4, dtype=torch.int64, device=dev)
for rl in (150, 1000, 5000, 20000, 100000, 4_000_000):
    n = total // rl
    offs = (np.arange(n + 1, dtype=np.uint64) * np.uint64(rl)); offs[-1] = total
    do = torch.from_numpy(offs.view(np.int64)).to(dev)
    work = torch.empty(L.mgScanWorkBytes(total, n, cap), dtype=torch.uint8, device=dev)
    L.mgProfileEnable(1); L.mgProfileReset()
    for _ in range(3):
        mg.check(L.seqhashScanBatchDevice(sh, r.data_ptr(), total, do.data_ptr(), n, dk.data_ptr(), None, None, cap, dcount.data_ptr(), work.data_ptr(), st))
    torch.cuda.synchronize()
    out = {}
    for i in range(L.mgProfileKernels()):
        nm = C.c_char_p(); ms_ = C.c_double(); cnt = C.c_uint64(); L.mgProfileGet(i, C.byref(nm), C.byref(ms_), C.byref(cnt))
        if cnt.value: out[nm.value.decode()] = ms_.value / cnt.value
    L.mgProfileEnable(0)
    print("read length %8d: scan %.3f ms (%.2f Tbp/s)  tile info %.3f  compact %.3f" % (rl, out["mgScanKernel"], total / out["mgScanKernel"] / 1e9, out["mgTileInfoKernel"], out["mgSegCompactKernel"]))
