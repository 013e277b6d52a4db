#!/usr/bin/env python3
"""dev probe: scan kernel time per Gbp for the three phase-A modes: FAST (21,64), POW2 (31,4), ANY (19,31: the reference's default k, w)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import modimizer_amd as mg
L = mg.lib(); dev = torch.device("cuda", 0); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
total = 2_000_000_000
r = torch.empty(L.mgPackedWords(total), dtype=torch.int32, device=dev)
mg.check(L.mgSynthGenome(r.data_ptr(), total, 12345, st))
rl = 20000; n = total // rl
offs = (np.arange(n + 1, dtype=np.uint64) * np.uint64(rl)); offs[-1] = total
do = torch.from_numpy(offs.view(np.int64)).to(dev)
for k, w in ((21, 64), (31, 4), (19, 31), (15, 8), (31, 97)):
    sh = mg.seqhashCreate(k, w, 17)
    cap = int(total / w * 1.3) + 65536
    dk = torch.empty(cap, dtype=torch.int64, device=dev); dp = torch.empty(cap, dtype=torch.int32, device=dev); dr = torch.empty(cap, dtype=torch.int32, device=dev)
    dcount = torch.zeros(4, dtype=torch.int64, device=dev)
    work = torch.empty(L.mgScanWorkBytes(total, n, cap), dtype=torch.uint8, device=dev)
    res = []
    for where in (False, True):
        L.mgProfileEnable(1); L.mgProfileReset()
        for _ in range(3):
            mg.check(L.seqhashScanBatchDevice(sh, r.data_ptr(), total, do.data_ptr(), n, dk.data_ptr(), dp.data_ptr() if where else None, dr.data_ptr() if where else None, cap, dcount.data_ptr(), work.data_ptr(), st))
        torch.cuda.synchronize()
        for i in range(L.mgProfileKernels()):
            nm = C.c_char_p(); ms_ = C.c_double(); cnt = C.c_uint64(); L.mgProfileGet(i, C.byref(nm), C.byref(ms_), C.byref(cnt))
            if cnt.value and nm.value == b"mgScanKernel": res.append(ms_.value / cnt.value / (total / 1e9))
        L.mgProfileEnable(0)
    print("k=%2d d=%3d: scan %.3f ms per Gbp (k-mers only), %.3f (with pos / read); %d modimizers" % (k, w, res[0], res[1], int(dcount[0].item())))
    del dk, dp, dr, work
