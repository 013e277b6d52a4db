#!/usr/bin/env python3
"""dev probe: a modset built batch after batch (what a file of many batches does): ten 1 Gbp batches of ONT-like reads from one genome
added to ONE modset (k=21 d=64, table bits 30), the time of every add and its kernels -- against the same 10 Gbp in one add."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import modimizer_amd as mg
from modimizer_amd import synth
cx = bench.Ctx(); cx.torch, cx.dist, cx.mg, cx.synth = torch, None, mg, synth
cx.dev = torch.device("cuda", 0); torch.cuda.set_device(0)
cx.L = L = mg.lib(); mg.check(L.mgSetDevice(0))
cx.stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
G = 333_000_000
genome = bench.make_genome(cx, G, 333)
sh = mg.seqhashCreate(21, 64, 17)
per = int(float(os.environ.get("PROBE_BATCH_GBP", "1")) * 1e9); nb = int(os.environ.get("PROBE_BATCHES", "10"))
batches = [bench.make_reads(cx, per, genome, G, 100 + b, 0.05, 200 + b) for b in range(nb)]
ms = mg.modsetCreate(sh, 30)
n_hash = C.c_uint64(0)
for rep in range(2):
    mg.check(L.mgModsetClear(ms, cx.stream)); torch.cuda.synchronize()
    L.mgProfileOnly(-1); L.mgProfileEnable(1)
    tot = 0
    for b, (reads, d_off, offs, n_reads) in enumerate(batches):
        L.mgProfileReset(); torch.cuda.synchronize(); t0 = time.perf_counter()
        mg.check(L.mgAddReadsDevice(ms, reads.data_ptr(), per, d_off.data_ptr(), n_reads, C.byref(n_hash), cx.stream))
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3; tot += dt
        if True:
            prof = bench.read_profile(L, mg)
            top = {k.replace("Kernel", "").replace("mg", ""): round(v[0], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:6]}
            print("rep %d batch %2d: %6.3f ms, entries %9d  %s" % (rep, b, dt, ms.contents.max, top), flush=True)
    print("%s: all %d batches: %.3f ms" % ("table grown from nothing" if not rep else "after a clear (the slots stay)", nb, tot))
