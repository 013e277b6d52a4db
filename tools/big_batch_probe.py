#!/usr/bin/env python3
"""dev probe: ONE batch at the edge of the packed partition element -- 40 Gbp at k=21 d=64: 6.2e8 modimizers, ordinals of 30 bits, 34 bits
of the mixed k-mer: 64 bits exactly -- against the same reads inserted in four passes (MODGPU_ADD_CHUNK): the same modimizer count, entries,
value[] and depth[] (first-occurrence order does not depend on where a stream is cut).  usage: big_batch_probe.py [Gbp] [passes]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import modimizer_amd as mg
from modimizer_amd import synth
cx = bench.Ctx(); cx.torch, cx.dist, cx.mg, cx.synth = torch, None, mg, synth
cx.dev = torch.device("cuda", 0); torch.cuda.set_device(0)
cx.L = L = mg.lib(); mg.check(L.mgSetDevice(0))
cx.stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
total = int(float(sys.argv[1]) * 1e9) if len(sys.argv) > 1 else 40_000_000_000
pieces = int(sys.argv[2]) if len(sys.argv) > 2 else 4
G = total // 30
genome = bench.make_genome(cx, G, 4241)
reads, d_offsets, offsets, n_reads = bench.make_reads(cx, total, genome, G, 4242, 0.05, 4243)
del genome
sh = mg.seqhashCreate(21, 64, 17)
res = []
for how, kn in (("one pass", {}), ("%d passes" % pieces, dict(ADD_CHUNK=-(-total // 64 // pieces) + 1000))):
    with mg.knobs(**kn):
        ms = mg.modsetCreate(sh, 32)
        n_hash = C.c_uint64(0)
        mg.check(L.mgAddReadsDevice(ms, reads.data_ptr(), total, d_offsets.data_ptr(), n_reads, C.byref(n_hash), cx.stream))      # (the first call allocates)
        mg.check(L.mgModsetClear(ms, cx.stream))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mg.check(L.mgAddReadsDevice(ms, reads.data_ptr(), total, d_offsets.data_ptr(), n_reads, C.byref(n_hash), cx.stream))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        tot = n_hash.value
        mg.check(L.modsetSyncToHost(ms, 0))
        U = ms.contents.max
        value = np.ctypeslib.as_array(ms.contents.value, (U + 1,))[1:].copy()
        depth = np.ctypeslib.as_array(ms.contents.depth, (U + 1,))[1:].copy()
        print("%-10s: %.1f ms = %.0f Gbp/s, %d modimizers, %d entries" % (how, dt * 1e3, total / dt / 1e9, tot, U), flush=True)
        res.append((tot, U, value, depth))
        L.modsetDestroy(ms); torch.cuda.empty_cache()
a, b = res
assert a[0] == b[0] and a[1] == b[1], (a[:2], b[:2])
assert np.array_equal(a[2], b[2]), "value[] differs"
assert np.array_equal(a[3], b[3]), "depth[] differs"
print("BIG_BATCH_OK: value[] and depth[] of %d entries identical" % a[1])
