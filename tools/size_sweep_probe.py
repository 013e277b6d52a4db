#!/usr/bin/env python3
"""dev probe: the step (clear + scan + build, k=21 d=64, table bits 28) by batch size -- 0.05 .. 10 Gbp of ONT-like reads -- with the
per-kernel table, to see what does not scale down (a fixed cost, a launch shape made for the headline's 10 Gbp).
usage: size_sweep_probe.py [sizes in Gbp ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import modimizer_amd as mg
from modimizer_amd import synth
sizes = [float(x) for x in sys.argv[1:]] or [0.05, 0.2, 0.5, 1, 2, 5]
cx = bench.Ctx()
cx.torch, cx.dist, cx.mg, cx.synth = torch, None, mg, synth
cx.dev = torch.device("cuda", 0); torch.cuda.set_device(0)
cx.L = L = mg.lib(); mg.check(L.mgSetDevice(0))
cx.stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
G = 333_000_000
genome = bench.make_genome(cx, G, 333)
sh = mg.seqhashCreate(21, 64, 17)
for gbp in sizes:
    total = int(gbp * 1e9) // 16 * 16
    reads, d_offsets, offsets, n_reads = bench.make_reads(cx, total, genome, G, 21, 0.05, 22)
    ms = mg.modsetCreate(sh, 28)
    n_hash = C.c_uint64(0)
    def step():
        mg.check(L.mgModsetClear(ms, cx.stream))
        mg.check(L.mgAddReadsDevice(ms, reads.data_ptr(), total, d_offsets.data_ptr(), n_reads, C.byref(n_hash), cx.stream))
    dt, kern, table, regions = bench.best_of_two(cx, step, 5)
    per = {kn.replace("Kernel", "").replace("mg", ""): round(v[0] / gbp, 3) for kn, v in sorted(table.items(), key=lambda kv: -kv[1][0])[:9]}
    tot_k = sum(v[0] for v in table.values())
    if os.environ.get("PROBE_RAW"):
        print("   raw:", {k_: v for k_, v in table.items() if "Scan" in k_ or "Lookup" in k_}, {k_: v for k_, v in kern.items() if "Scan" in k_})
    print("%6.2f Gbp: %7.3f ms per step = %6.3f ms per Gbp (kernels %6.3f, the rest %6.3f per step); per Gbp: %s" %
          (gbp, dt / 5 * 1e3, dt / 5 * 1e3 / gbp, tot_k, dt / 5 * 1e3 - tot_k, per), flush=True)
    L.modsetDestroy(ms); del reads, d_offsets
    torch.cuda.empty_cache()

# ---- the query path by batch size: a reference modset of the whole genome, then mgQueryReadsDevice on batches of each size ----
if os.environ.get("PROBE_QUERY"):
    ms = mg.modsetCreate(sh, 28)
    ref_off = torch.tensor([0, G], dtype=torch.int64, device=cx.dev)
    nh = C.c_uint64(0)
    mg.check(L.mgAddReadsDevice(ms, genome.data_ptr(), G, ref_off.data_ptr(), 1, C.byref(nh), cx.stream))
    torch.cuda.synchronize()
    print("query path: reference of %d bases, %d entries" % (G, ms.contents.max))
    for gbp in sizes:
        total = int(gbp * 1e9) // 16 * 16
        reads, d_offsets, offsets, n_reads = bench.make_reads(cx, total, genome, G, 21, 0.05, 22)
        cap = int(total / 64 * 1.3) + (1 << 16)
        q = [torch.empty(cap, dtype=torch.int32, device=cx.dev) for _ in range(3)]
        n_seeds = C.c_uint64(0)
        def step():
            mg.check(L.mgQueryReadsDevice(ms, reads.data_ptr(), total, d_offsets.data_ptr(), n_reads, q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), cap, C.byref(n_seeds), cx.stream))
        dt, kern, table, regions = bench.best_of_two(cx, step, 5)
        per = {kn.replace("Kernel", "").replace("mg", ""): round(v[0] / gbp, 3) for kn, v in sorted(table.items(), key=lambda kv: -kv[1][0])[:8]}
        tot_k = sum(v[0] for v in table.values())
        print("%6.2f Gbp: %7.3f ms per call = %6.3f ms per Gbp (kernels %6.3f, the rest %6.3f per call); per Gbp: %s" %
              (gbp, dt / 5 * 1e3, dt / 5 * 1e3 / gbp, tot_k, dt / 5 * 1e3 - tot_k, per), flush=True)
        del reads, d_offsets, q
        torch.cuda.empty_cache()
    L.modsetDestroy(ms)
