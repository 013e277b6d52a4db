"""dev: SURVEY §8(f) N3 / N4 timed -- modasm's read ingest (mgReadsetRead: scan + lookup + hit lists + invBuild) from host bytes against a
modset built from a genome, and the minimizer batch scan (seqhashMinimizerBatchDevice), on ONT-like reads.  usage: n34_probe.py [Gbp] [genome Mbp] [self]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import modimizer_amd as mg
import bench
gbp = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
gmbp = float(sys.argv[2]) if len(sys.argv) > 2 else 300
L = mg.lib(); mg.check(L.mgSetDevice(0))
cx = bench.Ctx(); cx.torch, cx.mg, cx.L = torch, mg, L
from modimizer_amd import synth
cx.synth = synth; cx.dev = torch.device("cuda", 0); cx.stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
k, d, bits = 21, 64, 28
G = int(gmbp * 1e6)
genome = bench.make_genome(cx, G, 333)
sh = mg.seqhashCreate(k, d, 17); ms = mg.modsetCreate(sh, bits)
off = torch.tensor([0, G], dtype=torch.int64, device=cx.dev); n = C.c_uint64()
mg.check(L.mgAddReadsDevice(ms, genome.data_ptr(), G, off.data_ptr(), 1, C.byref(n), cx.stream))
total = int(gbp * 1e9)
reads, d_off, offs, n_reads = bench.make_reads(cx, total, genome, G, 4000, 0.05, 5000)
total = int(offs[n_reads])
d_bytes = torch.empty(total, dtype=torch.uint8, device=cx.dev)
mg.check(L.mgUnpackDevice(reads.data_ptr(), total, d_bytes.data_ptr(), cx.stream)); torch.cuda.synchronize()
h = d_bytes.cpu().numpy(); del d_bytes
o64 = offs[:n_reads + 1].astype(np.int64)
if len(sys.argv) > 3 and sys.argv[3] == "self":                    # the modset of the reads themselves: every modimizer is a hit (bench.py's readset_ingest)
    mg.check(L.mgModsetClear(ms, None)); mg.check(L.mgAddReadsDevice(ms, reads.data_ptr(), total, d_off.data_ptr(), n_reads, C.byref(n), cx.stream))
for it in range(5):
    rs = L.mgReadsetCreate(ms)
    os.environ["MODGPU_SEED_TIMING"] = "1"; L.mgReloadKnobs()
    t0 = time.perf_counter(); rc = L.mgReadsetRead(rs, h.ctypes.data, o64.ctypes.data, n_reads); dt = time.perf_counter() - t0
    R = C.cast(rs, C.POINTER(mg.MgReadset)).contents
    print("mgReadsetRead: %.2f Gbp, %d reads, %d hits: %.3f s = %.2f Gbp/s (rc %d)" % (total / 1e9, n_reads, R.totHit, dt, total / dt / 1e9, rc), flush=True)
    L.mgReadsetDestroy(rs)
# N4: minimizers of the same batch, device resident
w = 31
shm = mg.seqhashCreate(19, w, 17)
cap = int(total / (w / 2 + 1) + total / 8 + n_reads + 1024)
dH = torch.empty(cap, dtype=torch.int64, device=cx.dev); dP = torch.empty(cap, dtype=torch.int32, device=cx.dev); dS = torch.empty(n_reads + 2, dtype=torch.int64, device=cx.dev)
nm = C.c_uint64()
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mg.check(L.seqhashMinimizerBatchDevice(shm, reads.data_ptr(), total, d_off.data_ptr(), n_reads, dH.data_ptr(), dP.data_ptr(), dS.data_ptr(), cap, C.byref(nm), cx.stream))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("seqhashMinimizerBatchDevice k=19 w=31: %d minimizers of %.2f Gbp in %.2f ms = %.1f Gbp/s" % (nm.value, total / 1e9, dt * 1e3, total / dt / 1e9), flush=True)
