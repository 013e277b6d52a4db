"""dev probe: the scan alone on poly-A reads (every start a modimizer): kernel time and the segment re-scan it triggers"""
import ctypes as C, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import modimizer_amd as mg
from modimizer_amd import synth
L = mg.lib(); dev = torch.device("cuda", 0); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(3)
G = 50_000_000
def scan(name, genome, err=0.05):
    total = 1_000_000_000
    starts, offs, strands = synth.ont_read_plan(total, len(genome), 21, n50=20000, lo=500, hi=200000)
    bases = synth.reads_from_genome(genome, starts, offs, strands, err, 22)
    total = len(bases); n = len(starts)
    hb = torch.from_numpy(bases).to(dev); packed = torch.empty(L.mgPackedWords(total), dtype=torch.int32, device=dev)
    mg.check(L.mgPackDevice(hb.data_ptr(), total, packed.data_ptr(), st)); do = torch.from_numpy(offs.astype(np.int64)).to(dev)
    sh = mg.seqhashCreate(21, 64, 17)
    cap = int(total / 64 * 3) + 65536
    dk = torch.empty(cap, dtype=torch.int64, device=dev); dcount = torch.zeros(4, dtype=torch.int64, device=dev)
    work = torch.empty(L.mgScanWorkBytes(total, n, cap), dtype=torch.uint8, device=dev)
    L.mgProfileEnable(1); L.mgProfileReset()
    for _ in range(3):
        mg.check(L.seqhashScanBatchDevice(sh, packed.data_ptr(), total, do.data_ptr(), n, dk.data_ptr(), None, None, cap, dcount.data_ptr(), work.data_ptr(), st))
    torch.cuda.synchronize()
    for i in range(L.mgProfileKernels()):
        nm = C.c_char_p(); ms_ = C.c_double(); cnt = C.c_uint64(); L.mgProfileGet(i, C.byref(nm), C.byref(ms_), C.byref(cnt))
        if cnt.value and nm.value == b"mgScanKernel": print("%-44s scan %.3f ms per Gbp, %d modimizers" % (name, ms_.value / cnt.value / (total / 1e9), int(dcount[0].item())))
    L.mgProfileEnable(0)
iid = rng.integers(0, 4, G).astype(np.uint8)
scan("iid", iid)
for tail, every in ((30, 3000), (30, 30000), (100, 10000), (25, 3000)):
    g = iid.copy()
    for p in range(1000, G - 200, every): g[p:p + tail] = 0
    scan("poly-A of %d every %d (%.1f %%)" % (tail, every, 100.0 * tail / every), g)
    if tail == 30 and every == 3000: scan("  the same, reads without errors", g, 0.0)
