"""dev probe: a batch of identical k-mers through the direct (atomic) table path: time per call"""
import ctypes as C, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import modimizer_amd as mg
L = mg.lib(); dev = torch.device("cuda", 0); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rl = 10000; nr = 10000
offs = np.arange(nr + 1, dtype=np.int64) * rl
rng = np.random.default_rng(1)
for name, bases in (("poly-A 0.1 Gbp", np.zeros(nr * rl, np.uint8)), ("random 0.1 Gbp", rng.integers(0, 4, nr * rl).astype(np.uint8))):
    total = len(bases)
    hb = torch.from_numpy(bases).to(dev); packed = torch.empty(L.mgPackedWords(total), dtype=torch.int32, device=dev)
    mg.check(L.mgPackDevice(hb.data_ptr(), total, packed.data_ptr(), st)); do = torch.from_numpy(offs).to(dev)
    sh = mg.seqhashCreate(21, 64, 17); ms = mg.modsetCreate(sh, 28); nh = C.c_uint64(0)
    for rep in range(2):
        mg.check(L.mgModsetClear(ms, st)); torch.cuda.synchronize(); t0 = time.time()
        mg.check(L.mgAddReadsDevice(ms, packed.data_ptr(), total, do.data_ptr(), nr, C.byref(nh), st)); torch.cuda.synchronize(); dt = time.time() - t0
    print("%s [%s]: %d modimizers, %d entries, %.2f ms" % (name, os.environ.get("MODGPU_TABLE_PATH", "auto"), nh.value, ms.contents.max, dt * 1e3))
