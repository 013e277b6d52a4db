#!/usr/bin/env python3
"""dev probe: batches with k-mers of very many copies (identical reads, poly-A, a two-base repeat) through mgAddReadsDevice: time and the kernels it goes to"""
import ctypes as C, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import modimizer_amd as mg
L = mg.lib(); dev = torch.device("cuda", 0); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(5)
def run(name, bases, offs, bits=28):
    total = len(bases); n = len(offs) - 1
    hb = torch.from_numpy(bases).to(dev)
    packed = torch.empty(L.mgPackedWords(total), dtype=torch.int32, device=dev)
    mg.check(L.mgPackDevice(hb.data_ptr(), total, packed.data_ptr(), st))
    do = torch.from_numpy(offs.astype(np.int64)).to(dev)
    sh = mg.seqhashCreate(21, 64, 17); ms = mg.modsetCreate(sh, bits)
    nh = C.c_uint64(0)
    for rep in range(2):
        mg.check(L.mgModsetClear(ms, st)); torch.cuda.synchronize(); t0 = time.time()
        mg.check(L.mgAddReadsDevice(ms, packed.data_ptr(), total, do.data_ptr(), n, C.byref(nh), st)); torch.cuda.synchronize()
        dt = time.time() - t0
    print("%-40s %.2f Gbp, %d modimizers, %d entries: %.1f ms (%.1f Gbp/s)" % (name, total / 1e9, nh.value, ms.contents.max, dt * 1e3, total / dt / 1e9))
    L.mgProfileEnable(1); L.mgProfileReset()
    mg.check(L.mgModsetClear(ms, st)); mg.check(L.mgAddReadsDevice(ms, packed.data_ptr(), total, do.data_ptr(), n, C.byref(nh), st)); torch.cuda.synchronize()
    ks = []
    for i in range(L.mgProfileKernels()):
        nm = C.c_char_p(); ms_ = C.c_double(); cnt = C.c_uint64(); L.mgProfileGet(i, C.byref(nm), C.byref(ms_), C.byref(cnt))
        if cnt.value: ks.append((ms_.value, nm.value.decode()))
    L.mgProfileEnable(0)
    print("      " + ", ".join("%s %.2f" % (k.replace("Kernel", "").replace("mg", ""), v) for v, k in sorted(ks, reverse=True)[:6]))
    L.modsetDestroy(ms)
print([n for n in dir(L) if "Pack" in n][:5] if False else "")
rl = 10000
one = rng.integers(0, 4, rl).astype(np.uint8)
nr = 100000
# (a) all reads identical
bases = np.tile(one, nr); offs = np.arange(nr + 1, dtype=np.int64) * rl
run("1e5 copies of one 10 kb read", bases, offs)
# (b) half random, half copies
rnd = rng.integers(0, 4, nr // 2 * rl).astype(np.uint8)
bases = np.concatenate([rnd, np.tile(one, nr // 2)])
run("half random, half copies", bases, offs)
# (c) poly-A and a two-base repeat
run("poly-A", np.zeros(nr * rl, np.uint8), offs)
run("ACACAC...", np.tile(np.array([0, 1], np.uint8), nr * rl // 2), offs)
run("random (control)", rng.integers(0, 4, nr * rl).astype(np.uint8), offs)
