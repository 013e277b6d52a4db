#!/usr/bin/env python3
"""dev probe: reads from a genome with the repeat structure real genomes have (an Alu-like family with poly-A tails, satellite arrays,
microsatellites, N gaps read as A) against reads from an iid genome of the same size: ms per Gbp and the kernels it goes to"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import modimizer_amd as mg
from modimizer_amd import synth
L = mg.lib(); dev = torch.device("cuda", 0); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(11)
G = 100_000_000
def mutate(x, rate):
    x = x.copy(); m = rng.random(len(x)) < rate; x[m] = (x[m] + rng.integers(1, 4, int(m.sum()))) & 3; return x
def repeat_genome():
    parts = []; n = 0
    alu = rng.integers(0, 4, 300).astype(np.uint8); mono = rng.integers(0, 4, 171).astype(np.uint8)
    frac = {"alu": 0.10, "sat": 0.03, "ca": 0.01, "gap": float(os.environ.get("PROBE_GAP", "0"))}                      # shares of the genome's LENGTH
    have = dict.fromkeys(frac, 0)
    while n < G:
        kind = next((k for k in frac if have[k] < frac[k] * n), None) if n else None
        if kind == "alu":   p = np.concatenate([mutate(alu, 0.10), np.zeros(int(rng.integers(15, 45)), np.uint8)])          # an Alu-like copy with its poly-A tail
        elif kind == "sat": p = np.concatenate([mutate(mono, 0.02) for _ in range(int(rng.integers(200, 3000)))])           # a satellite array
        elif kind == "ca":  p = np.tile(np.array([1, 0], np.uint8), int(rng.integers(10, 60)))                              # (CA)n
        elif kind == "gap": p = np.zeros(int(rng.integers(1000, 200000)), np.uint8)                                         # an N gap, read as A
        else:               p = rng.integers(0, 4, int(rng.integers(200, 6000))).astype(np.uint8)
        if kind: have[kind] += len(p)
        parts.append(p); n += len(p)
    return np.concatenate(parts)[:G]
def run(name, genome):
    total = 1_000_000_000
    starts, offs, strands = synth.ont_read_plan(total, len(genome), 21, n50=20000, lo=500, hi=200000)
    bases = synth.reads_from_genome(genome, starts, offs, strands, 0.05, 22)
    total = len(bases); n = len(starts)
    hb = torch.from_numpy(bases).to(dev); packed = torch.empty(L.mgPackedWords(total), dtype=torch.int32, device=dev)
    mg.check(L.mgPackDevice(hb.data_ptr(), total, packed.data_ptr(), st)); do = torch.from_numpy(offs.astype(np.int64)).to(dev)
    sh = mg.seqhashCreate(21, 64, 17); ms = mg.modsetCreate(sh, 28); nh = C.c_uint64(0)
    for rep in range(2):
        mg.check(L.mgModsetClear(ms, st)); torch.cuda.synchronize(); t0 = time.time()
        mg.check(L.mgAddReadsDevice(ms, packed.data_ptr(), total, do.data_ptr(), n, C.byref(nh), st)); torch.cuda.synchronize(); dt = time.time() - t0
    mg.check(L.modsetSyncToHost(ms, 0)); v, d, _ = mg.modset_arrays(ms)
    top = np.sort(d[1:].astype(np.int64))[::-1][:5]
    print("%-28s %.2f Gbp, %d modimizers, %d entries, deepest %s: %.2f ms per Gbp" % (name, total / 1e9, nh.value, ms.contents.max, list(top), dt * 1e3 / (total / 1e9)))
    L.mgProfileEnable(1); L.mgProfileReset()
    mg.check(L.mgModsetClear(ms, st)); mg.check(L.mgAddReadsDevice(ms, packed.data_ptr(), total, do.data_ptr(), n, C.byref(nh), st)); torch.cuda.synchronize()
    ks = []
    for i in range(L.mgProfileKernels()):
        nm = C.c_char_p(); ms_ = C.c_double(); cnt = C.c_uint64(); L.mgProfileGet(i, C.byref(nm), C.byref(ms_), C.byref(cnt))
        if cnt.value: ks.append((ms_.value, nm.value.decode()))
    L.mgProfileEnable(0)
    print("      " + ", ".join("%s %.2f" % (k.replace("Kernel", "").replace("mg", ""), x) for x, k in sorted(ks, reverse=True)[:8]))
    L.modsetDestroy(ms)
run("iid genome", rng.integers(0, 4, G).astype(np.uint8))
run("genome with repeats", repeat_genome())
