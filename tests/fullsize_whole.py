#!/usr/bin/env python3
"""Whole-array parity at BASELINE size (checker side: imports oracle/).  Run by tests/test_gpu_fullsize.py in a fresh
process per variant, because the library reads its path knobs once:   python tests/fullsize_whole.py <config> [auto|flipped]

What the reference defines (modset.c:45-62, modutils.c:19-31): entry i of value[] is the i-th DISTINCT k-mer of the
(read,pos)-ordered modimizer stream (`++ms->max` at first sight), depth[i] the number of its occurrences, saturating at
65 535 (modutils.c:26).  So, given the ordered stream, the whole of value[1..max] / depth[1..max] follows by sorting:
np.unique(..., return_index, return_counts), entries ordered by first index.  The stream itself is the GPU scan's
(seqhashScanBatchDevice: k-mer, pos|isF, read of every modimizer), and it is pinned to the oracle on reads sampled over
the WHOLE batch plus whole-stream properties (reads non-decreasing, positions increasing inside a read).  The modset is
built by mgAddReadsDevice twice (clear in between): the second build runs in the configuration the library selects for
itself in steady state (flag polarity and merge-slot choice follow what the previous add saw), which is the one bench.py
times.  `flipped` forces the opposite polarity / merge-slot choice through the test knobs.

configs:  c2  BASELINE config 2 (10 Gbp ONT-like, k=21 d=64, table bits 30)
          c4  one GPU's block of config 4 (12.5 Gbp of the 100 Gbp set, 3.33 Gbp genome)
          c5  BASELINE config 5 (6 666 667 x 150 b, k=31 d=4, table bits 28)
MODGPU_FULLSIZE_SCALE=<f> shrinks the workload (development on small boxes).
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (k, w, bits, total bases, genome bases, error rate, plan kind, seeds (genome, plan, errors))
    "c2": (21, 64, 30, 10_000_000_000, 333_333_333, 0.05, "ont", (4241, 4242, 4243)),
    "c4": (21, 64, 30, 12_500_000_000, 3_333_333_333, 0.05, "ont", (4241, 4242, 4243)),
    "c5": (31, 4, 28, 6_666_667 * 150, 20_000_000, 0.005, "fixed150", (555, 556, 557)),
}


def first_occurrence_arrays(km):
    """value[1..] / depth[1..] that sequential insertion of the stream km produces: the distinct k-mers in order of first
    occurrence and their occurrence counts saturated at 65 535 (oracle/orc_modset.c orcFirstOccurrences: threads by hash class)"""
    from oracle import pyoracle as po
    OL = po.lib()
    OL.orcFirstOccurrences.restype = C.c_int64
    OL.orcFirstOccurrences.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
    n = len(km)
    flag = np.zeros(n, np.uint8); cnt = np.zeros(n, np.uint32)
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    u = OL.orcFirstOccurrences(km.ctypes.data, n, threads, flag.ctypes.data, cnt.ctypes.data)
    assert u >= 0, "orcFirstOccurrences: allocation failed"
    f = flag.view(np.bool_)
    return km[f], np.minimum(cnt[f], 65535).astype(np.uint16)


def main():
    name = sys.argv[1]
    variant = sys.argv[2] if len(sys.argv) > 2 else "auto"
    k, w, bits, total, G, err, kind, (sg, sp, se) = CONFIGS[name]
    scale = float(os.environ.get("MODGPU_FULLSIZE_SCALE", "1"))
    if scale != 1:
        total = int(total * scale); G = max(int(G * scale), 100_000)
    if variant == "flipped":
        # the steady-state choice of the library for this workload, inverted (both knobs are read once, at the first add)
        new_share_high = name in ("c2", "c4")                 # > 50 % of the modimizers become new entries (c5 at 50x: 16 %)
        dense = name == "c4"                                  # buckets more than half full
        os.environ["MODGPU_FLAG_POLARITY"] = "0" if new_share_high else "1"
        os.environ["MODGPU_MERGE_SLOTS"] = "0" if dense else "1"
    import modimizer_amd as mg
    from modimizer_amd import synth
    from oracle import pyoracle as po
    L = mg.lib()
    mg.check(L.mgSetDevice(0))
    t0 = time.time()

    if kind == "ont":
        plan = synth.ont_read_plan(total, G, sp)
    else:
        n_reads = total // 150
        plan = synth.fixed_read_plan(n_reads, 150, G, sp); total = n_reads * 150
    starts, offs, strands = plan
    n_reads = len(starts)
    d_g = mg.DeviceBuffer(L.mgPackedWords(G) * 4)
    mg.check(L.mgSynthGenome(d_g.ptr, G, sg, None))
    d_s = mg.DeviceBuffer.from_numpy(starts); d_of = mg.DeviceBuffer.from_numpy(offs); d_st = mg.DeviceBuffer.from_numpy(strands)
    d_r = mg.DeviceBuffer(L.mgPackedWords(total) * 4)
    mg.check(L.mgSynthReads(d_g.ptr, G, d_s.ptr, d_of.ptr, d_st.ptr, n_reads, total, err, se, d_r.ptr, None))
    mg.check(L.mgStreamSynchronize(None))
    d_g.free(); d_s.free(); d_st.free()

    sh = mg.seqhashCreate(k, w, 17)
    ms = mg.modsetCreate(sh, bits)
    n = C.c_uint64()
    mg.check(L.mgAddReadsDevice(ms, d_r.ptr, total, d_of.ptr, n_reads, C.byref(n), None))
    S1, U1 = n.value, ms.contents.max
    mg.check(L.mgModsetClear(ms, None))
    mg.check(L.mgAddReadsDevice(ms, d_r.ptr, total, d_of.ptr, n_reads, C.byref(n), None))     # the steady-state configuration
    S, U = n.value, ms.contents.max
    assert (S, U) == (S1, U1), ("two builds of the same batch differ", S1, U1, S, U)
    mg.check(L.modsetSyncToHost(ms, 0))
    value = np.ctypeslib.as_array(ms.contents.value, (U + 1,))[1:].copy()
    depth = np.ctypeslib.as_array(ms.contents.depth, (U + 1,))[1:].copy()
    L.modsetDestroy(ms)
    t_build = time.time() - t0

    # ---- the ordered modimizer stream, from the scan entry point ---------------------------------------------------
    cap = S + 4096
    d_k = mg.DeviceBuffer(cap * 8); d_p = mg.DeviceBuffer(cap * 4); d_i = mg.DeviceBuffer(cap * 4)
    d_c = mg.DeviceBuffer(64); d_w = mg.DeviceBuffer(L.mgScanWorkBytes(total, n_reads, cap))
    mg.check(L.seqhashScanBatchDevice(sh, d_r.ptr, total, d_of.ptr, n_reads, d_k.ptr, d_p.ptr, d_i.ptr, cap, d_c.ptr, d_w.ptr, None))
    cnt = d_c.to_numpy(np.uint64, 4)
    assert int(cnt[0]) == S and int(cnt[1]) == 0, ("scan entry point and build disagree on the number of modimizers", cnt, S)
    d_w.free()
    km = d_k.to_numpy(np.uint64, S); d_k.free()
    pf = d_p.to_numpy(np.uint32, S); d_p.free()
    rd = d_i.to_numpy(np.uint32, S); d_i.free()
    # whole-stream order: reads non-decreasing; inside a read positions strictly increase (seqhash.c:184: pos = iMin)
    same = rd[1:] == rd[:-1]
    assert np.all(rd[1:] >= rd[:-1]) and rd[-1] < n_reads
    pos = pf & np.uint32(mg.MG_POS_MASK)
    assert np.all(pos[1:][same] > pos[:-1][same])
    del same
    first = np.searchsorted(rd, np.arange(n_reads + 1, dtype=np.uint32))
    # reads sampled over the WHOLE batch against the oracle: k-mers, positions, strands
    rng = np.random.default_rng(11)
    n_sample = 1000 if kind == "ont" else 20000
    sample = np.unique(np.concatenate([[0, n_reads - 1], rng.integers(0, n_reads, n_sample - 2)]))
    oh = po.Hasher(k, w, 17)
    offs64 = offs.astype(np.int64)
    d_b = mg.DeviceBuffer(1 << 20)
    checked = 0
    for r in sample:
        a, b = int(offs64[r]), int(offs64[r + 1])
        a16 = a - a % 16
        nb = b - a16
        if nb > d_b.nbytes:
            d_b.free(); d_b = mg.DeviceBuffer(2 * nb)
        mg.check(L.mgUnpackDevice(C.c_void_p(d_r.ptr.value + a16 // 4), nb, d_b.ptr, None))
        bases = d_b.to_numpy(np.uint8, nb)[a - a16:]
        ek, ep, ef = oh.scan(bases)
        lo, hi = int(first[r]), int(first[r + 1])
        assert hi - lo == len(ek), ("read", int(r), hi - lo, len(ek))
        assert np.array_equal(km[lo:hi], ek) and np.array_equal(pos[lo:hi], ep.astype(np.uint32)) \
            and np.array_equal((pf[lo:hi] >> 31).astype(np.uint8), ef), ("read", int(r))
        checked += hi - lo
    d_b.free(); d_r.free(); d_of.free()
    del pf, rd, pos, first
    t_scan = time.time() - t0 - t_build

    # ---- first-occurrence order and counts of the whole stream, on the host ------------------------------------------
    want_value, want_depth = first_occurrence_arrays(km)
    assert len(want_value) == U, ("distinct modimizers", len(want_value), "entries", U)
    bad = np.flatnonzero(want_value != value)
    assert bad.size == 0, ("value[] differs from first-occurrence order at", bad[:5] + 1, "of", U)
    bad = np.flatnonzero(want_depth != depth)
    assert bad.size == 0, ("depth[] differs at", bad[:5] + 1, "of", U)
    print("fullsize_whole %s %s: %d bases, %d reads, %d modimizers, %d entries: value[] and depth[] pinned entirely; "
          "%d sampled reads (%d modimizers) == oracle; build %.1f s, scan+sample %.1f s, host reconstruction %.1f s"
          % (name, variant, total, n_reads, S, U, len(sample), checked, t_build, t_scan, time.time() - t0 - t_build - t_scan))
    print("FULLSIZE_WHOLE_OK")


if __name__ == "__main__":
    main()
