"""BASELINE configs 2, 3, 5 and one GPU's block of config 4 at their full sizes on the GPU, through the C ABI.

The oracle cannot build a 3 Gbp reference or sketch 1 Gbp of short reads in seconds, so at full size the checks are
(i) bit-exact comparisons with the oracle on a sample it does finish quickly, tied to the full-size result by a
property of the reference's algorithm — indices are handed out in order of first occurrence (modset.c:57), so the
modset of a PREFIX of the input is a prefix of the modset of the whole input — and (ii) size-independent properties
(histogram sums, every seed of a sampled read against a host-side dictionary of the GPU-built value[] array, Q-line
tallies against the seeds).
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

import modimizer_amd as mg
from modimizer_amd import synth
from oracle import pyoracle as po
import util

pytestmark = pytest.mark.gpu


def device_reads(L, genome_bases, genome_seed, plan, err, err_seed, keep_genome=False):
    starts, offs, strands = plan
    total = int(offs[-1])
    d_g = mg.DeviceBuffer(L.mgPackedWords(genome_bases) * 4)
    mg.check(L.mgSynthGenome(d_g.ptr, genome_bases, genome_seed, None))
    d_s = mg.DeviceBuffer.from_numpy(starts); d_of = mg.DeviceBuffer.from_numpy(offs); d_st = mg.DeviceBuffer.from_numpy(strands)
    d_r = mg.DeviceBuffer(L.mgPackedWords(total) * 4)
    mg.check(L.mgSynthReads(d_g.ptr, genome_bases, d_s.ptr, d_of.ptr, d_st.ptr, len(starts), total, err, err_seed, d_r.ptr, None))
    mg.check(L.mgStreamSynchronize(None))
    if not keep_genome:
        d_g.free()
    return d_r, d_of, (d_g if keep_genome else None)


def unpack_range(L, d_packed, first_base, n_bases):
    """bases [first_base, first_base + n_bases) of a packed device stream as host bytes (first_base a multiple of 16)"""
    assert first_base % 16 == 0
    d_b = mg.DeviceBuffer(max(n_bases, 16))
    src = C.c_void_p(d_packed.ptr.value + first_base // 4)
    mg.check(L.mgUnpackDevice(src, n_bases, d_b.ptr, None))
    out = d_b.to_numpy(np.uint8, n_bases)
    d_b.free()
    return out


def test_config5_full_size_depth_histogram():
    """50x of a 20 Mbp genome as 6 666 667 reads of 150 b, k=31 d=4, table bits 28 (modutils.c:19-63): the whole set on
    the GPU; the first 20 000 reads also through the oracle, bit-exact, and as a prefix of the full-size modset"""
    L = mg.lib()
    k, w, bits = 31, 4, 28
    G, n_reads, rl = 20_000_000, 6_666_667, 150
    plan = synth.fixed_read_plan(n_reads, rl, G, 556)
    total = n_reads * rl
    d_r, d_of, _ = device_reads(L, G, 555, plan, 0.005, 557)
    sh = mg.seqhashCreate(k, w, 17)
    ms = mg.modsetCreate(sh, bits)
    n = C.c_uint64()
    mg.check(L.mgAddReadsDevice(ms, d_r.ptr, total, d_of.ptr, n_reads, C.byref(n), None))
    S, U = n.value, ms.contents.max
    assert 0.18 < S / total < 0.22 and 0 < U < S and U < (1 << 26) - 1            # ~N/5 modimizers; under the table's capacity (modset.c:58)
    d_h = mg.DeviceBuffer(65536 * 8)
    mg.check(L.mgMemsetD(d_h.ptr, 0, 65536 * 8, None))
    mg.check(L.modsetDepthHistogramDevice(ms, d_h.ptr, None))
    h = d_h.to_numpy(np.uint64, 65536)
    assert int(h.sum()) == U and h[0] == 0
    assert h[65535] == 0 and int((h * np.arange(65536, dtype=np.uint64)).sum()) == S     # 50x: nothing saturates; depth sum == hash count
    # 50x coverage at 0.5 % errors: a peak of error k-mers at depth 1 and a genomic mode well above it
    genomic_mode = int(h[5:200].argmax()) + 5
    assert h[1] > h[2] > h[3] and 15 <= genomic_mode <= 60

    # the first 20 000 reads: host mirror of the generator, the oracle, and the product on that prefix alone
    m = 20_000
    genome = synth.iid_bases(G, 555)
    sub_plan = (plan[0][:m], plan[1][:m + 1], plan[2][:m])
    host = synth.reads_from_genome(genome, *sub_plan, 0.005, 557)
    assert np.array_equal(host, unpack_range(L, d_r, 0, m * rl))                     # device generator == host mirror
    oh = po.Hasher(k, w, 17); oms = po.Modset(oh, 24)
    offs = sub_plan[1].astype(np.int64)
    tot = sum(oms.add_sequence(host[offs[r]:offs[r + 1]]) for r in range(m))
    ms2 = mg.modsetCreate(sh, 24)
    n2 = C.c_uint64()
    mg.check(L.mgAddReadsDevice(ms2, d_r.ptr, m * rl, d_of.ptr, m, C.byref(n2), None))
    assert n2.value == tot
    mg.check(L.modsetSyncToHost(ms2, 0))
    v2, dep2, _ = mg.modset_arrays(ms2)
    assert ms2.contents.max == oms.max and np.array_equal(v2[1:], oms.values()[1:]) and np.array_equal(dep2[1:], oms.depths()[1:])
    # prefix property: the full-size set starts with exactly these entries, in this order, with depths at least as large
    mg.check(L.modsetSyncToHost(ms, 0))
    vf = np.ctypeslib.as_array(ms.contents.value, (U + 1,))
    df = np.ctypeslib.as_array(ms.contents.depth, (U + 1,))
    assert np.array_equal(vf[1:oms.max + 1], oms.values()[1:])
    assert np.all(df[1:oms.max + 1] >= oms.depths()[1:])
    assert len(np.unique(vf[1:])) == U
    assert int(df[1:].astype(np.int64).sum()) == S
    L.modsetDestroy(ms); L.modsetDestroy(ms2)
    d_r.free(); d_of.free(); d_h.free()


@pytest.mark.parametrize("name,total,G,bits", [("config 2", 10_000_000_000, 333_333_333, 30),
                                               ("one block of config 4", 12_500_000_000, 3_333_333_333, 30)])
def test_config2_and_config4_block_prefix_exact(name, total, G, bits):
    """BASELINE config 2 (10 Gbp at 30x of a 333 Mbp genome) and one GPU's block of config 4 (12.5 Gbp of the 100 Gbp set:
    3.75x of a 3.33 Gbp genome, 1.66e8 distinct modimizers), k=21 d=64, built at full size by mgAddReadsDevice; the first
    ~40 Mbp of reads also through the oracle: bit-exact on their own, and as the prefix of the full-size value[] / depth[]"""
    L = mg.lib()
    k, w = 21, 64
    plan = synth.ont_read_plan(total, G, 4242)
    d_r, d_of, _ = device_reads(L, G, 4241, plan, 0.05, 4243)
    n_reads = len(plan[0])
    sh = mg.seqhashCreate(k, w, 17)
    ms = mg.modsetCreate(sh, bits)
    n = C.c_uint64()
    mg.check(L.mgAddReadsDevice(ms, d_r.ptr, total, d_of.ptr, n_reads, C.byref(n), None))
    S, U = n.value, ms.contents.max
    assert abs(S / (total / 64.0) - 1) < 0.01 and 0 < U <= S, name
    d_h = mg.DeviceBuffer(65536 * 8)
    mg.check(L.mgMemsetD(d_h.ptr, 0, 65536 * 8, None))
    mg.check(L.modsetDepthHistogramDevice(ms, d_h.ptr, None))
    h = d_h.to_numpy(np.uint64, 65536)
    assert int(h.sum()) == U and h[0] == 0 and h[65535] == 0
    assert int((h * np.arange(65536, dtype=np.uint64)).sum()) == S

    offs = plan[1].astype(np.int64)
    m = int(np.searchsorted(offs, 40_000_000))                        # reads wholly inside the first ~40 Mbp
    host = unpack_range(L, d_r, 0, int(offs[m]))
    oh = po.Hasher(k, w, 17); oms = po.Modset(oh, 24)
    tot = sum(oms.add_sequence(host[offs[r]:offs[r + 1]]) for r in range(m))
    ms2 = mg.modsetCreate(sh, 24)
    n2 = C.c_uint64()
    mg.check(L.mgAddReadsDevice(ms2, d_r.ptr, int(offs[m]), d_of.ptr, m, C.byref(n2), None))
    assert n2.value == tot
    mg.check(L.modsetSyncToHost(ms2, 0))
    v2, dep2, _ = mg.modset_arrays(ms2)
    assert ms2.contents.max == oms.max and np.array_equal(v2[1:], oms.values()[1:]) and np.array_equal(dep2[1:], oms.depths()[1:])
    mg.check(L.modsetSyncToHost(ms, 0))
    vf = np.ctypeslib.as_array(ms.contents.value, (U + 1,))
    df = np.ctypeslib.as_array(ms.contents.depth, (U + 1,))
    assert np.array_equal(vf[1:oms.max + 1], oms.values()[1:]), name        # indices in order of first occurrence (modset.c:57)
    assert np.all(df[1:oms.max + 1] >= oms.depths()[1:])
    assert int(df[1:].astype(np.int64).sum()) == S
    # a sample of the entries behind the prefix: each is a k-mer the scan of SOME read produces, and finds its own index
    rng = np.random.default_rng(7)
    pick = np.sort(rng.choice(np.arange(oms.max + 1, U + 1), 200_000, replace=False))
    d_v = mg.DeviceBuffer.from_numpy(np.ascontiguousarray(vf[pick])); d_o = mg.DeviceBuffer(len(pick) * 4)
    mg.check(L.modsetFindBatchDevice(ms, d_v.ptr, len(pick), d_o.ptr, None))
    assert np.array_equal(d_o.to_numpy(np.uint32, len(pick)), pick.astype(np.uint32))
    L.modsetDestroy(ms); L.modsetDestroy(ms2)
    d_r.free(); d_of.free(); d_h.free(); d_v.free(); d_o.free()


def test_config3_full_size_reference_and_queries(tmp_path):
    """a 3 Gbp reference in 24 sequences of 125 Mbp (table bits 28; its occurrences stay under modmap's 1 << 26 cap,
    modmap.c:111,363) built by mgReferenceRead; a 10 Gbp batch of ONT-like reads from it queried on the device"""
    L = mg.lib()
    k, w, bits = 21, 64, 28
    n_seq, seq_len = 24, 125_000_000
    G = n_seq * seq_len
    sh = mg.seqhashCreate(k, w, 17); oh = po.Hasher(k, w, 17)
    ms = mg.modsetCreate(sh, bits)
    d_g = mg.DeviceBuffer(L.mgPackedWords(G) * 4)
    mg.check(L.mgSynthGenome(d_g.ptr, G, 333, None))
    genome = unpack_range(L, d_g, 0, G)                                  # 3 GB of host bytes, as a FASTA parser would hold them
    assert np.array_equal(genome[:100_000], synth.iid_bases(100_000, 333))
    ref_off = (np.arange(n_seq + 1, dtype=np.int64) * seq_len)
    names = (C.c_char_p * n_seq)(*[b"chr%d" % (i + 1) for i in range(n_seq)])
    ref = L.mgReferenceCreate(ms, 1 << 26)                              # modmap.c:363
    out = str(tmp_path / "ref.txt")
    with mg.CFile(out, "w") as f:
        assert L.mgReferenceRead(ref, genome.ctypes.data, ref_off.ctypes.data, n_seq, names, True, f) == 0
    R = C.cast(ref, C.POINTER(mg.MgReference)).contents
    U, occ = ms.contents.max, R.max
    lines = open(out).read().splitlines()
    assert lines[0] == "  %d hashes from %d reference sequences, total length %d" % (occ, n_seq, G)
    c1, c2, cM = (int(x) for x in re.match(r"\s+(\d+) copy 1, (\d+) copy 2, (\d+) multiple", lines[1]).groups())
    assert c1 + c2 + cM == U and occ < (1 << 26) and abs(occ / (G / 64.0) - 1) < 0.01
    assert c1 > 0.99 * U                                               # a random genome: almost every modimizer is single copy
    # prefix property against the oracle: the first 40 Mbp of chr1 alone
    m = 40_000_000
    oms = po.Modset(oh, 22)
    oms.add_sequence(genome[:m])
    v = np.ctypeslib.as_array(ms.contents.value, (U + 1,))
    assert np.array_equal(v[1:oms.max + 1], oms.values()[1:])
    # occurrences of chr1's prefix: (index, offset, id) are the oracle's modimizers in order (modmap.c:112-116)
    ek, ep, ef = oh.scan(genome[:m])
    n0 = len(ek)
    r_index = np.ctypeslib.as_array(R.index, (occ,)); r_off = np.ctypeslib.as_array(R.offset, (occ,)); r_id = np.ctypeslib.as_array(R.id, (occ,))
    assert np.array_equal(r_off[:n0], ep.astype(np.uint32)) and not r_id[:n0].any()
    assert np.array_equal(v[r_index[:n0]], ek)
    # per-sequence occurrence counts add up, ids are sorted (sequences in order)
    assert np.all(np.diff(r_id.astype(np.int64)) >= 0) and r_id[-1] == n_seq - 1
    del genome

    # ---- a 10 Gbp batch of queries, device resident --------------------------------------------------
    total = 10_000_000_000
    plan = synth.ont_read_plan(total, G, 4000)
    n_reads = len(plan[0])
    d_s = mg.DeviceBuffer.from_numpy(plan[0]); d_of = mg.DeviceBuffer.from_numpy(plan[1]); d_st = mg.DeviceBuffer.from_numpy(plan[2])
    d_r = mg.DeviceBuffer(L.mgPackedWords(total) * 4)
    mg.check(L.mgSynthReads(d_g.ptr, G, d_s.ptr, d_of.ptr, d_st.ptr, n_reads, total, 0.05, 5000, d_r.ptr, None))
    cap = int(total / w * 1.3) + (1 << 16)
    d_ix = mg.DeviceBuffer(cap * 4); d_ps = mg.DeviceBuffer(cap * 4); d_rd = mg.DeviceBuffer(cap * 4)
    ns = C.c_uint64()
    mg.check(L.mgQueryReadsDevice(ms, d_r.ptr, total, d_of.ptr, n_reads, d_ix.ptr, d_ps.ptr, d_rd.ptr, cap, C.byref(ns), None))
    S = ns.value
    assert abs(S / (total / 64.0) - 1) < 0.01
    s_ix = d_ix.to_numpy(np.uint32, S); s_ps = d_ps.to_numpy(np.uint32, S); s_rd = d_rd.to_numpy(np.uint32, S)
    assert np.all(np.diff(s_rd.astype(np.int64)) >= 0) and s_rd[-1] < n_reads and s_ix.max() <= U
    hit = float((s_ix != 0).mean())
    assert 0.30 < hit < 0.40                                            # (1 - 0.05)^21 = 0.34 of the k-mers survive 5 % substitutions
    first = np.searchsorted(s_rd, np.arange(n_reads + 1))               # seeds of read r: [first[r], first[r+1])
    # sampled reads against the oracle: positions, strands, and the index of every k-mer via a dictionary of value[]
    order = np.argsort(v[1:], kind="stable")
    vs = v[1:][order]
    rng = np.random.default_rng(5)
    offs = plan[1].astype(np.int64)
    sample = np.concatenate([[0, n_reads - 1], rng.integers(0, n_reads, 998)])
    for r in sample:
        a, b = int(offs[r]), int(offs[r + 1])
        a16 = a - a % 16
        bases = unpack_range(L, d_r, a16, b - a16)[a - a16:]
        ek, ep, ef = oh.scan(bases)
        lo, hi = int(first[r]), int(first[r + 1])
        assert hi - lo == len(ek), r
        assert np.array_equal(s_ps[lo:hi] & mg.MG_POS_MASK, ep.astype(np.uint32)) and np.array_equal((s_ps[lo:hi] >> 31).astype(np.uint8), ef)
        at = np.searchsorted(vs, ek)
        at[at >= len(vs)] = 0
        want = np.where(vs[at] == ek, order[at] + 1, 0).astype(np.uint32)
        assert np.array_equal(s_ix[lo:hi], want), r
    # Q lines (modmap.c:208-211) of the first 1500 reads from host bytes: tallies == what the device seeds say
    nq = 1500
    qb = unpack_range(L, d_r, 0, int(offs[nq]))
    qn = (C.c_char_p * nq)(*[b"q%d" % i for i in range(nq)])
    qout = str(tmp_path / "q.txt")
    with mg.CFile(qout, "w") as f:
        assert L.mgQueryProcess(ref, qb.ctypes.data, offs[:nq + 1].ctypes.data, nq, qn, f) == 0
    info = np.ctypeslib.as_array(ms.contents.info, (U + 1,))
    qlines = [l for l in open(qout).read().splitlines() if l.startswith("Q\t")]
    assert len(qlines) == nq
    for r, l in enumerate(qlines):
        nm, ln, rest = l.split("\t")[1:4]
        miss, k1, k2, kM = (int(x) for x in re.match(r"(\d+) miss, (\d+) copy1, (\d+) copy2, (\d+) multi", rest).groups())
        ix = s_ix[first[r]:first[r + 1]]
        cls = info[ix[ix != 0]] & 3
        assert int(ln) == offs[r + 1] - offs[r] and miss == int((ix == 0).sum())
        assert (k1, k2, kM) == (int((cls == 1).sum()), int((cls == 2).sum()), int((cls == 3).sum()))
    L.mgReferenceDestroy(ref)
    L.modsetDestroy(ms)
    for d in (d_g, d_r, d_s, d_of, d_st, d_ix, d_ps, d_rd):
        d.free()


@pytest.mark.parametrize("variant", ["both"])
@pytest.mark.parametrize("config", ["c2", "c4", "c5"])
def test_whole_value_and_depth_arrays_at_full_size(config, variant):
    """EVERY index of value[] / depth[] at BASELINE size (1.03e8 entries of config 2, 1.65e8 of a config-4 block, 3.3e7 of
    config 5): the GPU scan's ordered modimizer stream — pinned to the oracle ENTIRELY (every read of the batch scanned by
    oracle/orc_seqhash.c orcScanCheckMany on the host's cores and compared in place) —
    gives, by a host-side sort (first-occurrence order, counts saturated at 65 535: modset.c:57, modutils.c:26), the arrays
    the build must produce; compared entirely.  `auto`: the configuration the library selects in steady state (second of
    two builds: flag polarity and merge-slot choice follow the previous add), the one bench.py times; `flipped`: the other
    polarity / merge-slot choice forced and, round 6, the other table sizing (configs 2 and 4 brought to load 0.7 after the dedup
    kernel's count, config 5 left at its occurrences' bound).  tests/fullsize_whole.py `both`: one process per config, the reads
    made and the stream pinned once, the flipped build checked against the same expectation (the knobs are re-read: mgReloadKnobs)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "tests", "fullsize_whole.py"), config, variant],
                       capture_output=True, text=True, timeout=3000, env=dict(os.environ, PYTHONPATH=util.ROOT))
    assert r.returncode == 0 and "FULLSIZE_WHOLE_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_config3_reference_arrays_whole_at_full_size():
    """config 3's 3 Gbp reference (24 x 125 Mbp, mgReferenceRead = referenceFastaRead + referencePack, modmap.c:74-134): the
    whole modimizer stream of the 24 sequences against the oracle, then value[], the info copy classes, ref->index / offset /
    id of all 4.7e7 occurrences, ref->depth, loc[] and rev[] rebuilt on the host from that stream and compared entirely
    (tests/fullsize_whole.py c3ref)"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "tests", "fullsize_whole.py"), "c3ref"],
                       capture_output=True, text=True, timeout=3000, env=dict(os.environ, PYTHONPATH=util.ROOT))
    assert r.returncode == 0 and "FULLSIZE_WHOLE_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_reference_default_parameters_whole_at_full_size():
    """the reference's own defaults (modmap.c:314-317, modutils.c:140: k = 19, w = 31, seed 17 -- a non-power-of-two d, the
    MG_MODE_ANY scan) on config 2's 10 Gbp of reads: whole stream against the oracle, whole value[] / depth[]"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "tests", "fullsize_whole.py"), "refdef", "auto"],
                       capture_output=True, text=True, timeout=3000, env=dict(os.environ, PYTHONPATH=util.ROOT))
    assert r.returncode == 0 and "FULLSIZE_WHOLE_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_config3_query_seeds_whole_at_full_size():
    """config 3's query batch: every one of the 1.56e8 seeds of a 10 Gbp batch against the 3 Gbp reference -- index (from a sorted
    copy of value[]), position, strand and read -- with the batch's k-mer stream pinned entirely to the oracle, on the direct and
    the partitioned lookup path (tests/fullsize_whole.py c3q)"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "tests", "fullsize_whole.py"), "c3q"],
                       capture_output=True, text=True, timeout=3000, env=dict(os.environ, PYTHONPATH=util.ROOT))
    assert r.returncode == 0 and "FULLSIZE_WHOLE_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("config,scale", [("c2", 0.004), ("c2", 0.03), ("c2", 0.1), ("c5", 0.2), ("refdef", 0.05)])
def test_whole_arrays_at_sizes_between_the_small_tests_and_baseline(config, scale):
    """The same whole-stream / whole-array comparison at batch sizes where the launch shapes change with the size (round 4): 40 Mbp
    (0.6 M modimizers: the atomic path whatever the table), 0.3 Gbp (8192 scan workers of several tiles, rank-lookup slices of 2^17
    ordinals), 1 Gbp (what the file entry points hand on: 30 000 workers, slices of 2^18), a fifth of config 5, a twentieth of the
    reference-default run."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "tests", "fullsize_whole.py"), config, "auto"],
                       capture_output=True, text=True, timeout=1500, env=dict(os.environ, PYTHONPATH=util.ROOT, MODGPU_FULLSIZE_SCALE=str(scale)))
    assert r.returncode == 0 and "FULLSIZE_WHOLE_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
