"""The per-read facade's short-read leg (mg_host.c mgIterScanHost: the library's own scalar loop behind modRCiterator for
reads below the launch-latency crossover; reference semantics seqhash.c:60-79,154-196) pinned WITHOUT a GPU against the
golden vectors the compiled reference produced, the oracle, and -- where oracle/_ref is built -- the reference library."""
import ctypes as C

import numpy as np
import pytest

import modimizer_amd as mg
from oracle import pyoracle as po
import util


def host_scan(sh, bases):
    """mgIterScanHost -> (kmer, pos, isF) out of the replay block {n, n k-mers, n words pos | isF << 31}"""
    L = mg.lib()
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    addr = L.mgIterScanHost(sh, bases.ctypes.data, len(bases))
    assert addr
    n = int(C.cast(C.c_void_p(addr), C.POINTER(C.c_uint64))[0])
    km = np.ctypeslib.as_array(C.cast(C.c_void_p(addr + 8), C.POINTER(C.c_uint64)), (max(n, 1),))[:n].copy()
    pf = np.ctypeslib.as_array(C.cast(C.c_void_p(addr + 8 * (n + 1)), C.POINTER(C.c_uint32)), (max(n, 1),))[:n].copy()
    mg._libc.free(addr)
    return km, (pf & np.uint32(mg.MG_POS_MASK)).astype(np.int32), (pf >> 31).astype(np.uint8)


@pytest.mark.parametrize("ci", range(9))
def test_golden_vectors(ci):
    """all nine golden (k, d, seed) configs x their edge reads (empty, len < k, len == k, homopolymers, palindromes with
    hashF == hashR ties, ...): vectors written by the compiled reference (tests/golden/make_golden.py)"""
    k, w, seed = util.scan_configs()[ci]
    sh = mg.seqhashCreate(k, w, seed)
    n = 0
    for name, b, gk, gp, gf in util.scan_cases(ci):
        a, p, f = host_scan(sh, b)
        assert np.array_equal(a, gk) and np.array_equal(p, gp) and np.array_equal(f, gf), (ci, name)
        n += len(gk)
    assert n > 0


@pytest.mark.parametrize("k,w,seed", [(21, 64, 17), (31, 4, 17), (19, 31, 17), (16, 32, 0), (11, 1, 3), (1, 1, 17), (2, 3, 5),
                                      (31, 97, 9), (27, 1024, 17), (5, 2, 17), (31, 1, 5), (12, 7, 99), (21, 96, 17),
                                      (31, 2147483647, 1), (17, 1 << 20, 4)])
def test_vs_oracle_random(k, w, seed):
    sh = mg.seqhashCreate(k, w, seed); oh = po.Hasher(k, w, seed)
    rng = np.random.default_rng(k * 1009 + w % 1000)
    tot = 0
    for n in [0, 1, k - 1, k, k + 1, 2 * k, 63, 64, 65, 150, 151, 257, 1023, 1024 + k - 1, 1024 + k, 4095, 4096, 20011]:
        b = rng.integers(0, 4, n).astype(np.uint8)
        if n >= 4 * k:
            b[n // 2:n // 2 + 2 * k] = 0                                  # a homopolymer run: hashF == hashR ties inside
            b[n // 4 + k:n // 4 + 2 * k] = 3 - b[n // 4:n // 4 + k][::-1]    # a k-mer followed by its reverse complement
        a, p, f = host_scan(sh, b)
        ek, ep, ef = oh.scan(b)
        assert np.array_equal(a, ek) and np.array_equal(p, ep) and np.array_equal(f, ef), (k, w, n)
        tot += len(a)
    assert tot > 0 or w > 1000


def test_known_answers():
    """read 0 of `gen 1000 10000` -> 150 / 2409 / 316 modimizers with these first entries"""
    b = np.zeros(10000, np.uint8)
    xv = 0x9E3779B97F4A7C15
    for i in range(10000):
        xv ^= (xv << 13) & 0xFFFFFFFFFFFFFFFF; xv ^= xv >> 7; xv ^= (xv << 17) & 0xFFFFFFFFFFFFFFFF
        b[i] = xv >> 62
    for (k, w), n, first in (((21, 64), 150, [(1, 0x142be04b2e6, 1), (102, 0x3f9f59c97da, 0), (125, 0x3483585ba75, 0)]),
                             ((31, 4), 2409, [(1, 0x0ae3dd91c7bd05fa, 0), (3, 0x02be04b2e620d17a, 1), (21, 0x220d17ac66102775, 1)]),
                             ((19, 31), 316, [(6, 0x2f812cb988, 1), (86, 0x2e6b397896, 1), (115, 0x1ba759fe7d, 0)])):
        sh = mg.seqhashCreate(k, w, 17)
        assert sh.contents.factor1 == 0x49308bb9003cb3ad
        a, p, f = host_scan(sh, b)
        assert len(a) == n
        assert [(int(p[i]), int(a[i]), int(f[i])) for i in range(3)] == first


def test_bytes_are_taken_modulo_4():
    """FASTQ keeps bytes dna2indexConv maps below 0 ((char) -2, seqio.c:328-331): the packer keeps their low two bits
    (mg_pack.c), and so must this loop, or the two legs of modRCiterator would disagree on such reads"""
    sh = mg.seqhashCreate(21, 8, 17)
    rng = np.random.default_rng(3)
    b = rng.integers(0, 4, 3000).astype(np.uint8)
    junk = b.copy(); junk[::7] |= 0xFC
    assert all(np.array_equal(x, y) for x, y in zip(host_scan(sh, b), host_scan(sh, junk)))


def test_crossover_setter():
    L = mg.lib()
    was = L.mgIterHostBelow(-1)
    assert was >= 0
    assert L.mgIterHostBelow(123) == was and L.mgIterHostBelow(-1) == 123
    assert L.mgIterHostBelow(1 << 30) == 123 and L.mgIterHostBelow(-1) == was      # 1 << 30: the defaults by w again


needs_ref = pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref not built (reference tree absent)")


@needs_ref
@pytest.mark.parametrize("k,w,seed", [(21, 64, 17), (31, 4, 17), (19, 31, 17), (16, 32, 0), (12, 7, 99), (1, 2, 3), (31, 1, 5)])
def test_vs_compiled_reference(k, w, seed):
    """the reference's own modRCiterator / modRCnext (oracle/_ref/libmodref.so, compiled from the sources in place)"""
    R = po.ref()
    rsh = R.seqhashCreate(k, w, seed)
    sh = mg.seqhashCreate(k, w, seed)
    rng = np.random.default_rng(k * 131 + w)
    for n in [0, 1, k - 1, k, k + 1, 2 * k, 150, 257, 1000, 4000]:
        b = rng.integers(0, 4, n).astype(np.uint8)
        if n > 100:
            b[n // 3:n // 3 + 50] = 0
        assert all(np.array_equal(x, y) for x, y in zip(host_scan(sh, b), po.ref_scan(rsh, b))), (k, w, n)


def test_crossover_knob_counts_after_the_first_use():
    """MODGPU_ITER_HOST_BELOW set between two calls of one process (mg.knobs -> mgReloadKnobs) is looked up again (the value was
    cached on first use and the knob silently ignored afterwards); 1 << 30 goes back to what the environment says, not to the built-in
    defaults over its head"""
    L = mg.lib()
    was = L.mgIterHostBelow(-1)                      # (looked up: cached from here on)
    with mg.knobs(ITER_HOST_BELOW="777"):
        assert L.mgIterHostBelow(-1) == 777
        assert L.mgIterHostBelow(5) == 777 and L.mgIterHostBelow(-1) == 5
        assert L.mgIterHostBelow(1 << 30) == 5 and L.mgIterHostBelow(-1) == 777
    assert L.mgIterHostBelow(-1) == was
