"""Layer 1 of libmodgpu (the reference's seqhash.h/modset.h API, host side) against the golden
vectors and the oracle.  No GPU is needed: these functions work on the host arrays."""
import ctypes as C
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

import modimizer_amd as mg
from modimizer_amd import fasta
from oracle import pyoracle as po
import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_seqhash_create_and_strings():
    L = mg.lib()
    sh = mg.seqhashCreate(21, 64, 17)
    c = sh.contents
    assert (c.seed, c.k, c.w, c.shift1, c.shift2) == (17, 21, 64, 22, 42)
    assert c.mask == 0x3ffffffffff and c.factor1 == 0x49308bb9003cb3ad and c.factor2 == 0x0fb4e87f75655103
    assert [c.patternRC[b] for b in range(4)] == [(3 - b) << 40 for b in range(4)]
    assert L.seqString(0x142be04b2e6, 21) == po.lib().orcSeqString(0x142be04b2e6, 21)
    assert L.seqString(0b00011011, 4) == b"acgt"
    for ci, (k, w, seed) in enumerate(util.scan_configs()):
        s2 = mg.seqhashCreate(k, w, seed)
        v = util.scan_vectors()["c%d_factor1" % ci]
        assert s2.contents.factor1 == int(v[0]) and s2.contents.factor2 == int(v[1])
        L.mgSeqhashDestroy(s2)


def test_seqhash_write_read_roundtrip(tmp_path):
    L = mg.lib()
    sh = mg.seqhashCreate(19, 31, 5)
    p = str(tmp_path / "sh.bin")
    with mg.CFile(p, "w") as f:
        L.seqhashWrite(sh, f)
    raw = open(p, "rb").read()
    assert raw[:8] == b"SQHSHv2\0" and len(raw) == 88
    with mg.CFile(p, "r") as f:
        s2 = L.seqhashRead(f)
    assert bytes(C.string_at(C.addressof(s2.contents), 80)) == raw[8:]
    rep = str(tmp_path / "rep.txt")
    with mg.CFile(rep, "w") as f:
        L.seqhashReport(s2, f)
    assert open(rep).read() == "SH k 19  w/m 31  s 5\n"


def test_bad_parameters_raise():
    with pytest.raises(ValueError):
        mg.seqhashCreate(32, 64, 17)
    with pytest.raises(ValueError):
        mg.seqhashCreate(21, 0, 17)
    sh = mg.seqhashCreate(21, 64, 17)
    with pytest.raises(ValueError):
        mg.modsetCreate(sh, 19)
    with pytest.raises(ValueError):
        mg.modsetCreate(sh, 20, 1 << 18)


@pytest.mark.parametrize("snippet,msg", [
    ("L.seqhashCreate(0, 5, 1)", "seqhash k 0 must be between 1 and 32"),
    ("L.seqhashCreate(5, 0, 1)", "seqhash w 0 must be positive"),
    ("L.modsetCreate(L.seqhashCreate(5, 3, 1), 35, 0)", "table bits 35 must be between 20 and 34"),
    ("L.modsetCreate(L.seqhashCreate(5, 3, 1), 20, 262144)", "Modset size 262144 is too big for 20 bits"),
    ("ms = L.modsetCreate(L.seqhashCreate(5, 1, 1), 20, 3)\nfor i in range(5): L.modsetIndexFind(ms, i, 1)",
     "hashTableSize 3 is too small for 3"),
])
def test_die_behaviour_matches_reference(snippet, msg):
    """errors print 'FATAL ERROR: ...' and exit(-1) (utils.c:19-30)"""
    code = "import modimizer_amd as mg\nL = mg.lib()\n" + snippet + "\n"
    env = dict(os.environ, MODGPU_NO_TORCH="1", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert r.returncode == 255, (r.returncode, r.stderr)
    assert "FATAL ERROR: " + msg in r.stderr


def test_pack_host_layout():
    b = np.array([0, 1, 2, 3] * 5 + [3, 3, 0], np.uint8)      # 23 bases
    w = mg.pack_host(b)
    assert len(w) == 2 + 8
    assert w[0] == int("00011011" * 4, 2)
    assert w[1] == int("00011011" + "111100" + "0" * 18, 2)
    assert not w[2:].any()
    # bytes above 3 only contribute their low two bits (N is patched to 0 by the callers, modmap.c:97)
    assert mg.pack_host(np.array([4, 7, 255], np.uint8))[0] == int("001111" + "0" * 26, 2)


def _host_add(L, ms, kmers):
    """modutils.c:24-27 on the host arrays"""
    for km in kmers:
        ix = L.modsetIndexFind(ms, int(km), 1)
        d = (int(ms.contents.depth[ix]) + 1) & 0xffff
        ms.contents.depth[ix] = d if d else 0xffff


def _snapshot(ms, B):
    v, d, i = mg.modset_arrays(ms)
    idx = np.ctypeslib.as_array(ms.contents.index, (1 << B,))
    nz = np.nonzero(idx)[0]
    return v, d, i, nz.astype(np.uint32), idx[nz].copy()


def _check(L, ms, g, tag, B, tmp):
    v, d, i, pos, val = _snapshot(ms, B)
    assert np.array_equal(v[1:], g[tag + "_value"][1:])
    assert np.array_equal(d, g[tag + "_depth"]) and np.array_equal(i, g[tag + "_info"])
    assert np.array_equal(pos, g[tag + "_index_pos"]) and np.array_equal(val, g[tag + "_index_val"])
    with mg.CFile(tmp, "w") as f:
        L.modsetSummary(ms, f)
    assert open(tmp, "rb").read() == g[tag + "_summary"].tobytes()


def test_modset_host_ops_vs_golden(golden_dir, tmp_path):
    """scalar modsetIndexFind + merge + prune + pack + write, all on host arrays (modset.c)"""
    L = mg.lib()
    g = np.load(os.path.join(util.GOLDEN, "modset_ops.npz"))
    k, w, seed, B = (int(x) for x in g["params"])
    sh = mg.seqhashCreate(k, w, seed)
    oh = po.Hasher(k, w, seed)      # checker: supplies the modimizer streams (the GPU scan has its own tests)
    a, b = mg.modsetCreate(sh, B), mg.modsetCreate(sh, B)
    for ms, fn in ((a, "reads.fa"), (b, "reads2.fa")):
        names, bases, offs = fasta.read_fasta(os.path.join(golden_dir, fn))
        for r in range(len(names)):
            _host_add(L, ms, oh.scan(bases[offs[r]:offs[r + 1]])[0])
    for i in range(1, b.contents.max + 1):
        b.contents.info[i] = (i % 4) | ((i % 3 == 0) * 8)
    for i in range(1, a.contents.max + 1):
        a.contents.info[i] = ((i // 2) % 4) | ((i % 5 == 0) * 16)
    tmp = str(tmp_path / "s.txt")
    _check(L, a, g, "a", B, tmp)
    _check(L, b, g, "b", B, tmp)
    # .mod file (modset.c:79-88): byte-identical to the reference's once value[0] is zeroed
    a.contents.value[0] = 0
    mod = str(tmp_path / "a.mod")
    with mg.CFile(mod, "w") as f:
        L.modsetWrite(a, f)
    data = open(mod, "rb").read()
    assert data[:104] == g["a_mod_header"].tobytes()
    assert hashlib.sha256(data).digest() == g["a_mod_sha256"].tobytes()
    with mg.CFile(mod, "r") as f:
        a2 = L.modsetRead(f)
    assert a2.contents.max == a.contents.max and a2.contents.size == a.contents.max + 1
    for km in g["a_value"][1:50]:
        assert L.modsetIndexFind(a2, int(km), 0) == L.modsetIndexFind(a, int(km), 0) != 0
    assert L.modsetIndexFind(a2, 123456789, 0) == 0
    # merge (modset.c:106-128), prune (:64-77), pack (:36-43)
    assert L.modsetMerge(a, b)
    _check(L, a, g, "merged", B, tmp)
    L.modsetDepthPrune(a, 2, 30)
    _check(L, a, g, "pruned", B, tmp)
    assert L.modsetPack(a) and a.contents.size == int(g["packed_size"][0])
    assert not L.modsetPack(a)
    # incompatible hashers do not merge (modset.c:111)
    other = mg.modsetCreate(mg.seqhashCreate(k, w + 1, seed), B)
    assert not L.modsetMerge(a, other)
    for ms in (a, b, a2, other):
        L.modsetDestroy(ms)


def test_fasta_reader_matches_seqio_conventions(golden_dir):
    names, bases, offs = fasta.read_fasta(os.path.join(golden_dir, "reads.fa"))
    assert names[-3:] == ["short", "exact21", "withN"] and offs[-1] == len(bases) and bases.max() <= 3
    assert offs[-1] - offs[-2] == 109          # Ns kept (as base 0), nothing dropped
    assert list(bases[offs[-2]:offs[-2] + 12]) == [0, 1, 2, 3, 0, 0, 0, 0, 0, 1, 2, 3]


def test_pack_host_matches_the_word_layout():
    """mgPackHost (AVX2 and portable loops): base i in bits [30-2(i%16), 32-2(i%16)) of word i/16, only the low two bits of
    a byte count (FASTQ keeps other bytes as 0xFE), MG_PACK_PAD zero words follow"""
    code = ("import sys, numpy as np\nsys.path.insert(0, 'ROOTDIR')\nimport modimizer_amd as mg\nrng = np.random.default_rng(4)\n"
            "for n in list(range(0, 70)) + [127, 128, 129, 255, 256, 257, 1000, 4097, 65536 + 5]:\n"
            "    b = rng.integers(0, 256, n).astype(np.uint8)\n    w = mg.pack_host(b)\n"
            "    ref = np.zeros(len(w), np.uint32)\n"
            "    for i in range(n): ref[i // 16] |= np.uint32((int(b[i]) & 3) << (30 - 2 * (i % 16)))\n"
            "    assert np.array_equal(w, ref), n\n    assert len(w) == (n + 15) // 16 + 8\nprint('ok')").replace("ROOTDIR", ROOT)
    for no in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                           env=dict(os.environ, MODGPU_NO_TORCH="1", MODGPU_NO_AVX2=no))
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-1500:]


def test_format_f2_equals_printf():
    """the parallel formatter of the Q / M lines (mg_callers.c fmtF2) against printf ("%.2f") itself: every ratio a / b the lines
    can hold for small a, b (ties like 1/8 = 0.125 -> 0.12, 3/8 -> 0.38, and 0.005-neighbours such as 1/200), random ratios,
    values beyond 1, and the non-finite cases (0/0 -> -nan or nan as glibc prints it, x/0 -> inf)"""
    import ctypes as C
    L = mg.lib()
    buf = C.create_string_buffer(64)

    def f2(x):
        n = L.mgFormatF2(buf, x)
        return buf.raw[:n].decode()
    bad = []
    for b in range(1, 401):
        for a in range(0, b + 1):
            x = a / b
            if f2(x) != "%.2f" % x:
                bad.append((a, b))
    assert not bad, bad[:10]
    rng = np.random.default_rng(3)
    for a, b in zip(rng.integers(0, 10**6, 200000), rng.integers(1, 10**6, 200000)):
        x = int(a) / int(b)
        assert f2(x) == "%.2f" % x, (a, b)
    for x in (0.0, 0.125, 0.375, 0.625, 0.875, 2.675, 1e-9, 0.004999999999999999, 0.005, 0.015, 0.025, 123456.785, 1e14 + 0.125, 2.5e15,
              float("inf"), 1e20):
        assert f2(x) == "%.2f" % x, x
    # what glibc's printf makes of the rest (the library calls snprintf itself there): the sign of a nan is printed
    for x, want in ((float("nan"), "nan"), (-float("nan"), "-nan"), (-0.0, "-0.00"), (-1.5, "-1.50"), (-float("inf"), "-inf")):
        assert f2(x) == want, x
