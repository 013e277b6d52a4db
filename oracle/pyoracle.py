"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, when present, the compiled
reference (oracle/_ref/libmodref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under modimizer_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_TREE = "/root/reference"

U64P = C.POINTER(C.c_uint64)
I32P = C.POINTER(C.c_int32)
U32P = C.POINTER(C.c_uint32)
U8P = C.POINTER(C.c_uint8)
I64P = C.POINTER(C.c_int64)


def build(force=False):
    """make -C oracle (liboracle.so, and _ref when the reference tree is present)."""
    so = os.path.join(HERE, "liboracle.so")
    srcs = [os.path.join(HERE, f) for f in ("orc_seqhash.c", "orc_modset.c", "orc_readset.c", "oracle.h")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    need_ref = os.path.isdir(REF_TREE) and not os.path.exists(os.path.join(HERE, "_ref", "libmodref.so"))
    if stale or need_ref:
        subprocess.check_call(["make", "-C", HERE, "-s"] + (["-B"] if force else []))
    return so


class OrcHasher(C.Structure):
    _fields_ = [("seed", C.c_int), ("k", C.c_int), ("w", C.c_int), ("shift1", C.c_int),
                ("mask", C.c_uint64), ("factor1", C.c_uint64), ("factor2", C.c_uint64)]


class OrcModset(C.Structure):
    _fields_ = [("hasher", OrcHasher), ("tableBits", C.c_int), ("size", C.c_uint32),
                ("tableSize", C.c_uint64), ("tableMask", C.c_uint64),
                ("index", U32P), ("value", U64P), ("depth", C.POINTER(C.c_uint16)), ("info", U8P),
                ("max", C.c_uint32), ("overflow", C.c_int)]


class OrcReference(C.Structure):
    _fields_ = [("ms", C.POINTER(OrcModset)), ("size", C.c_uint32), ("max", C.c_uint32),
                ("index", U32P), ("offset", U32P), ("id", U32P), ("depth", U32P),
                ("rev", U32P), ("loc", U32P), ("nSeq", C.c_int), ("totLen", C.c_int64),
                ("n1", C.c_uint32), ("n2", C.c_uint32), ("nM", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orcHasherInit.argtypes = [C.POINTER(OrcHasher), C.c_int, C.c_int, C.c_int]
        L.orcHasherInit.restype = C.c_int
        L.orcScanRead.argtypes = [C.POINTER(OrcHasher), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.orcScanRead.restype = C.c_int64
        L.orcMinimizerRead.argtypes = L.orcScanRead.argtypes
        L.orcMinimizerRead.restype = C.c_int64
        L.orcModsetCreate.argtypes = [C.POINTER(OrcHasher), C.c_int, C.c_uint32]
        L.orcModsetCreate.restype = C.POINTER(OrcModset)
        L.orcModsetDestroy.argtypes = [C.POINTER(OrcModset)]
        L.orcModsetFind.argtypes = [C.POINTER(OrcModset), C.c_uint64, C.c_int]
        L.orcModsetFind.restype = C.c_uint32
        L.orcModsetPack.argtypes = [C.POINTER(OrcModset)]
        L.orcModsetDepthPrune.argtypes = [C.POINTER(OrcModset), C.c_int, C.c_int]
        L.orcModsetMerge.argtypes = [C.POINTER(OrcModset), C.POINTER(OrcModset)]
        L.orcModsetMerge.restype = C.c_int
        L.orcAddSequence.argtypes = [C.POINTER(OrcModset), C.c_void_p, C.c_int64]
        L.orcAddSequence.restype = C.c_int64
        L.orcDepthHistogram.argtypes = [C.POINTER(OrcModset), C.c_void_p]
        L.orcScanMany.argtypes = [C.POINTER(OrcHasher), C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(OrcModset)]
        L.orcScanMany.restype = C.c_int64
        L.orcSplitmix64.argtypes = [C.c_uint64]
        L.orcSplitmix64.restype = C.c_uint64
        L.orcXorshiftBases.argtypes = [C.c_uint64, C.c_void_p, C.c_int64]
        L.orcXorshiftBases.restype = C.c_uint64
        L.orcSeqString.argtypes = [C.c_uint64, C.c_int]
        L.orcSeqString.restype = C.c_char_p
        L.orcReferenceCreate.argtypes = [C.POINTER(OrcModset), C.c_uint32]
        L.orcReferenceCreate.restype = C.POINTER(OrcReference)
        L.orcReferenceDestroy.argtypes = [C.POINTER(OrcReference)]
        L.orcReferenceAddSequence.argtypes = [C.POINTER(OrcReference), C.c_void_p, C.c_int64, C.c_int]
        L.orcReferenceAddSequence.restype = C.c_int
        L.orcReferenceFinish.argtypes = [C.POINTER(OrcReference), C.c_int]
        _lib = L
    return _lib


_libc = C.CDLL(None)
_libc.fopen.restype = C.c_void_p
_libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
_libc.fclose.argtypes = [C.c_void_p]
_libc.free.argtypes = [C.c_void_p]


def _with_file(path, fn):
    f = _libc.fopen(path.encode(), b"w")
    if not f:
        raise OSError("fopen failed: " + path)
    try:
        fn(C.c_void_p(f))
    finally:
        _libc.fclose(f)


def xorshift_bases(n, state=0x9E3779B97F4A7C15):
    """(bases, next state) of the SURVEY §8(c) generator"""
    out = np.empty(n, np.uint8)
    st = lib().orcXorshiftBases(state, out.ctypes.data, n)
    return out, st


class Hasher:
    """seqhashCreate(k, w, seed) restated (seqhash.c:20-37)."""

    def __init__(self, k, w, seed=17):
        self.c = OrcHasher()
        if lib().orcHasherInit(C.byref(self.c), k, w, seed) != 0:
            raise ValueError("bad seqhash parameters k=%d w=%d" % (k, w))
        self.k, self.w, self.seed = k, w, seed
        self.factor1 = self.c.factor1

    def scan(self, bases):
        """bases: uint8 array of 0..3.  Returns (kmer u64[], pos i32[], isF u8[])."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        cap = max(len(bases) - self.k + 1, 0)
        kmer = np.empty(cap, np.uint64); pos = np.empty(cap, np.int32); isf = np.empty(cap, np.uint8)
        n = lib().orcScanRead(C.byref(self.c), bases.ctypes.data, len(bases),
                              kmer.ctypes.data, pos.ctypes.data, isf.ctypes.data, cap)
        return kmer[:n].copy(), pos[:n].copy(), isf[:n].copy()

    def minimizers(self, bases):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        cap = max(len(bases) - self.k + 1, 0)
        h = np.empty(cap, np.uint64); pos = np.empty(cap, np.int32); isf = np.empty(cap, np.uint8)
        n = lib().orcMinimizerRead(C.byref(self.c), bases.ctypes.data, len(bases),
                                   h.ctypes.data, pos.ctypes.data, isf.ctypes.data, cap)
        return h[:n].copy(), pos[:n].copy(), isf[:n].copy()

    def hash(self, x):
        return ((int(x) * int(self.c.factor1)) & 0xFFFFFFFFFFFFFFFF) >> self.c.shift1


class Modset:
    """modset.c restated; entries 1..max."""

    def __init__(self, hasher, bits, size=0):
        self.h = hasher
        self.p = lib().orcModsetCreate(C.byref(hasher.c), bits, size)
        if not self.p:
            raise ValueError("bad modset parameters")

    def close(self):
        if self.p:
            lib().orcModsetDestroy(self.p); self.p = None

    def __del__(self):
        self.close()

    @property
    def max(self):
        return self.p.contents.max

    def find(self, kmer, is_add=False):
        return lib().orcModsetFind(self.p, int(kmer), int(is_add))

    def add_sequence(self, bases):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        return lib().orcAddSequence(self.p, bases.ctypes.data, len(bases))

    def values(self):
        n = self.max + 1
        return np.ctypeslib.as_array(self.p.contents.value, (n,)).copy()

    def depths(self):
        n = self.max + 1
        return np.ctypeslib.as_array(self.p.contents.depth, (n,)).copy()

    def infos(self):
        n = self.max + 1
        return np.ctypeslib.as_array(self.p.contents.info, (n,)).copy()

    def index_table(self):
        return np.ctypeslib.as_array(self.p.contents.index, (int(self.p.contents.tableSize),))

    def histogram(self):
        h = np.zeros(65536, np.uint64)
        lib().orcDepthHistogram(self.p, h.ctypes.data)
        return h

    def merge(self, other):
        return bool(lib().orcModsetMerge(self.p, other.p))

    def prune(self, lo, hi):
        lib().orcModsetDepthPrune(self.p, lo, hi)

    def pack(self):
        return bool(lib().orcModsetPack(self.p))

    def summary_text(self, tmp):
        L = lib()
        L.orcModsetSummary.argtypes = [C.POINTER(OrcModset), C.c_void_p]
        _with_file(tmp, lambda f: L.orcModsetSummary(self.p, f))
        return open(tmp).read()

    def hist_text(self, tmp):
        L = lib()
        L.orcDepthHistogramPrint.argtypes = [C.POINTER(OrcModset), C.c_void_p]
        _with_file(tmp, lambda f: L.orcDepthHistogramPrint(self.p, f))
        return open(tmp).read()

    def text_dump(self, tmp):
        L = lib()
        L.orcModsetWriteText.argtypes = [C.POINTER(OrcModset), C.c_void_p]
        _with_file(tmp, lambda f: L.orcModsetWriteText(self.p, f))
        return open(tmp).read()

    def set_copy(self, c1, c2, cm):
        """modutils -s (modutils.c:205-214): copy class from depth thresholds"""
        c = self.p.contents
        for u in range(1, c.max + 1):
            d = c.depth[u]
            cls = 0 if d < c1 else (1 if d < c2 else (2 if d < cm else 3))
            c.info[u] = (c.info[u] | 3) if cls == 3 else ((c.info[u] & 0xfc) | cls)

    def write_mod(self, path):
        L = lib()
        L.orcModsetWrite.argtypes = [C.POINTER(OrcModset), C.c_void_p]
        L.orcModsetWrite.restype = C.c_int
        _with_file(path, lambda f: L.orcModsetWrite(self.p, f))


class OrcRead(C.Structure):
    _fields_ = [("len", C.c_int32), ("nHit", C.c_int32), ("nMiss", C.c_int32), ("nCopy", C.c_int32 * 4)]


class OrcReadset(C.Structure):
    _fields_ = [("ms", C.POINTER(OrcModset)), ("nReads", C.c_int), ("capReads", C.c_int),
                ("reads", C.POINTER(OrcRead)), ("hitStart", C.POINTER(C.c_uint64)),
                ("hit", C.POINTER(C.c_uint32)), ("dx", C.POINTER(C.c_uint16)),
                ("totHit", C.c_uint64), ("capHit", C.c_uint64),
                ("invStart", C.POINTER(C.c_uint64)), ("invSpace", C.POINTER(C.c_uint32))]


class Readset:
    """modasm.c read ingest restated (orc_readset.c): readsetFileRead + invBuild over reads in memory."""

    def __init__(self, modset):
        L = lib()
        L.orcReadsetCreate.restype = C.POINTER(OrcReadset)
        L.orcReadsetCreate.argtypes = [C.POINTER(OrcModset)]
        for fn in ("orcReadsetDestroy", "orcReadsetBegin", "orcReadsetFinish"):
            getattr(L, fn).argtypes = [C.POINTER(OrcReadset)]; getattr(L, fn).restype = None
        L.orcReadsetAddRead.argtypes = [C.POINTER(OrcReadset), C.c_void_p, C.c_int64]; L.orcReadsetAddRead.restype = None
        L.orcReadsetStats.argtypes = [C.POINTER(OrcReadset), C.c_void_p]; L.orcReadsetStats.restype = None
        L.orcReadsetWrite.argtypes = [C.POINTER(OrcReadset), C.c_void_p]; L.orcReadsetWrite.restype = C.c_int
        self.ms = modset
        self.p = L.orcReadsetCreate(modset.p)

    def read(self, seqs):
        L = lib()
        L.orcReadsetBegin(self.p)
        for s in seqs:
            s = np.ascontiguousarray(s, dtype=np.uint8)
            L.orcReadsetAddRead(self.p, s.ctypes.data, len(s))
        L.orcReadsetFinish(self.p)

    def arrays(self):
        r = self.p.contents
        n, tot, m = r.nReads, int(r.totHit), self.ms.max
        rd = [r.reads[i] for i in range(1, n + 1)]
        as_np = lambda p, k, dt: np.ctypeslib.as_array(p, (max(k, 1),))[:k].astype(dt).copy()
        return {"len": np.array([x.len for x in rd]), "nHit": np.array([x.nHit for x in rd]),
                "nMiss": np.array([x.nMiss for x in rd]), "nCopy": np.array([list(x.nCopy) for x in rd]).reshape(n, 4),
                "hitStart": as_np(r.hitStart, n + 2, np.uint64)[1:], "hit": as_np(r.hit, tot, np.uint32),
                "dx": as_np(r.dx, tot, np.uint16), "totHit": tot,
                "invStart": as_np(r.invStart, m + 2, np.uint64), "invSpace": as_np(r.invSpace, int(r.invStart[m + 1]), np.uint32)}

    def stats_text(self, tmp):
        _with_file(tmp, lambda f: lib().orcReadsetStats(self.p, f))
        return open(tmp).read()

    def write(self, path):
        _with_file(path, lambda f: lib().orcReadsetWrite(self.p, f))

    def close(self):
        if self.p:
            lib().orcReadsetDestroy(self.p); self.p = None


class Reference:
    """modmap.c Reference restated: build from sequences, then query reads."""

    def __init__(self, modset, size=1 << 26):
        self.ms = modset
        self.p = lib().orcReferenceCreate(modset.p, size)
        self.names = []

    def add_sequence(self, name, bases, is_add=True):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        self.names.append(name)
        if not lib().orcReferenceAddSequence(self.p, bases.ctypes.data, len(bases), int(is_add)):
            raise RuntimeError("reference size overflow")

    def finish(self, is_add=True):
        lib().orcReferenceFinish(self.p, int(is_add))

    def arrays(self):
        r = self.p.contents
        n = r.max
        g = lambda ptr, m: np.ctypeslib.as_array(ptr, (m,)).copy() if m else np.zeros(0, np.uint32)
        mx = self.ms.max + 1
        return dict(index=g(r.index, n), offset=g(r.offset, n), id=g(r.id, n),
                    depth=g(r.depth, mx), rev=g(r.rev, n), loc=g(r.loc, mx),
                    n1=r.n1, n2=r.n2, nM=r.nM)

    def query(self, read_name, bases, tmp):
        """Returns (text of the Q line and M lines, seedIndex[], seedPos[])."""
        L = lib()
        L.orcQueryRead.argtypes = [C.POINTER(OrcReference), C.c_char_p, C.c_void_p, C.c_int64,
                                   C.POINTER(C.c_char_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.orcQueryRead.restype = C.c_int64
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        cap = max(len(bases), 1)
        six = np.zeros(cap, np.uint32); spos = np.zeros(cap, np.uint32)
        names = (C.c_char_p * max(len(self.names), 1))(*[n.encode() for n in self.names])
        out = {}

        def run(f):
            out["n"] = L.orcQueryRead(self.p, read_name.encode(), bases.ctypes.data, len(bases),
                                      names, f, six.ctypes.data, spos.ctypes.data, cap)
        _with_file(tmp, run)
        n = out["n"]
        return open(tmp).read(), six[:n].copy(), spos[:n].copy()

    def close(self):
        if self.p:
            lib().orcReferenceDestroy(self.p); self.p = None


# ----------------------------------------------------------------------------------------------
# The compiled reference itself (oracle/_ref/libmodref.so), when present.

class RefSeqhash(C.Structure):       # seqhash.h:15-23
    _fields_ = [("seed", C.c_int), ("k", C.c_int), ("w", C.c_int), ("mask", C.c_uint64),
                ("shift1", C.c_int), ("shift2", C.c_int), ("factor1", C.c_uint64),
                ("factor2", C.c_uint64), ("patternRC", C.c_uint64 * 4)]


class RefIterator(C.Structure):      # seqhash.h:25-34
    _fields_ = [("sh", C.POINTER(RefSeqhash)), ("s", C.c_void_p), ("sEnd", C.c_void_p),
                ("h", C.c_uint64), ("hRC", C.c_uint64), ("hashBuf", C.c_void_p), ("fBuf", C.c_void_p),
                ("base", C.c_int), ("iStart", C.c_int), ("iMin", C.c_int), ("isDone", C.c_bool)]


class RefModset(C.Structure):        # modset.h:17-28
    _fields_ = [("hasher", C.POINTER(RefSeqhash)), ("tableBits", C.c_int), ("size", C.c_uint32),
                ("tableSize", C.c_uint64), ("tableMask", C.c_uint64), ("index", U32P),
                ("value", U64P), ("depth", C.POINTER(C.c_uint16)), ("info", U8P), ("max", C.c_uint32)]


_ref = None


def ref_path():
    return os.path.join(HERE, "_ref", "libmodref.so")


def have_ref():
    if os.path.isdir(REF_TREE):
        build()
    return os.path.exists(ref_path())


def ref():
    global _ref
    if _ref is None:
        R = C.CDLL(ref_path())
        R.seqhashCreate.argtypes = [C.c_int, C.c_int, C.c_int]
        R.seqhashCreate.restype = C.POINTER(RefSeqhash)
        R.modRCiterator.argtypes = [C.POINTER(RefSeqhash), C.c_void_p, C.c_int]
        R.modRCiterator.restype = C.POINTER(RefIterator)
        R.modRCnext.argtypes = [C.POINTER(RefIterator), U64P, C.POINTER(C.c_int), C.POINTER(C.c_bool)]
        R.modRCnext.restype = C.c_bool
        R.minimizerRCiterator.argtypes = R.modRCiterator.argtypes
        R.minimizerRCiterator.restype = C.POINTER(RefIterator)
        R.minimizerRCnext.argtypes = R.modRCnext.argtypes
        R.minimizerRCnext.restype = C.c_bool
        R.modsetCreate.argtypes = [C.POINTER(RefSeqhash), C.c_int, C.c_uint32]
        R.modsetCreate.restype = C.POINTER(RefModset)
        R.modsetIndexFind.argtypes = [C.POINTER(RefModset), C.c_uint64, C.c_int]
        R.modsetIndexFind.restype = C.c_uint32
        R.modsetDestroy.argtypes = [C.POINTER(RefModset)]
        R.modsetMerge.argtypes = [C.POINTER(RefModset), C.POINTER(RefModset)]
        R.modsetMerge.restype = C.c_bool
        R.modsetDepthPrune.argtypes = [C.POINTER(RefModset), C.c_int, C.c_int]
        R.modsetPack.argtypes = [C.POINTER(RefModset)]
        R.modsetPack.restype = C.c_bool
        R.modsetSummary.argtypes = [C.POINTER(RefModset), C.c_void_p]
        R.modsetWrite.argtypes = [C.POINTER(RefModset), C.c_void_p]
        R.seqString.argtypes = [C.c_uint64, C.c_int]
        R.seqString.restype = C.c_char_p
        _ref = R
    return _ref


def _ref_iter_free(it):
    # seqhash.h:54-55 (static in the header, so not exported)
    _libc.free(it.contents.hashBuf); _libc.free(it.contents.fBuf)
    _libc.free(C.cast(it, C.c_void_p))


def ref_scan(sh, bases, minimizer=False):
    """Run the reference's own iterator over one read. Returns (u64[], pos[], isF[])."""
    R = ref()
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    mk, nx = (R.minimizerRCiterator, R.minimizerRCnext) if minimizer else (R.modRCiterator, R.modRCnext)
    it = mk(sh, bases.ctypes.data, len(bases))
    u = C.c_uint64(); p = C.c_int(); f = C.c_bool()
    ks, ps, fs = [], [], []
    while nx(it, C.byref(u), C.byref(p), C.byref(f)):
        ks.append(u.value); ps.append(p.value); fs.append(int(f.value))
    _ref_iter_free(it)
    return np.array(ks, np.uint64), np.array(ps, np.int32), np.array(fs, np.uint8)


def ref_add_sequence(ms, bases):
    """modutils.c:19-31 driven through the reference's own exported functions."""
    R = ref()
    kmers, _, _ = ref_scan(ms.contents.hasher, bases)
    for km in kmers:
        ix = R.modsetIndexFind(ms, int(km), 1)
        d = (int(ms.contents.depth[ix]) + 1) & 0xffff
        ms.contents.depth[ix] = d if d else 0xffff
    return len(kmers)


def ref_text(fn, ms, tmp):
    _with_file(tmp, lambda f: fn(ms, f))
    return open(tmp, "rb").read()
