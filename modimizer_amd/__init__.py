"""modimizer_amd — MI355X-native seqhash + modset hot path (libmodgpu.so) behind the reference's API.

This module is a thin ctypes binding over the C ABI declared in include/modgpu.h; all work happens
in the HIP library.  There is no CPU fallback: if the library or a HIP device is missing the batch
entry points raise.
"""
import contextlib
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MODGPU_LIB") or os.path.join(_HERE, "libmodgpu.so")   # MODGPU_LIB: a variant build (tools/ablate_*.sh)
CSRC = os.path.join(_HERE, "csrc")

U64P = C.POINTER(C.c_uint64)
U32P = C.POINTER(C.c_uint32)

MG_POS_MASK = 0x7FFFFFFF
MG_FWD_BIT = 0x80000000


class Seqhash(C.Structure):            # include/modgpu.h (reference seqhash.h:15-23)
    _fields_ = [("seed", C.c_int), ("k", C.c_int), ("w", C.c_int), ("mask", C.c_uint64),
                ("shift1", C.c_int), ("shift2", C.c_int), ("factor1", C.c_uint64),
                ("factor2", C.c_uint64), ("patternRC", C.c_uint64 * 4)]


class SeqhashRCiterator(C.Structure):  # reference seqhash.h:25-34
    _fields_ = [("sh", C.POINTER(Seqhash)), ("s", C.c_void_p), ("sEnd", C.c_void_p),
                ("h", C.c_uint64), ("hRC", C.c_uint64), ("hashBuf", C.c_void_p), ("fBuf", C.c_void_p),
                ("base", C.c_int), ("iStart", C.c_int), ("iMin", C.c_int), ("isDone", C.c_bool)]


class Modset(C.Structure):             # reference modset.h:17-28
    _fields_ = [("hasher", C.POINTER(Seqhash)), ("tableBits", C.c_int), ("size", C.c_uint32),
                ("tableSize", C.c_uint64), ("tableMask", C.c_uint64), ("index", U32P),
                ("value", U64P), ("depth", C.POINTER(C.c_uint16)), ("info", C.POINTER(C.c_uint8)),
                ("max", C.c_uint32)]


class MgReference(C.Structure):       # include/modgpu.h (modmap.c:35-47)
    _fields_ = [("ms", C.POINTER(Modset)), ("size", C.c_uint32), ("max", C.c_uint32),
                ("index", U32P), ("offset", U32P), ("id", U32P), ("depth", U32P), ("rev", U32P), ("loc", U32P),
                ("nSeq", C.c_int), ("names", C.POINTER(C.c_char_p)), ("len", U32P)]


class MgReadset(C.Structure):         # include/modgpu.h (modasm.c:30-57,79-86)
    _fields_ = [("ms", C.POINTER(Modset)), ("nReads", C.c_int), ("capReads", C.c_int),
                ("len", C.POINTER(C.c_int)), ("nHit", C.POINTER(C.c_int)), ("nMiss", C.POINTER(C.c_int)),
                ("nCopy", C.POINTER(C.c_int * 4)), ("hitStart", U64P), ("hit", U32P), ("dx", C.POINTER(C.c_uint16)),
                ("totHit", C.c_uint64), ("capHit", C.c_uint64), ("invStart", U64P), ("invSpace", U32P)]


class MgSeqBatch(C.Structure):        # include/modgpu.h
    _fields_ = [("bases", C.POINTER(C.c_int8)), ("offsets", C.POINTER(C.c_int64)), ("names", C.POINTER(C.c_char_p)),
                ("nSeq", C.c_int), ("total", C.c_int64), ("isFastq", C.c_int), ("basesCap", C.c_int64)]


# every symbol include/modgpu.h declares (tests check the library exports all of them)
EXPORTS = [
    "seqhashCreate", "seqhashWrite", "seqhashRead", "seqhashReport", "modRCiterator", "modRCnext",
    "minimizerRCiterator", "minimizerRCnext", "seqString", "mgSeqhashDestroy", "mgSeqhashRCiteratorDestroy",
    "modsetCreate", "modsetDestroy", "modsetWrite", "modsetRead", "modsetIndexFind", "modsetSummary",
    "modsetPack", "modsetDepthPrune", "modsetMerge",
    "mgLastError", "mgDeviceCount", "mgSetDevice", "mgVersion", "mgSourceHash", "mgDeviceAlloc", "mgDeviceFree",
    "mgMemcpyH2D", "mgMemcpyD2H", "mgMemsetD", "mgStreamSynchronize",
    "mgPackedWords", "mgPackHost", "mgPackDevice", "mgUnpackDevice", "mgUploadPack",
    "mgScanWorkBytes", "seqhashScanBatchDevice", "seqhashScanBatch", "seqhashMinimizerBatchDevice", "seqhashMinimizerBatch",
    "modsetAddBatchDevice", "modsetFindBatchDevice", "modsetSyncToHost", "mgXferThreadCount", "mgCopyD2HBig", "mgCopyH2DBig", "mgModsetDeviceRelease",
    "mgModsetHostChanged", "modsetDepthHistogramDevice", "mgAddReadsDevice", "mgQueryReadsDevice", "mgQueryReadsDeviceAsync", "mgQueryReadsDeviceWait",
    "mgAddSequenceBatch", "mgDepthHistogram", "mgSynthGenome", "mgSynthReads",
    "mgInsertReadsDevice", "mgAddSequences", "mgModsetWriteText", "mgReferenceCreate", "mgReferenceDestroy",
    "mgReferenceRead", "mgQueryProcess", "mgReferenceWrite", "mgGzipOpenWrite", "mgFzOpen", "mgGzipOpenRead", "mgReferenceLoad",
    "mgCommInitAll", "mgCommGetUniqueId", "mgCommInitRank", "mgCommRank", "mgCommSize", "mgCommDestroy", "mgHistogramAllReduce", "mgDepthAllReduce", "mgModsetMergeRankOrder",
    "mgReadsetCreate", "mgReadsetDestroy", "mgReadsetRead", "mgReadsetFileRead", "mgReadsetStats", "mgReadsetWrite", "mgReadsetLoad",
    "mgSeqOpen", "mgSeqNextBatch", "mgSeqBatchFree", "mgSeqClose", "mgSeqReleaseBuffers", "mgReleaseBuffers", "mgTextParseFileDevice", "mgAddSequenceFile", "mgReferenceFastaRead", "mgQueryFile",
    "mgIterScanHost", "mgIterHostBelow", "mgReloadKnobs", "mgFormatF2", "mgModsetMergeArrays", "mgModsetMergeDeviceArrays", "mgModsetClear", "mgModsetDeviceSlots", "mgSetVerbose", "mgProfileEnable", "mgProfileOnly", "mgProfileReset", "mgProfileKernels", "mgProfileGet",
]


@contextlib.contextmanager
def knobs(**kv):
    """MODGPU_<NAME>=value (None: unset) for the duration of the block.  The library reads its environment knobs once
    (csrc/mg_knobs.c); mgReloadKnobs makes it read them again, on the way in and on the way out."""
    old = {}
    for k, v in kv.items():
        name = "MODGPU_" + k
        old[name] = os.environ.get(name)
        if v is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = str(v)
    lib().mgReloadKnobs()
    try:
        yield
    finally:
        for name, v in old.items():
            if v is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = v
        lib().mgReloadKnobs()


def source_hash():
    """the hash csrc/Makefile bakes into the library (mg_version.c): sha256 over the `sha256sum` listing of every source, 16 hex digits"""
    import hashlib
    names = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".c", ".h"))) + ["Makefile", "../../include/modgpu.h", "../../include/modgpu_compat.h"]
    inc = os.path.join(_HERE, "..", "include")
    path = lambda n: os.path.join(inc, os.path.basename(n)) if n.startswith("../") else os.path.join(CSRC, n)
    listing = "".join("%s  %s\n" % (hashlib.sha256(open(path(n), "rb").read()).hexdigest(), n) for n in names)
    return hashlib.sha256(listing.encode()).hexdigest()[:16]


def binary_hash(path=None):
    """the source hash a built libmodgpu.so carries, read out of the file (no dlopen); None if there is no such file or marker"""
    import re
    try:
        m = re.search(rb"MODGPU_SRC_HASH=([0-9a-f]{16})", open(path or LIB_PATH, "rb").read())
    except OSError:
        return None
    return m.group(1).decode() if m else None


def build(force=False):
    """Compile libmodgpu.so for gfx950 in-tree (hipcc cross-compiles without a GPU) whenever the binary is not the one these sources
    make: the hash baked into it differs from the tree's (a stale .so that rode along cannot be used), or it does not exist."""
    import fcntl
    if os.environ.get("MODGPU_LIB"):                    # a variant build under tools/: the caller's business
        return LIB_PATH
    def make_if_stale():
        if force or binary_hash() != source_hash():
            subprocess.check_call(["make", "-C", CSRC, "-j8", "-s"] + (["-B"] if force else []))
            if binary_hash() != source_hash():
                raise RuntimeError("libmodgpu.so does not carry the hash of its sources after make: %s != %s" % (binary_hash(), source_hash()))
    try:
        lk = open(os.path.join(CSRC, ".build.lock"), "w")     # tests start many processes: one make at a time
    except OSError:                                            # (a read-only tree: nothing to serialise, and nothing to build unless the binary is stale)
        make_if_stale()
        return LIB_PATH
    with lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        make_if_stale()
    return LIB_PATH


_lib = None


def lib():
    """Load libmodgpu.so, building it first if it is not the build of the sources in the tree. Raises if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if os.environ.get("MODGPU_NO_TORCH", "0") != "1":
        # torch bundles its own libamdhip64.so.7; load it first so both sides share ONE HIP runtime
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    build()
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    if not os.environ.get("MODGPU_LIB"):
        L.mgSourceHash.restype = C.c_char_p
        if L.mgSourceHash().decode() != source_hash():
            raise RuntimeError("the loaded libmodgpu.so was not built from the sources in this tree")
    vp, u64, u32, i32, i64 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_int64
    SH, MS, IT = C.POINTER(Seqhash), C.POINTER(Modset), C.POINTER(SeqhashRCiterator)

    def sig(name, res, *args):
        if os.environ.get("MODGPU_LIB") and not hasattr(L, name):      # a variant build of another round (A/B runs): what it lacks stays unbound
            return
        f = getattr(L, name); f.restype = res; f.argtypes = list(args)
    sig("seqhashCreate", SH, i32, i32, i32)
    sig("seqhashWrite", None, SH, vp); sig("seqhashRead", SH, vp); sig("seqhashReport", None, SH, vp)
    sig("modRCiterator", IT, SH, vp, i32)
    sig("modRCnext", C.c_bool, IT, U64P, C.POINTER(i32), C.POINTER(C.c_bool))
    sig("minimizerRCiterator", IT, SH, vp, i32)
    sig("mgIterScanHost", vp, SH, vp, i32); sig("mgIterHostBelow", i32, i32); sig("mgReloadKnobs", None)
    sig("mgFormatF2", i32, C.c_char_p, C.c_double)
    sig("minimizerRCnext", C.c_bool, IT, U64P, C.POINTER(i32), C.POINTER(C.c_bool))
    sig("seqString", C.c_char_p, u64, i32)
    sig("mgSeqhashDestroy", None, SH); sig("mgSeqhashRCiteratorDestroy", None, IT)
    sig("modsetCreate", MS, SH, i32, u32); sig("modsetDestroy", None, MS)
    sig("modsetWrite", None, MS, vp); sig("modsetRead", MS, vp)
    sig("modsetIndexFind", u32, MS, u64, i32); sig("modsetSummary", None, MS, vp)
    sig("modsetPack", C.c_bool, MS); sig("modsetDepthPrune", None, MS, i32, i32)
    sig("modsetMerge", C.c_bool, MS, MS)
    sig("mgLastError", C.c_char_p); sig("mgDeviceCount", i32); sig("mgSetDevice", i32, i32)
    sig("mgVersion", C.c_char_p)
    sig("mgSourceHash", C.c_char_p)
    sig("mgDeviceAlloc", i32, C.POINTER(vp), C.c_size_t); sig("mgDeviceFree", i32, vp)
    sig("mgMemcpyH2D", i32, vp, vp, C.c_size_t, vp); sig("mgMemcpyD2H", i32, vp, vp, C.c_size_t, vp)
    sig("mgMemsetD", i32, vp, i32, C.c_size_t, vp); sig("mgStreamSynchronize", i32, vp)
    sig("mgPackedWords", C.c_size_t, u64); sig("mgPackHost", None, vp, u64, vp)
    sig("mgUploadPack", i32, vp, u64, vp, vp); sig("mgPackDevice", i32, vp, u64, vp, vp); sig("mgUnpackDevice", i32, vp, u64, vp, vp)
    sig("mgScanWorkBytes", C.c_size_t, u64, u32, u64)
    sig("seqhashScanBatchDevice", i32, SH, vp, u64, vp, u32, vp, vp, vp, u64, vp, vp, vp)
    sig("seqhashScanBatch", i64, SH, vp, vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp))
    sig("seqhashMinimizerBatchDevice", i32, SH, vp, u64, vp, u32, vp, vp, vp, u64, U64P, vp)
    sig("seqhashMinimizerBatch", i64, SH, vp, vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp))
    sig("modsetAddBatchDevice", i32, MS, vp, u64, vp, i32, vp)
    sig("modsetFindBatchDevice", i32, MS, vp, u64, vp, vp)
    sig("modsetSyncToHost", i32, MS, i32); sig("mgXferThreadCount", i32); sig("mgCopyD2HBig", i32, vp, vp, C.c_size_t); sig("mgCopyH2DBig", i32, vp, vp, C.c_size_t); sig("mgModsetDeviceRelease", i32, MS)
    sig("mgModsetHostChanged", None, MS)
    sig("modsetDepthHistogramDevice", i32, MS, vp, vp)
    sig("mgAddReadsDevice", i32, MS, vp, u64, vp, u32, U64P, vp)
    sig("mgQueryReadsDevice", i32, MS, vp, u64, vp, u32, vp, vp, vp, u64, U64P, vp)
    sig("mgQueryReadsDeviceAsync", i32, MS, vp, u64, vp, u32, vp, vp, vp, u64, C.POINTER(vp), vp); sig("mgQueryReadsDeviceWait", i32, vp, U64P, vp)
    sig("mgAddSequenceBatch", i64, MS, vp, vp, i32); sig("mgDepthHistogram", None, MS, vp)
    sig("mgSynthGenome", i32, vp, u64, u64, vp)
    sig("mgSynthReads", i32, vp, u64, vp, vp, vp, u32, u64, C.c_double, u64, vp, vp)
    sig("mgInsertReadsDevice", i32, MS, vp, u64, vp, u32, vp, vp, vp, u64, U64P, vp)
    sig("mgAddSequences", i32, MS, vp, vp, i32, vp); sig("mgModsetWriteText", None, MS, vp)
    sig("mgReferenceCreate", vp, MS, u32); sig("mgReferenceDestroy", None, vp)
    sig("mgReferenceRead", i32, vp, vp, vp, i32, C.POINTER(C.c_char_p), C.c_bool, vp)
    sig("mgQueryProcess", i32, vp, vp, vp, i32, C.POINTER(C.c_char_p), vp)
    RS = C.POINTER(MgReadset)
    sig("mgReadsetCreate", RS, MS); sig("mgReadsetDestroy", None, RS)
    sig("mgReadsetRead", i32, RS, vp, vp, i32); sig("mgReadsetFileRead", i32, RS, C.c_char_p)
    sig("mgReadsetStats", None, RS, vp); sig("mgReadsetWrite", None, RS, C.c_char_p); sig("mgReadsetLoad", RS, C.c_char_p)
    sig("mgSeqOpen", vp, C.c_char_p); sig("mgSeqNextBatch", i32, vp, C.c_int64, C.POINTER(MgSeqBatch))
    sig("mgSeqBatchFree", None, C.POINTER(MgSeqBatch)); sig("mgSeqClose", None, vp); sig("mgSeqReleaseBuffers", None); sig("mgReleaseBuffers", None)
    sig("mgTextParseFileDevice", i32, C.c_char_p, C.POINTER(vp), C.POINTER(vp), C.POINTER(i64))
    sig("mgAddSequenceFile", i32, MS, C.c_char_p, vp); sig("mgReferenceFastaRead", i32, vp, C.c_char_p, C.c_bool, vp)
    sig("mgQueryFile", i32, vp, C.c_char_p, vp)
    sig("mgReferenceWrite", None, vp, C.c_char_p); sig("mgGzipOpenWrite", vp, C.c_char_p); sig("mgGzipOpenRead", vp, C.c_char_p); sig("mgFzOpen", vp, C.c_char_p, C.c_char_p);
    sig("mgCommInitAll", i32, C.POINTER(vp), i32, C.POINTER(i32)); sig("mgCommGetUniqueId", i32, vp); sig("mgCommInitRank", i32, C.POINTER(vp), i32, i32, vp, i32)
    sig("mgCommRank", i32, vp); sig("mgCommSize", i32, vp); sig("mgCommDestroy", None, vp)
    sig("mgHistogramAllReduce", i32, MS, vp, vp); sig("mgDepthAllReduce", i32, MS, vp); sig("mgModsetMergeRankOrder", i32, MS, vp, i32)
    sig("mgReferenceLoad", C.POINTER(MgReference), C.c_char_p)
    sig("mgModsetMergeArrays", C.c_bool, MS, vp, vp, vp, u32)
    sig("mgModsetMergeDeviceArrays", C.c_bool, MS, vp, vp, vp, u32)
    sig("mgModsetClear", i32, MS, vp); sig("mgModsetDeviceSlots", u64, MS); sig("mgSetVerbose", None, i32)
    sig("mgProfileEnable", None, i32); sig("mgProfileOnly", None, i32); sig("mgProfileReset", None); sig("mgProfileKernels", i32)
    sig("mgProfileGet", i32, i32, C.POINTER(C.c_char_p), C.POINTER(C.c_double), U64P)
    _lib = L
    return L


class ModgpuError(RuntimeError):
    pass


def check(status):
    if status != 0:
        raise ModgpuError("libmodgpu status %d: %s" % (status, lib().mgLastError().decode()))


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]
_libc.fopen.restype = C.c_void_p
_libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
_libc.fclose.argtypes = [C.c_void_p]


class CFile:
    """FILE* for the reference-style functions that print to a FILE (modsetSummary etc.)."""

    def __init__(self, path, mode="w"):
        self.f = _libc.fopen(path.encode(), mode.encode())
        if not self.f:
            raise OSError("cannot open " + path)

    def __enter__(self):
        return C.c_void_p(self.f)

    def __exit__(self, *a):
        _libc.fclose(self.f)


# ------------------------------------------------------------------------------------------------
# Host-side mirror of the reference interface (same names / argument meaning).

def seqhashCreate(k, w, seed=17):
    """reference seqhash.c:20-37.  Invalid k/w make the C library die() exactly like the reference;
    here they raise ValueError first so a Python caller survives."""
    if k < 1 or k >= 32:
        raise ValueError("seqhash k %d must be between 1 and 32" % k)
    if w < 1:
        raise ValueError("seqhash w %d must be positive" % w)
    return lib().seqhashCreate(k, w, seed)


def modsetCreate(sh, bits, size=0):
    """reference modset.c:15-31"""
    if bits < 20 or bits > 34:
        raise ValueError("table bits %d must be between 20 and 34" % bits)
    if size >= (1 << bits) >> 2:
        raise ValueError("Modset size %u is too big for %d bits" % (size, bits))
    return lib().modsetCreate(sh, bits, size)


def iterate(sh, bases, minimizer=False):
    """Drive modRCiterator/modRCnext (or the minimizer pair) over one read, as the reference's
    callers do (modutils.c:22-29).  bases: uint8 array of 0..3."""
    L = lib()
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    mk, nx = (L.minimizerRCiterator, L.minimizerRCnext) if minimizer else (L.modRCiterator, L.modRCnext)
    it = mk(sh, bases.ctypes.data, len(bases))
    u = C.c_uint64(); p = C.c_int(); f = C.c_bool()
    ks, ps, fs = [], [], []
    while nx(it, C.byref(u), C.byref(p), C.byref(f)):
        ks.append(u.value); ps.append(p.value); fs.append(int(f.value))
    L.mgSeqhashRCiteratorDestroy(it)
    return np.array(ks, np.uint64), np.array(ps, np.int32), np.array(fs, np.uint8)


def scan_batch(sh, bases, offsets):
    """seqhashScanBatch: all modimizers of a batch of reads in (read,pos) order.
    Returns (kmer u64[], pos i32[], isF u8[], survStart i64[nReads+1])."""
    L = lib()
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n_reads = len(offsets) - 1
    pk, pp, pf, ps = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    n = L.seqhashScanBatch(sh, bases.ctypes.data, offsets.ctypes.data, n_reads,
                           C.byref(pk), C.byref(pp), C.byref(pf), C.byref(ps))
    if n < 0:
        raise ModgpuError("seqhashScanBatch failed: " + L.mgLastError().decode())

    def take(ptr, ctype, count, dtype):
        a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), (max(count, 1),))[:count].astype(dtype, copy=True)
        _libc.free(ptr)
        return a
    kmer = take(pk, C.c_uint64, n, np.uint64)
    pos = take(pp, C.c_int, n, np.int32)
    isf = take(pf, C.c_bool, n, np.uint8)
    st = take(ps, C.c_int64, n_reads + 1, np.int64)
    return kmer, pos, isf, st


def minimizer_batch(sh, bases, offsets):
    """seqhashMinimizerBatch: what minimizerRCnext returns for every read of a batch, reads in order.
    Returns (hash u64[], pos i32[], isF u8[], start i64[nReads+1])."""
    L = lib()
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n_reads = len(offsets) - 1
    pk, pp, pf, ps = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    n = L.seqhashMinimizerBatch(sh, bases.ctypes.data, offsets.ctypes.data, n_reads,
                                C.byref(pk), C.byref(pp), C.byref(pf), C.byref(ps))
    if n < 0:
        raise ModgpuError("seqhashMinimizerBatch failed: " + L.mgLastError().decode())

    def take(ptr, ctype, count, dtype):
        a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), (max(count, 1),))[:count].astype(dtype, copy=True)
        _libc.free(ptr)
        return a
    return (take(pk, C.c_uint64, n, np.uint64), take(pp, C.c_int, n, np.int32), take(pf, C.c_bool, n, np.uint8),
            take(ps, C.c_int64, n_reads + 1, np.int64))


class DeviceBuffer:
    """A raw device allocation owned through the C ABI (no torch needed)."""

    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        self.nbytes = int(nbytes)
        check(lib().mgDeviceAlloc(C.byref(self.ptr), self.nbytes))

    @classmethod
    def from_numpy(cls, arr):
        arr = np.ascontiguousarray(arr)
        b = cls(max(arr.nbytes, 16))
        if arr.nbytes:
            check(lib().mgMemcpyH2D(b.ptr, arr.ctypes.data, arr.nbytes, None))
            check(lib().mgStreamSynchronize(None))
        return b

    def to_numpy(self, dtype, count):
        out = np.empty(count, dtype)
        if out.nbytes:
            check(lib().mgMemcpyD2H(out.ctypes.data, self.ptr, out.nbytes, None))
        return out

    def free(self):
        if self.ptr:
            lib().mgDeviceFree(self.ptr); self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def pack_host(bases):
    """mgPackHost: bytes 0..3 -> 2-bit packed uint32 words (first base in the top bits)."""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    words = np.zeros(lib().mgPackedWords(len(bases)), np.uint32)
    lib().mgPackHost(bases.ctypes.data, len(bases), words.ctypes.data)
    return words


def modset_arrays(ms):
    """Host view (copies) of value/depth/info for entries 0..max of a Modset*."""
    m = ms.contents
    n = m.max + 1
    return (np.ctypeslib.as_array(m.value, (n,)).copy(),
            np.ctypeslib.as_array(m.depth, (n,)).copy(),
            np.ctypeslib.as_array(m.info, (n,)).copy())


def add_sequence_batch(ms, bases, offsets):
    """mgAddSequenceBatch: modutils.c:19-31 over a batch of reads (GPU). Returns total hashes."""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = lib().mgAddSequenceBatch(ms, bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1)
    if n < 0:
        raise ModgpuError("mgAddSequenceBatch failed: " + lib().mgLastError().decode())
    return n
