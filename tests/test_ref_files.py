"""The reference's on-disk formats for a modmap Reference: <root>.mod + <root>.ref (modmap.c:136-182).

Fixtures modmap_*_files.{mod,ref} were written by the reference itself (modmap -f ref.fa -w stem) and
modmap_*_files.stdout.txt is what it prints for -r stem -q queries.fa (tests/golden/make_golden.py).
"""
import ctypes as C
import gzip
import os
import subprocess

import numpy as np
import pytest

import modimizer_amd as mg
from modimizer_amd import fasta
from tests import util

TAGS = {"k21d64": (21, 64), "k15d8": (15, 8)}
STEMS = ["modmap_k21d64_files", "modmap_k15d8_files", "modmap_many_files"]
MODMAP_REF = os.path.join(util.ROOT, "oracle", "_ref", "modmap_ref")


def strip_timing(text):
    return "\n".join(l for l in text.splitlines() if not l.startswith("user\t") and "resources used" not in l) + "\n"


def arr(p, n, dt=np.uint32):
    return np.ctypeslib.as_array(p, shape=(max(int(n), 1),))[:int(n)].astype(dt).copy()


def ref_arrays(ref):
    r = ref.contents
    ms = r.ms.contents
    m = ms.max + 1
    d = {k: arr(getattr(r, k), r.max) for k in ("index", "offset", "id", "rev")}
    d["depth"] = arr(r.depth, m); d["loc"] = arr(r.loc, m); d["len"] = arr(r.len, r.nSeq)
    d["names"] = [r.names[i].decode() for i in range(r.nSeq)]
    d["max"] = r.max; d["msmax"] = ms.max
    d["value"] = arr(ms.value, m, np.uint64)[1:]
    d["info"] = arr(ms.info, m, np.uint8)[1:]
    d["mdepth"] = arr(ms.depth, m, np.uint16)[1:]
    d["mindex"] = arr(ms.index, ms.tableSize)
    return d


def same_arrays(a, b):
    assert a.keys() == b.keys()
    for k in a:
        if isinstance(a[k], np.ndarray):
            assert np.array_equal(a[k], b[k]), k
        else:
            assert a[k] == b[k], k


def ref_file_mask(raw, ms_max, ref_max):
    """the .ref bytes with the fields that hold process addresses zeroed: Array.base (array.h:43) and
    the DICT's name pointers (dict.c:95)"""
    b = bytearray(raw)
    m = ms_max + 1
    at = 16 + 4 * (4 * ref_max + 2 * m)             # header + index, offset, id, rev + depth, loc
    b[at + 8:at + 16] = bytes(8)                     # Array.base
    dim = int.from_bytes(b[at + 16:at + 20], "little")
    at += 32 + 4 * dim
    bits = int.from_bytes(b[at:at + 4], "little"); n = int.from_bytes(b[at + 4:at + 8], "little")
    at += 8 + 4 * (1 << bits)
    b[at:at + 8 * (n + 1)] = bytes(8 * (n + 1))
    return bytes(b)


def mod_file_mask(raw, table_bits):
    """the .mod bytes without value[0], which the reference never initialises (modset.c:27)"""
    b = bytearray(raw)
    at = 8 + 4 + 4 + 8 + 80 + 4 * (1 << table_bits)
    b[at:at + 8] = bytes(8)
    return bytes(b)


@pytest.mark.parametrize("tag", list(TAGS))
def test_load_reference_files(tag, golden_dir):
    L = mg.lib()
    ref = L.mgReferenceLoad(os.path.join(golden_dir, "modmap_%s_files" % tag).encode())
    d = ref_arrays(ref)
    names, bases, offs = fasta.read_fasta(os.path.join(golden_dir, "ref.fa"))
    assert d["names"] == names and list(d["len"]) == list(np.diff(offs))
    k, w = TAGS[tag]
    sh = ref.contents.ms.contents.hasher.contents
    assert (sh.k, sh.w, sh.seed) == (k, w, 17) and ref.contents.ms.contents.tableBits == 20
    # CSR inverse (modmap.c:74-91): occurrences grouped by modset index, in occurrence order
    assert d["depth"].sum() == d["max"] and d["depth"][0] == 0
    assert np.array_equal(d["loc"][1:], np.cumsum(d["depth"])[:-1])
    order = np.argsort(d["index"], kind="stable")
    assert np.array_equal(d["rev"], order.astype(np.uint32))
    assert d["index"].min() >= 1 and d["index"].max() <= d["msmax"]
    L.mgReferenceDestroy(ref)


@pytest.mark.parametrize("name", STEMS)
def test_write_reference_files_bytes(name, golden_dir, tmp_path):
    """load the reference's files, write them again: same bytes inside the gzip streams (the "many"
    set has 1100 sequences: grown DICT table and length Array)"""
    L = mg.lib()
    stem = os.path.join(golden_dir, name)
    ref = L.mgReferenceLoad(stem.encode())
    out = str(tmp_path / "again")
    L.mgReferenceWrite(ref, out.encode())
    r = ref.contents
    ms_max, ref_max = r.ms.contents.max, r.max
    assert gzip.open(out + ".mod").read() == gzip.open(stem + ".mod").read()
    assert ref_file_mask(gzip.open(out + ".ref").read(), ms_max, ref_max) == \
        ref_file_mask(gzip.open(stem + ".ref").read(), ms_max, ref_max)
    ref2 = L.mgReferenceLoad(out.encode())
    same_arrays(ref_arrays(ref), ref_arrays(ref2))
    L.mgReferenceDestroy(ref); L.mgReferenceDestroy(ref2)


def test_plain_files_are_accepted(golden_dir, tmp_path):
    """fzopen reads gzip or plain alike (utils.c:107-127)"""
    L = mg.lib()
    stem = os.path.join(golden_dir, "modmap_k21d64_files")
    out = str(tmp_path / "plain")
    for ext in (".mod", ".ref"):
        open(out + ext, "wb").write(gzip.open(stem + ext).read())
    a, b = L.mgReferenceLoad(stem.encode()), L.mgReferenceLoad(out.encode())
    same_arrays(ref_arrays(a), ref_arrays(b))
    L.mgReferenceDestroy(a); L.mgReferenceDestroy(b)


def test_many_sequences_shape(tmp_path):
    """more names than the DICT's and the Array's first allocation: table doubled (dict.c:166-186), dim doubled"""
    L = mg.lib()
    stem = os.path.join(util.GOLDEN, "modmap_k21d64_files")
    ref = L.mgReferenceLoad(stem.encode())
    r = ref.contents
    n = 1500
    keep = (r.nSeq, C.cast(r.names, C.c_void_p).value, C.cast(r.len, C.c_void_p).value)   # addresses, not views
    names = (C.c_char_p * n)(*[("contig_%d" % i).encode() for i in range(n)])
    lens = (C.c_uint32 * n)(*range(100, 100 + n))
    r.nSeq, r.names, r.len = n, C.cast(names, C.POINTER(C.c_char_p)), C.cast(lens, mg.U32P)
    out = str(tmp_path / "many")
    L.mgReferenceWrite(ref, out.encode())
    r.nSeq, r.names, r.len = keep[0], C.cast(keep[1], C.POINTER(C.c_char_p)), C.cast(keep[2], mg.U32P)
    raw = gzip.open(out + ".ref").read()
    at = 16 + 4 * (4 * r.max + 2 * (r.ms.contents.max + 1))
    magic, dim, size, mx = (int.from_bytes(raw[at + o:at + o + 4], "little") for o in (0, 16, 20, 24))
    assert (magic, dim, size, mx) == (8918274, 2048, 4, n)
    at += 32 + 4 * dim
    bits, cnt = int.from_bytes(raw[at:at + 4], "little"), int.from_bytes(raw[at + 4:at + 8], "little")
    assert (bits, cnt) == (13, n)                      # 1024 -> 2048 at 308 names, 4096 at 615, 8192 at 1229
    table = np.frombuffer(raw[at + 8:at + 8 + 4 * (1 << bits)], np.int32)
    assert sorted(table[table > 0]) == list(range(1, n + 1))
    ref2 = L.mgReferenceLoad(out.encode())
    assert [ref2.contents.names[i].decode() for i in range(n)] == ["contig_%d" % i for i in range(n)]
    assert list(arr(ref2.contents.len, n)) == list(range(100, 100 + n))
    L.mgReferenceDestroy(ref); L.mgReferenceDestroy(ref2)


@pytest.mark.skipif(not os.path.exists(MODMAP_REF), reason="oracle/_ref/modmap_ref not built")
@pytest.mark.parametrize("tag", list(TAGS))
def test_reference_program_reads_our_files(tag, golden_dir, tmp_path):
    """modmap -r <files written here> -q queries.fa prints what it prints for its own files"""
    L = mg.lib()
    ref = L.mgReferenceLoad(os.path.join(golden_dir, "modmap_%s_files" % tag).encode())
    out = str(tmp_path / "ours")
    L.mgReferenceWrite(ref, out.encode())
    L.mgReferenceDestroy(ref)
    r = subprocess.run([MODMAP_REF, "-r", out, "-q", os.path.join(golden_dir, "queries.fa")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-500:]
    got = strip_timing(r.stdout).replace(out, "STEM").replace(os.path.join(golden_dir, "queries.fa"), "queries.fa")
    want = util.golden_text("modmap_%s_files.stdout.txt" % tag).replace("modmap_%s_files" % tag, "STEM")
    assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(TAGS))
def test_gpu_built_reference_files_match(tag, golden_dir, tmp_path):
    """reference built on the GPU from ref.fa, written here: the reference's own files byte for byte
    (index[] slot layout included), apart from value[0] and the stored addresses"""
    L = mg.lib()
    k, w = TAGS[tag]
    sh = mg.seqhashCreate(k, w, 17)
    ms = mg.modsetCreate(sh, 20)
    ref = C.cast(L.mgReferenceCreate(ms, 1 << 26), C.POINTER(mg.MgReference))
    names, bases, offs = fasta.read_fasta(os.path.join(golden_dir, "ref.fa"))
    with mg.CFile(str(tmp_path / "log.txt"), "w") as f:
        cn = (C.c_char_p * len(names))(*[n.encode() for n in names])
        assert L.mgReferenceRead(ref, bases.ctypes.data, offs.ctypes.data, len(names), cn, True, f) == 0
    out = str(tmp_path / "gpu")
    L.mgReferenceWrite(ref, out.encode())
    stem = os.path.join(golden_dir, "modmap_%s_files" % tag)
    r = ref.contents
    assert mod_file_mask(gzip.open(out + ".mod").read(), 20) == mod_file_mask(gzip.open(stem + ".mod").read(), 20)
    assert ref_file_mask(gzip.open(out + ".ref").read(), r.ms.contents.max, r.max) == \
        ref_file_mask(gzip.open(stem + ".ref").read(), r.ms.contents.max, r.max)
    L.mgReferenceDestroy(ref)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(TAGS))
def test_query_against_loaded_reference(tag, golden_dir, tmp_path):
    """modmap -r stem -q queries.fa with the lookups on the GPU"""
    L = mg.lib()
    ref = L.mgReferenceLoad(os.path.join(golden_dir, "modmap_%s_files" % tag).encode())
    qn, qb, qo = fasta.read_fasta(os.path.join(golden_dir, "queries.fa"))
    out = str(tmp_path / "q.txt")
    with mg.CFile(out, "w") as f:
        cq = (C.c_char_p * len(qn))(*[n.encode() for n in qn])
        assert L.mgQueryProcess(ref, qb.ctypes.data, qo.ctypes.data, len(qn), cq, f) == 0
    want = [l for l in util.golden_text("modmap_%s_files.stdout.txt" % tag).splitlines() if l[:2] in ("Q\t", "M\t")]
    assert open(out).read().splitlines() == want
    L.mgReferenceDestroy(ref)


MODUTILS_REF = os.path.join(util.ROOT, "oracle", "_ref", "modutils_ref")


def _host_modset(bits=24, n=300_000, seed=5):
    """a host-only Modset (scalar API, no device) large enough for its .mod to be several gzip members (index[] alone is 4 << bits bytes)"""
    L = mg.lib()
    sh = mg.seqhashCreate(21, 64, 17)
    ms = mg.modsetCreate(sh, bits)
    rng = np.random.default_rng(seed)
    for kmer in rng.integers(0, 1 << 42, n, dtype=np.uint64):
        ix = L.modsetIndexFind(ms, int(kmer), 1)
        ms.contents.depth[ix] += 1
    return sh, ms


def test_multi_member_gzip_is_the_single_stream(tmp_path):
    """mgGzipOpenWrite: the members of the file, decompressed one after the other, are exactly the bytes modsetWrite (modset.c:79-88)
    hands over -- written here into a plain file for comparison -- and there are several of them"""
    import ctypes as C
    L = mg.lib()
    sh, ms = _host_modset()
    plain, gz = str(tmp_path / "plain.mod"), str(tmp_path / "multi.mod")
    with mg.CFile(plain, "w") as f:
        L.modsetWrite(ms, f)
    libc = C.CDLL(None); libc.fclose.argtypes = [C.c_void_p]
    for threads in ("1", "5"):
        with mg.knobs(GZIP_THREADS=threads):
            f = L.mgGzipOpenWrite(gz.encode())
            assert f
            L.modsetWrite(ms, C.c_void_p(f))
            assert libc.fclose(C.c_void_p(f)) == 0
        raw = open(gz, "rb").read()
        import re
        assert len(re.findall(rb"\x1f\x8b\x08\x04\x00\x00\x00\x00[\x00\x02\x04]\x03\x0c\x00MG\x08\x00", raw)) >= 4      # 64 MiB of index[] in members of 16 MiB, each with its sizes in the extra field
        assert gzip.decompress(raw) == open(plain, "rb").read()
    # ... and read back by the team (mgGzipOpenRead: the members found by the sizes in their extra fields): modsetRead gets the same set
    f = L.mgGzipOpenRead(gz.encode())
    assert f
    L.modsetRead.restype = C.POINTER(mg.Modset)
    ms2 = L.modsetRead(C.c_void_p(f))
    assert libc.fclose(C.c_void_p(f)) == 0
    a, b = ms.contents, ms2.contents
    assert (a.max, a.tableBits) == (b.max, b.tableBits)
    for name, n, dt in (("index", 1 << a.tableBits, np.uint32), ("value", a.max + 1, np.uint64), ("depth", a.max + 1, np.uint16), ("info", a.max + 1, np.uint8)):
        x, y = arr(getattr(a, name), n, dt), arr(getattr(b, name), n, dt)
        assert np.array_equal(x[1:] if name == "value" else x, y[1:] if name == "value" else y), name
    L.modsetDestroy(ms2)
    # reads of every size across member boundaries: the bytes are the plain file's
    want = open(plain, "rb").read()
    libc.fread.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]; libc.fread.restype = C.c_size_t
    for sizes in ([7, 104, (16 << 20) - 50, 1, 33 << 20, 5], [len(want) + 10], [1 << 20] * 70):
        f = L.mgGzipOpenRead(gz.encode()); got = b""
        for n in sizes:
            buf = C.create_string_buffer(n)
            k = libc.fread(buf, 1, n, C.c_void_p(f)); got += buf.raw[:k]
        assert libc.fclose(C.c_void_p(f)) == 0
        assert got == want[:len(got)] and len(got) == min(sum(sizes), len(want))
    # a gzip file somebody else wrote, a plain file: not this reader's (the callers then take gzopen)
    other = str(tmp_path / "other.gz")
    with gzip.open(other, "wb") as g:
        g.write(b"hello" * 1000)
    assert not L.mgGzipOpenRead(other.encode()) and not L.mgGzipOpenRead(plain.encode())
    # small writes only: they collect and still come out as one valid stream; an empty file is one empty member
    f = L.mgGzipOpenWrite(gz.encode())
    libc.fputs.argtypes = [C.c_char_p, C.c_void_p]
    for i in range(1000):
        libc.fputs(b"line %d\n" % i, C.c_void_p(f))
    assert libc.fclose(C.c_void_p(f)) == 0
    assert gzip.decompress(open(gz, "rb").read()) == b"".join(b"line %d\n" % i for i in range(1000))
    f = L.mgGzipOpenWrite(gz.encode()); assert libc.fclose(C.c_void_p(f)) == 0
    assert gzip.decompress(open(gz, "rb").read()) == b""
    L.modsetDestroy(ms)


@pytest.mark.skipif(not os.path.exists(MODUTILS_REF), reason="oracle/_ref/modutils_ref not built")
def test_reference_program_reads_a_multi_member_mod(tmp_path):
    """the reference's own modutils -r (fzopen + gzread, utils.c:107-127; modsetRead modset.c:90-104) on a .mod of several gzip members
    prints the summary it prints for the same set written by its own single gzwrite stream"""
    import ctypes as C
    L = mg.lib()
    sh, ms = _host_modset(bits=24, n=200_000, seed=9)
    ours = str(tmp_path / "ours.mod")
    f = L.mgGzipOpenWrite(ours.encode())
    L.modsetWrite(ms, C.c_void_p(f))
    libc = C.CDLL(None); libc.fclose.argtypes = [C.c_void_p]
    assert libc.fclose(C.c_void_p(f)) == 0
    theirs = str(tmp_path / "theirs.mod")
    with gzip.open(theirs, "wb", compresslevel=6) as g:               # one member, as gzopen "w" makes it
        g.write(gzip.decompress(open(ours, "rb").read()))
    outs = []
    for path in (ours, theirs):
        r = subprocess.run([MODUTILS_REF, "-r", path], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-500:]
        outs.append(strip_timing(r.stdout + r.stderr).replace(path, "FILE"))
    assert outs[0] == outs[1] and "number of entries %d" % ms.contents.max in outs[0]
    L.modsetDestroy(ms)
