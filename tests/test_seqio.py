"""The file front end (mg_seqio.c) against the reference's seqio.c, through what modutils/modmap print.

Fixtures: tests/golden/make_golden.py (make_seqio_inputs, gen_seqio): `modutils -c 20 15 4 17 -a <file>
-wt dump` run by the reference program on each text file.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import modimizer_amd as mg
from tests import util

FILES = {"mixed.fa": "mixed_fa", "mixed.fa.gz": "mixed_fa", "unterminated.fa": "unterminated_fa", "mixed.fq": "mixed_fq", "header_last.fa": "header_last_fa"}


def parse_file(path, max_bases, threads):
    """every record of the file through mgSeqOpen/mgSeqNextBatch: (names, [bases arrays])"""
    L = mg.lib()
    os.environ["MODGPU_PARSE_THREADS"] = str(threads); L.mgReloadKnobs()
    r = L.mgSeqOpen(path.encode())
    assert r
    names, seqs = [], []
    b = mg.MgSeqBatch()
    while True:
        n = L.mgSeqNextBatch(r, max_bases, C.byref(b))
        if n == 0:
            break
        offs = np.ctypeslib.as_array(b.offsets, shape=(n + 1,)).copy()
        assert offs[0] == 0 and offs[-1] == b.total and np.all(np.diff(offs) >= 0)
        bases = np.ctypeslib.as_array(b.bases, shape=(max(int(b.total), 1),))[:b.total].astype(np.uint8)
        for i in range(n):
            names.append(b.names[i].decode()); seqs.append(bases[offs[i]:offs[i + 1]].copy())
        L.mgSeqBatchFree(C.byref(b))
    L.mgSeqClose(r)
    del os.environ["MODGPU_PARSE_THREADS"]; L.mgReloadKnobs()
    return names, seqs


def added_line(golden_stdout):
    return [l for l in golden_stdout.splitlines() if l.startswith("added ")][0]


@pytest.mark.parametrize("fname", list(FILES))
@pytest.mark.parametrize("max_bases,threads", [(1 << 40, 1), (1, 3), (5000, 8)])
def test_parse_matches_reference_program(fname, max_bases, threads, golden_dir, tmp_path):
    """parse here, add with the oracle: the line and the -wt dump the reference prints for the file"""
    from oracle import pyoracle as orc
    names, seqs = parse_file(os.path.join(golden_dir, fname), max_bases, threads)
    h = orc.Hasher(15, 4, 17)
    ms = orc.Modset(h, 20)
    tot_hash = sum(ms.add_sequence(s) for s in seqs)
    line = "added %d sequences total length %d total hashes %d, new max %d" % (
        len(seqs), sum(len(s) for s in seqs), tot_hash, ms.max)
    tag = fname.replace(".", "_")
    assert line == added_line(util.golden_text("seqio_%s.stdout.txt" % tag))
    assert ms.text_dump(str(tmp_path / "d.txt")) == util.golden_text("seqio_%s.dump.txt" % FILES[fname])
    ms.close()


def test_names_and_shapes(golden_dir):
    names, seqs = parse_file(os.path.join(golden_dir, "mixed.fa"), 1 << 40, 4)
    assert names[:10] == ["plain60", "lower", "withN", "iupac", "crlf", "empty", "blank_lines", "gt_inside", "oneline", "short"]
    assert len(seqs[5]) == 0 and list(seqs[9]) == [0, 1, 2, 3, 0, 1]
    assert len(seqs[0]) == 3000 and len(seqs[4]) == 2000 and len(seqs[6]) == 2000 and len(seqs[7]) == 1004   # ">notaheader " inside a line: n, t, a, a are bases
    assert all(int(s.max(initial=0)) <= 3 for s in seqs)
    fq_names, fq = parse_file(os.path.join(golden_dir, "mixed.fq"), 700, 2)
    assert fq_names[0] == "fq0" and fq_names[-1] == "fqempty" and len(fq[-1]) == 0 and len(fq) == 61


def test_unterminated_last_record_is_reported(golden_dir, capfd):
    names, seqs = parse_file(os.path.join(golden_dir, "unterminated.fa"), 1 << 40, 2)
    assert names == ["plain60", "lower", "withN"]
    assert capfd.readouterr().err == util.golden_text("seqio_unterminated_fa.stderr.txt")


def test_header_line_as_last_line_is_an_incomplete_record(golden_dir, capfd):
    """a FASTA file that ends with a header line: the reference reads on for the sequence, meets the end of the file, reports
    "incomplete sequence record line N" and does NOT return the record (seqio.c:213-217,314) -- golden from the reference program"""
    names, seqs = parse_file(os.path.join(golden_dir, "header_last.fa"), 1 << 40, 2)
    assert names == ["a", "b"] and [len(s) for s in seqs] == [32, 28]
    assert capfd.readouterr().err == util.golden_text("seqio_header_last_fa.stderr.txt")


def test_unreadable_and_empty(tmp_path, capfd):
    L = mg.lib()
    assert not L.mgSeqOpen(str(tmp_path / "absent.fa").encode())
    (tmp_path / "empty.fa").write_text("")
    assert not L.mgSeqOpen(str(tmp_path / "empty.fa").encode())
    assert "unreadable or empty" in capfd.readouterr().err          # seqio.c:42


BAD = {
    "no_plus": ("@a\nACGT\nIIII\n@b\nAC\n+\nII\n", "missing + FASTQ line 3"),
    "qual_len": ("@a\nACGT\n+\nIII\n", "qual not same length as seq line 4"),
    "no_at": ("@a\nACGT\n+\nIIII\nACGT\n+\nIIII\n", "no initial @ for FASTQ record line 5"),
}


@pytest.mark.parametrize("case", list(BAD))
def test_malformed_fastq_dies_like_the_reference(case, tmp_path):
    text, msg = BAD[case]
    path = tmp_path / "bad.fq"
    path.write_text(text)
    code = ("import ctypes as C, modimizer_amd as mg; L = mg.lib(); r = L.mgSeqOpen(%r.encode()); "
            "b = mg.MgSeqBatch(); L.mgSeqNextBatch(r, 1 << 30, C.byref(b))" % str(path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=util.ROOT,
                       env=dict(os.environ, MODGPU_NO_TORCH="1"))
    assert r.returncode == 255 and ("FATAL ERROR: " + msg) in r.stderr


def test_large_record_and_many_small(tmp_path):
    """one record larger than the first window (16 MiB, seqio.c:36) and many tiny ones after it"""
    rng = np.random.default_rng(3)
    big = rng.integers(0, 4, 20_000_000).astype(np.uint8)
    small = [rng.integers(0, 4, int(n)).astype(np.uint8) for n in rng.integers(1, 200, 5000)]
    letters = np.frombuffer(b"ACGT", np.uint8)
    path = tmp_path / "big.fa"
    with open(path, "wb") as f:
        f.write(b">big\n")
        t = letters[big]
        for i in range(0, len(t), 1_000_000):
            f.write(t[i:i + 1_000_000].tobytes() + b"\n")
        for i, s in enumerate(small):
            f.write(b">s%d\n" % i + letters[s].tobytes() + b"\n")
    for max_bases, threads in ((1 << 40, 8), (1000, 2)):
        names, seqs = parse_file(str(path), max_bases, threads)
        assert len(seqs) == 5001 and names[0] == "big" and names[-1] == "s4999"
        assert np.array_equal(seqs[0], big)
        assert all(np.array_equal(a, b) for a, b in zip(seqs[1:], small))


@pytest.mark.parametrize("fastq", [False, True])
def test_records_whose_text_is_a_multiple_of_the_work_unit(fastq, tmp_path):
    """raw sequence text of exactly 1 MiB and 2 MiB (the conversion work unit, mg_seqio.c UNIT_BYTES): the unit
    count of the bookkeeping pass must equal the units the cutting pass creates (a surplus unit was read
    uninitialised: wild reads in the conversion workers)"""
    rng = np.random.default_rng(11)
    letters = np.frombuffer(b"ACGT", np.uint8)
    mib = 1 << 20
    path = tmp_path / ("mult.fq" if fastq else "mult.fa")
    want = []
    with open(path, "wb") as f:
        if fastq:
            for i, n in enumerate((mib, 5, 2 * mib, mib - 1, mib + 1, 0)):
                s = rng.integers(0, 4, n).astype(np.uint8)
                want.append(s)
                f.write(b"@q%d\n" % i + letters[s].tobytes() + b"\n+\n" + b"I" * n + b"\n")
        else:
            # FASTA: the raw text includes the newlines: 16384 lines of 63 bases + newline = 1 MiB exactly
            for i, lines in enumerate((16384, 3, 32768, 16383, 16385)):
                s = rng.integers(0, 4, lines * 63).astype(np.uint8)
                want.append(s)
                f.write(b">r%d\n" % i)
                f.write(b"".join(letters[s[j:j + 63]].tobytes() + b"\n" for j in range(0, len(s), 63)))
    for max_bases, threads in ((1 << 40, 4), (1 << 40, 1), (100, 3)):
        names, got = parse_file(str(path), max_bases, threads)
        assert len(got) == len(want)
        assert all(np.array_equal(a, b) for a, b in zip(got, want))


def test_many_short_fastq_records_across_slices(tmp_path, capfd):
    """records cut by the pool: line ends listed per 4 MiB slice, a record = four lines; a bad record
    deep in the file is reported with its line number, an unfinished last record is left out"""
    rng = np.random.default_rng(5)
    n = 60_000
    lens = rng.integers(0, 260, n)
    letters = np.frombuffer(b"ACGTN", np.uint8)
    seqs = [letters[rng.integers(0, 5, int(l))] for l in lens]
    path = tmp_path / "short.fq"
    with open(path, "wb") as f:
        for i, s in enumerate(seqs):
            f.write(b"@q%d extra words\n" % i + s.tobytes() + b"\n+\n" + b"I" * len(s) + b"\n")
        f.write(b"@tail\nACGT\n+\nII")                          # no closing newline
    assert os.path.getsize(path) > 3 * (4 << 20)
    conv = np.zeros(256, np.uint8); conv[ord("C")] = 1; conv[ord("G")] = 2; conv[ord("T")] = 3
    for max_bases, threads in ((1 << 40, 8), (1_000_000, 5), (1 << 40, 1)):
        names, got = parse_file(str(path), max_bases, threads)
        assert "incomplete sequence record line %d" % (4 * n + 4) in capfd.readouterr().err
        assert len(got) == n and names[0] == "q0" and names[-1] == "q%d" % (n - 1)
        assert all(np.array_equal(a, conv[b]) for a, b in zip(got, seqs))
    # the same file with record 41 234 broken
    bad = 41_234
    with open(path, "wb") as f:
        for i, s in enumerate(seqs):
            f.write(b"@q%d\n" % i + s.tobytes() + (b"\n-\n" if i == bad else b"\n+\n") + b"I" * len(s) + b"\n")
    code = ("import ctypes as C, modimizer_amd as mg; L = mg.lib(); r = L.mgSeqOpen(%r.encode()); "
            "b = mg.MgSeqBatch()\nwhile L.mgSeqNextBatch(r, 2_000_000, C.byref(b)): print(b.nSeq, flush=True)" % str(path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=util.ROOT,
                       env=dict(os.environ, MODGPU_NO_TORCH="1", MODGPU_PARSE_THREADS="6"))
    assert r.returncode == 255 and ("FATAL ERROR: missing + FASTQ line %d" % (4 * bad + 3)) in r.stderr
    taken = sum(int(x) for x in r.stdout.split())
    assert 0 < taken <= bad                                       # whole batches before the bad record were handed over


def bgzf_bytes(data, block, eof=True):
    """blocked gzip as bgzip writes it: members of `block` text bytes with the 'BC' extra field, then the empty end marker"""
    import struct
    import zlib
    out = []
    for i in range(0, len(data), block):
        ch = data[i:i + block]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        pay = c.compress(ch) + c.flush()
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(pay) + 25) + pay
                   + struct.pack("<II", zlib.crc32(ch), len(ch)))
    if eof:
        out.append(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    return b"".join(out)


@pytest.mark.parametrize("fname", ["mixed.fa", "mixed.fq", "many.fa"])
def test_blocked_gzip_is_read_like_the_text(fname, golden_dir, tmp_path):
    """BGZF members are inflated by the pool; the result is what zlib's stream reader (the reference's path) gives"""
    import gzip
    raw = open(os.path.join(golden_dir, fname), "rb").read()
    want = parse_file(os.path.join(golden_dir, fname), 1 << 40, 1)
    path = str(tmp_path / "x.gz")

    def same(got):
        return got[0] == want[0] and all(np.array_equal(a, b) for a, b in zip(got[1], want[1]))
    for block in (65280, 999, 37):
        open(path, "wb").write(bgzf_bytes(raw, block))
        assert same(parse_file(path, 1 << 40, 4)) and same(parse_file(path, 3000, 3))
    # blocks + an end marker in mid-file + more blocks + an ordinary gzip member: zlib's reader takes over there
    half = raw.index(b"\n", len(raw) // 2) + 1
    open(path, "wb").write(bgzf_bytes(raw[:half], 4000) + bgzf_bytes(raw[half:half + 100], 50, eof=False) + gzip.compress(raw[half + 100:]))
    assert same(parse_file(path, 1 << 40, 4))


def test_blocked_gzip_large_and_corrupt(tmp_path):
    rng = np.random.default_rng(9)
    letters = np.frombuffer(b"ACGT", np.uint8)
    reads = letters[rng.integers(0, 4, (150_000, 120))]
    text = b"".join(b">r%d\n" % i + reads[i].tobytes() + b"\n" for i in range(len(reads)))    # 19 MB: several windows
    blob = bgzf_bytes(text, 65280)
    path = tmp_path / "big.fa.gz"
    path.write_bytes(blob)
    names, seqs = parse_file(str(path), 1 << 40, 8)
    assert len(seqs) == len(reads) and names[-1] == "r%d" % (len(reads) - 1)
    conv = np.zeros(256, np.uint8); conv[ord("C")] = 1; conv[ord("G")] = 2; conv[ord("T")] = 3
    assert np.array_equal(np.concatenate(seqs), conv[reads].ravel())
    bad = bytearray(blob); bad[len(blob) // 2] ^= 0x55
    path.write_bytes(bytes(bad))
    code = ("import ctypes as C, modimizer_amd as mg; L = mg.lib(); r = L.mgSeqOpen(%r.encode()); "
            "b = mg.MgSeqBatch()\nwhile L.mgSeqNextBatch(r, 1 << 40, C.byref(b)): pass" % str(path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=util.ROOT,
                       env=dict(os.environ, MODGPU_NO_TORCH="1"))
    assert r.returncode != 0 and "FATAL ERROR" in r.stderr


# ------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("fname", list(FILES))
def test_add_sequence_file_gpu(fname, golden_dir, tmp_path):
    """modutils -c 20 15 4 17 -a <file> -wt: file -> parse -> pack -> scan -> modset, all here"""
    L = mg.lib()
    sh = mg.seqhashCreate(15, 4, 17)
    ms = mg.modsetCreate(sh, 20)
    out = str(tmp_path / "o.txt")
    os.environ["MODGPU_FILE_BATCH_MBP"] = "1"; L.mgReloadKnobs()
    try:
        with mg.CFile(out, "w") as f:
            assert L.mgAddSequenceFile(ms, os.path.join(golden_dir, fname).encode(), f) == 0
    finally:
        del os.environ["MODGPU_FILE_BATCH_MBP"]; L.mgReloadKnobs()
    tag = fname.replace(".", "_")
    assert open(out).read().strip() == added_line(util.golden_text("seqio_%s.stdout.txt" % tag))
    dump = str(tmp_path / "d.txt")
    with mg.CFile(dump, "w") as f:
        L.mgModsetWriteText(ms, f)
    assert open(dump).read() == util.golden_text("seqio_%s.dump.txt" % FILES[fname])
    L.modsetDestroy(ms)


@pytest.mark.gpu
@pytest.mark.parametrize("parser", ["device", "host", "device_small_windows"])
@pytest.mark.parametrize("tag", list(util.MODMAP_TAGS))
def test_modmap_from_files_gpu(tag, parser, golden_dir, tmp_path):
    """modmap -f ref.fa -q queries.fa from the files themselves (queries in several batches): the text parsed on the device
    (mg_textgpu.hip: the batches stay there, the record ids come out of the pinned text windows), by the host parser, and on
    the device with 4 KiB windows and 3000-base batches (ids, records and batches cut everywhere)"""
    L = mg.lib()
    kn = {"device": dict(TEXT_HOST=0), "host": dict(TEXT_HOST=1), "device_small_windows": dict(TEXT_HOST=0, TEXT_WINDOW_KB=4, FILE_BATCH_BASES=3000)}[parser]
    with mg.knobs(**kn):
        _modmap_from_files(L, tag, golden_dir, tmp_path)


def _modmap_from_files(L, tag, golden_dir, tmp_path):
    k, w = util.MODMAP_TAGS[tag]
    sh = mg.seqhashCreate(k, w, 17)
    ms = mg.modsetCreate(sh, 20)
    ref = L.mgReferenceCreate(ms, 1 << 26)
    out = str(tmp_path / "mm.txt")
    os.environ["MODGPU_FILE_BATCH_MBP"] = "1"; L.mgReloadKnobs()
    try:
        with mg.CFile(out, "w") as f:
            assert L.mgReferenceFastaRead(ref, os.path.join(golden_dir, "ref.fa").encode(), True, f) == 0
            assert L.mgQueryFile(ref, os.path.join(golden_dir, "queries.fa").encode(), f) == 0
    finally:
        del os.environ["MODGPU_FILE_BATCH_MBP"]; L.mgReloadKnobs()
    want = util.golden_text("modmap_%s.stdout.txt" % tag).splitlines()[1:]      # without the "initialised" line
    assert open(out).read().splitlines() == want
    L.mgReferenceDestroy(ref)


@pytest.mark.parametrize("fname", ["mixed.fa", "mixed.fq", "many.fa", "unterminated.fa"])
def test_portable_loops_equal_the_avx2_ones(fname, golden_dir):
    """MODGPU_NO_AVX2=1 (a fresh process: the choice is made once) parses every golden text file to the same records"""
    code = ("import sys, os, hashlib, numpy as np\nsys.path.insert(0, %r)\nfrom tests.test_seqio import parse_file\n"
            "names, seqs = parse_file(%r, 5000, 3)\nh = hashlib.sha256()\n"
            "[h.update(n.encode() + b'|' + s.tobytes() + b';') for n, s in zip(names, seqs)]\nprint(len(seqs), h.hexdigest())"
            % (util.ROOT, os.path.join(golden_dir, fname)))
    outs = []
    for no in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=util.ROOT,
                           env=dict(os.environ, MODGPU_NO_TORCH="1", MODGPU_NO_AVX2=no))
        assert r.returncode == 0, r.stderr[-1500:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1] and int(outs[0].split()[0]) > 0
