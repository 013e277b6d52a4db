"""The FASTA front end ON THE DEVICE (mg_textgpu.hip) against the host parser (mg_seqio.c, itself pinned to the reference's
seqio.c by tests/test_seqio.py) and against what the reference program prints for the golden text files.

mgAddSequenceFile takes the device parser by itself for plain FASTA text; MODGPU_TEXT_HOST=1 forces the host parser,
MODGPU_TEXT_WINDOW_KB / MODGPU_FILE_BATCH_BASES put window and batch edges everywhere (mg.knobs: the library reads its knobs once, mgReloadKnobs again)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import modimizer_amd as mg
from modimizer_amd import fasta
from tests import util
from tests.test_seqio import parse_file, added_line

pytestmark = pytest.mark.gpu


def device_records(path):
    L = mg.lib()
    pb, po, n = C.c_void_p(), C.c_void_p(), C.c_int64()
    rc = L.mgTextParseFileDevice(path.encode(), C.byref(pb), C.byref(po), C.byref(n))
    if rc:
        return rc, None
    offs = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_int64)), (n.value + 1,)).copy()
    bases = np.ctypeslib.as_array(C.cast(pb, C.POINTER(C.c_uint8)), (max(int(offs[-1]), 1),))[:int(offs[-1])].copy()
    mg._libc.free(pb); mg._libc.free(po)
    return 0, [bases[offs[i]:offs[i + 1]] for i in range(n.value)]


def crafted_fasta(rng, n_rec, kind):
    """text that exercises the parser: wrapped / unwrapped / CR LF lines, lower case, N, IUPAC and junk bytes, '>' inside a line and
    at the start of a sequence line (which makes it a record start, for both parsers), empty records, headers back to back, blank
    lines, tabs in headers, a very long single-line record"""
    letters = np.frombuffer(b"ACGTacgtNnRYKMxX*-. \t>1", np.uint8)
    wt = np.array([20, 20, 20, 20, 5, 5, 5, 5, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1], float); wt /= wt.sum()
    lines = []
    for r in range(n_rec):
        eol = b"\r" if rng.random() < 0.2 else b""
        lines.append(b">rec%d some description > with\tjunk" % r + eol)
        if kind == "tiny":
            n = int(rng.integers(0, 5))
        elif kind == "long" and r == n_rec // 2:
            n = 300_000
        else:
            n = int(rng.integers(0, 3000))
        seq = letters[rng.choice(len(letters), n, p=wt)].tobytes()
        width = int(rng.choice([0, 1, 7, 60, 80, 4096]))
        body = [seq] if (width == 0 or not seq) else [seq[i:i + width] for i in range(0, len(seq), width)]
        lines += [l + eol for l in body]
        if rng.random() < 0.1:
            lines.append(b"")                                                       # a blank line
    return b"\n".join(lines) + b"\n"


def last_line_is_header(text):
    """a FASTA text whose last line starts with '>': the reference reports that record as incomplete and does not return it
    (seqio.c:213-217,314), and the device parser leaves such a file to the host parser (-2)"""
    body = text[:-1] if text.endswith(b"\n") else text
    return body[body.rfind(b"\n") + 1:].startswith(b">")


@pytest.mark.parametrize("kind,n_rec,seed", [("mixed", 300, 1), ("tiny", 5000, 2), ("long", 40, 3), ("mixed", 3, 4)])
@pytest.mark.parametrize("window_kb,batch_bases", [(0, 0), (4, 0), (8, 3000), (64, 100000)])
def test_device_parser_equals_host_parser(kind, n_rec, seed, window_kb, batch_bases, tmp_path):
    path = str(tmp_path / "t.fa")
    text = crafted_fasta(np.random.default_rng(seed), n_rec, kind)
    if last_line_is_header(text):
        text += b"ACGT\n"                                   # (that case has its own test: test_device_parser_declines_what_it_does_not_take)
    open(path, "wb").write(text)
    _, want = parse_file(path, 1 << 40, 4)
    env = {}
    if window_kb:
        env["TEXT_WINDOW_KB"] = window_kb
    if batch_bases:
        env["FILE_BATCH_BASES"] = batch_bases
    with mg.knobs(**env):
        rc, got = device_records(path)
    assert rc == 0, mg.lib().mgLastError()
    assert len(got) == len(want)
    for i, (a, b) in enumerate(zip(got, want)):
        assert np.array_equal(a, b), (i, len(a), len(b))


def crafted_fastq(rng, n_rec, crlf=False, max_len=300):
    """valid FASTQ that tries the parser: quality lines that start with '@' or '+', empty reads, lower case, N and junk bytes in the
    sequence line (kept, as (char) -2: seqio.c:328-331), optionally CR LF line ends (the CR then counts as a base and as a quality)"""
    letters = np.frombuffer(b"ACGTacgtNnRYx*.", np.uint8)
    wt = np.array([20, 20, 20, 20, 5, 5, 5, 5, 2, 1, 1, 1, 1, 1, 1], float); wt /= wt.sum()
    qual = np.frombuffer(b"@+!5IFHJ#>", np.uint8)
    eol = b"\r\n" if crlf else b"\n"
    out = []
    for r in range(n_rec):
        n = int(rng.integers(0, max_len)) if rng.random() > 0.05 else 0
        out.append(b"@read%d/1 extra" % r + eol)
        out.append(letters[rng.choice(len(letters), n, p=wt)].tobytes() + eol)
        out.append((b"+" if r % 3 else b"+read%d/1 extra" % r) + eol)
        out.append(qual[rng.integers(0, len(qual), n)].tobytes() + eol)
    return b"".join(out)


@pytest.mark.parametrize("n_rec,seed,crlf", [(2000, 1, False), (500, 2, True), (1, 3, False), (30000, 4, False)])
@pytest.mark.parametrize("window_kb,batch_bases", [(0, 0), (4, 0), (8, 3000), (64, 100000)])
def test_device_fastq_parser_equals_host_parser(n_rec, seed, crlf, window_kb, batch_bases, tmp_path):
    path = str(tmp_path / "t.fq")
    open(path, "wb").write(crafted_fastq(np.random.default_rng(seed), n_rec, crlf, 40 if n_rec > 10000 else 300))
    _, want = parse_file(path, 1 << 40, 4)
    env = {}
    if window_kb:
        env["TEXT_WINDOW_KB"] = window_kb
    if batch_bases:
        env["FILE_BATCH_BASES"] = batch_bases
    with mg.knobs(**env):
        rc, got = device_records(path)
    assert rc == 0, mg.lib().mgLastError()
    assert len(got) == len(want) == n_rec
    for i, (a, b) in enumerate(zip(got, want)):
        assert np.array_equal(a & 3, b & 3), (i, len(a), len(b))           # the host parser's batch keeps (char) -2 for other bytes: 2 once packed


BROKEN = {
    "no_plus": lambda recs: recs[:700] + [recs[700].replace(b"\n+", b"\n-", 1)] + recs[701:],
    "no_at": lambda recs: recs[:901] + [b"#" + recs[901][1:]] + recs[902:],
    "short_qual": lambda recs: recs[:650] + [recs[650][:-2] + b"\n"] + recs[651:],
    "truncated": lambda recs: recs[:1200] + [b"\n".join(recs[1200].split(b"\n")[:2]) + b"\n"],
    "three_lines_more": lambda recs: recs + [b"@last\nACGT\n+\n"],
}


@pytest.mark.parametrize("kind", list(BROKEN))
def test_fastq_that_breaks_the_rules_goes_to_the_host_parser(kind, tmp_path):
    """a record without its '+', without its '@', with a quality line of another length; a file that ends in the middle of a record:
    the device parser adds what comes before (whole batches), then the host parser takes over at the first record not yet added and says
    what the reference says, with the reference's line number (seqio.c:213-217,326-339).  Same stderr, exit code, "added" line and
    modset as with the host parser from the start."""
    rng = np.random.default_rng(9)
    text = crafted_fastq(rng, 1500)
    lines = text.split(b"\n")[:-1]
    recs = [b"\n".join(lines[4 * i:4 * i + 4]) + b"\n" for i in range(len(lines) // 4)]
    for r in (650, 700, 901, 1200):                                                 # records the breakages touch: give them bases and plain lines
        recs[r] = b"@r%d\nACGTACGTAC\n+\nIIIIIIIIII\n" % r
    path = str(tmp_path / "bad.fq")
    open(path, "wb").write(b"".join(BROKEN[kind](recs)))
    code = r"""
import sys, numpy as np
import modimizer_amd as mg
L = mg.lib()
sh = mg.seqhashCreate(15, 4, 17); ms = mg.modsetCreate(sh, 20)
with mg.CFile(sys.argv[2], "w") as f:
    rc = L.mgAddSequenceFile(ms, sys.argv[1].encode(), f)
mg.check(L.modsetSyncToHost(ms, 0))
v, d, _ = mg.modset_arrays(ms)
print("rc", rc, "max", ms.contents.max, "sum", int(v[1:].sum() % (1 << 61)), int(d[1:].astype(np.int64).sum()))
"""
    res = []
    for host in ("1", "0"):
        env = dict(os.environ, MODGPU_TEXT_HOST=host, MODGPU_TEXT_WINDOW_KB="8", MODGPU_FILE_BATCH_BASES="5000", PYTHONPATH=util.ROOT)
        line = str(tmp_path / ("l%s.txt" % host))
        r = subprocess.run([sys.executable, "-c", code, path, line], capture_output=True, text=True, env=env)
        err = "\n".join(l for l in r.stderr.splitlines() if "amdgpu.ids" not in l)
        res.append((r.returncode, r.stdout, err, open(line).read() if os.path.exists(line) else ""))
    assert res[0] == res[1], res
    if kind in ("no_plus", "no_at", "short_qual"):
        assert res[0][0] != 0 and "FATAL ERROR" in res[0][2] and "line" in res[0][2]
    else:
        assert res[0][0] == 0 and "incomplete sequence record" in res[0][2]


def test_device_parser_declines_what_it_does_not_take(golden_dir, tmp_path):
    """gzip, an unterminated last line, a missing file: -2, so that mgAddSequenceFile goes to the host parser"""
    for name in ("mixed.fa.gz", "unterminated.fa", "header_last.fa"):      # (the last: a file that ends with a header line)
        rc, _ = device_records(os.path.join(golden_dir, name))
        assert rc == -2, name
    rc, _ = device_records(str(tmp_path / "nope.fa"))
    assert rc == -2


@pytest.mark.parametrize("fname", ["mixed.fa", "many.fa", "reads.fa", "mixed.fa.gz", "unterminated.fa", "mixed.fq", "header_last.fa"])
@pytest.mark.parametrize("window_kb", [0, 4])
def test_add_sequence_file_device_path_vs_host_path(fname, window_kb, golden_dir):
    """mgAddSequenceFile through the device parser (plain FASTA) or its fallback (the others) gives the modset and the "added"
    line the host parser gives; for the seqio golden files also the line the REFERENCE program printed"""
    code = r"""
import sys, os, ctypes as C, numpy as np
import modimizer_amd as mg
L = mg.lib()
sh = mg.seqhashCreate(15, 4, 17); ms = mg.modsetCreate(sh, 20)
with mg.CFile(sys.argv[2], "w") as f:
    assert L.mgAddSequenceFile(ms, sys.argv[1].encode(), f) == 0
mg.check(L.modsetSyncToHost(ms, 0))
v, d, _ = mg.modset_arrays(ms)
np.save(sys.argv[3], np.concatenate([v[1:], d[1:].astype(np.uint64)]))
"""
    import tempfile
    outs = []
    with tempfile.TemporaryDirectory() as td:
        for host in ("0", "1"):
            env = dict(os.environ, MODGPU_TEXT_HOST=host, PYTHONPATH=util.ROOT)
            if window_kb:
                env["MODGPU_TEXT_WINDOW_KB"] = str(window_kb); env["MODGPU_FILE_BATCH_BASES"] = "20000"
            line, arr = os.path.join(td, "l%s.txt" % host), os.path.join(td, "a%s.npy" % host)
            r = subprocess.run([sys.executable, "-c", code, os.path.join(golden_dir, fname), line, arr], capture_output=True, text=True, env=env)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append((open(line).read(), np.load(arr)))
    assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1])
    tag = "seqio_%s.stdout.txt" % fname.replace(".", "_")
    if os.path.exists(os.path.join(golden_dir, tag)):
        assert outs[0][0].strip() == added_line(util.golden_text(tag))


def _awkward_ids(rng, n):
    """record ids that try the id extraction (seqio.c:303-304: the header after its first byte up to the first white space): long, with
    punctuation and '>' / '@' inside, followed by a space / tab and a description, or by nothing; empty ids"""
    out = []
    alphabet = np.frombuffer(b"abcXYZ0189_:/|.#>@+-", np.uint8)
    for i in range(n):
        ln = int(rng.choice([0, 1, 5, 12, 40, 120]))
        core = alphabet[rng.integers(0, len(alphabet), ln)].tobytes() + (b"r%d" % i if rng.random() < 0.8 else b"")
        tail = [b"", b" description with spaces", b"\tafter a tab", b" > not a record", b" "][int(rng.integers(0, 5))]
        out.append((core, tail))
    return out


@pytest.mark.parametrize("fmt,crlf", [("fa", False), ("fq", False), ("fa", True), ("fq", True)])
@pytest.mark.parametrize("window_kb,batch_bases", [(0, 0), (4, 2500), (8, 40000)])
def test_query_file_device_parser_equals_host_parser(fmt, crlf, window_kb, batch_bases, golden_dir, tmp_path):
    """mgReferenceFastaRead + mgQueryFile (modmap.c:93-134,188-281) with the text parsed on the device against the same through the
    host parser (itself pinned to the reference's seqio.c): the same report, Q and M lines -- i.e. the same record ids, lengths,
    seeds -- for FASTA (wrapped lines) and FASTQ queries with awkward ids, windows of 4 KiB (ids cut by window edges) and batches
    of 2500 bases (ids carried from batch to batch)"""
    L = mg.lib()
    rng = np.random.default_rng(17 + window_kb)
    names, bases, offs = fasta.read_fasta(os.path.join(golden_dir, "ref.fa"))
    refseqs = [bases[offs[i]:offs[i + 1]] for i in range(len(names))]
    n = 700
    ids = _awkward_ids(rng, n)
    eol = b"\r\n" if crlf else b"\n"
    letters = np.frombuffer(b"ACGT", np.uint8)
    recs = []
    for i, (core, tail) in enumerate(ids):
        u = rng.random()
        if u < 0.6:
            s = refseqs[int(rng.integers(0, len(refseqs)))]
            ln = int(rng.integers(30, 2500)); a = int(rng.integers(0, max(1, len(s) - ln)))
            b = s[a:a + ln].copy()
            if rng.random() < 0.5:
                b = (3 - b[::-1]).astype(np.uint8)
        elif u < 0.9:
            b = rng.integers(0, 4, int(rng.integers(0, 600))).astype(np.uint8)
        else:
            b = np.zeros(int(rng.integers(0, 20)), np.uint8)
        txt = letters[b].tobytes()
        if fmt == "fa":
            body = b"".join(txt[j:j + 61] + eol for j in range(0, len(txt), 61))
            recs.append(b">" + core + tail + eol + body)
        else:
            recs.append(b"@" + core + tail + eol + txt + eol + b"+" + eol + b"I" * len(txt) + eol)
    qpath = str(tmp_path / ("q." + fmt))
    open(qpath, "wb").write(b"".join(recs))
    outs = []
    for host in (1, 0):
        kn = dict(TEXT_HOST=host)
        if window_kb and not host:
            kn.update(TEXT_WINDOW_KB=window_kb, FILE_BATCH_BASES=batch_bases)
        elif batch_bases:
            kn.update(FILE_BATCH_BASES=batch_bases)
        with mg.knobs(**kn):
            sh = mg.seqhashCreate(15, 8, 17); ms = mg.modsetCreate(sh, 20)
            ref = L.mgReferenceCreate(ms, 1 << 26)
            out = str(tmp_path / ("o%d.txt" % host))
            with mg.CFile(out, "w") as f:
                assert L.mgReferenceFastaRead(ref, os.path.join(golden_dir, "ref.fa").encode(), True, f) == 0
                assert L.mgQueryFile(ref, qpath.encode(), f) == 0
            outs.append(open(out, "rb").read())
            L.mgReferenceDestroy(ref); L.modsetDestroy(ms)
    assert outs[0] == outs[1], [x for x in zip(outs[0].splitlines(), outs[1].splitlines()) if x[0] != x[1]][:3]
    assert outs[0].count(b"\nQ\t") >= n - 1 and outs[0].count(b"\nM\t") > 5


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["w", "a"])
def test_query_lines_written_by_the_team_at_their_places(mode, golden_dir, tmp_path):
    """A batch of many reads is formatted by a team of threads, piece by piece, and the pieces are written in order (mg_callers.c
    queryFormatLines / queryWritePieces) around whatever the caller puts into the same FILE before and after -- a file opened for
    writing or for appending.  Against the single-thread output (MODGPU_PARSE_THREADS=1), through the device parser (a batch per
    window, formatter and writer threads behind it) and the host parser."""
    L = mg.lib()
    libc = mg._libc
    libc.fputs.argtypes = [C.c_char_p, C.c_void_p]
    rng = np.random.default_rng(99)
    names, bases, offs = fasta.read_fasta(os.path.join(golden_dir, "ref.fa"))
    refseqs = [bases[offs[i]:offs[i + 1]] for i in range(len(names))]
    letters = np.frombuffer(b"ACGT", np.uint8)
    n = 60000
    recs = []
    for i in range(n):
        s = refseqs[i % len(refseqs)]
        ln = int(rng.integers(40, 400)) if i % 50 else int(rng.integers(1500, 4000))
        a = int(rng.integers(0, max(1, len(s) - ln)))
        txt = letters[s[a:a + ln]].tobytes()
        recs.append(b"@q%d some text\n" % i + txt + b"\n+\n" + b"I" * len(txt) + b"\n")
    qpath = str(tmp_path / "q.fq")
    open(qpath, "wb").write(b"".join(recs))
    outs = {}
    for tag, kn in (("one", dict(PARSE_THREADS=1, TEXT_HOST=1)), ("team_host", dict(TEXT_HOST=1)), ("team_device", dict(TEXT_WINDOW_KB=2048)),
                    ("team_device_1", dict())):
        with mg.knobs(**kn):
            sh = mg.seqhashCreate(15, 8, 17); ms = mg.modsetCreate(sh, 20)
            ref = L.mgReferenceCreate(ms, 1 << 26)
            out = str(tmp_path / (tag + ".txt"))
            open(out, "wb").write(b"already there\n")
            with mg.CFile(out, "a" if mode == "a" else "r+") as f:
                if mode != "a":
                    libc.fseek.argtypes = [C.c_void_p, C.c_long, C.c_int]
                    libc.fseek(f, 0, 2)
                libc.fputs(b"before the reference\n", f)
                assert L.mgReferenceFastaRead(ref, os.path.join(golden_dir, "ref.fa").encode(), True, f) == 0
                libc.fputs(b"before the queries\n", f)
                assert L.mgQueryFile(ref, qpath.encode(), f) == 0
                libc.fputs(b"after the queries\n", f)
            outs[tag] = open(out, "rb").read()
            L.mgReferenceDestroy(ref); L.modsetDestroy(ms)
    one = outs["one"]
    assert one.startswith(b"already there\nbefore the reference\n") and one.endswith(b"after the queries\n")
    assert one.count(b"\nQ\t") == n and one.count(b"\nM\t") > 100
    for tag, o in outs.items():
        assert o == one, (tag, len(o), len(one))
