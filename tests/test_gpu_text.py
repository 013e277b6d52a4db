"""The FASTA front end ON THE DEVICE (mg_textgpu.hip) against the host parser (mg_seqio.c, itself pinned to the reference's
seqio.c by tests/test_seqio.py) and against what the reference program prints for the golden text files.

mgAddSequenceFile takes the device parser by itself for plain FASTA text; MODGPU_TEXT_HOST=1 forces the host parser,
MODGPU_TEXT_WINDOW_KB / MODGPU_FILE_BATCH_BASES put window and batch edges everywhere (knobs are read per call)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import modimizer_amd as mg
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


@pytest.mark.parametrize("kind,n_rec,seed", [("mixed", 300, 1), ("tiny", 5000, 2), ("long", 40, 3), ("mixed", 3, 4)])
@pytest.mark.parametrize("window_kb,batch_bases", [(0, 0), (4, 0), (8, 3000), (64, 100000)])
def test_device_parser_equals_host_parser(kind, n_rec, seed, window_kb, batch_bases, tmp_path):
    path = str(tmp_path / "t.fa")
    open(path, "wb").write(crafted_fasta(np.random.default_rng(seed), n_rec, kind))
    _, want = parse_file(path, 1 << 40, 4)
    env = {}
    if window_kb:
        env["MODGPU_TEXT_WINDOW_KB"] = str(window_kb)
    if batch_bases:
        env["MODGPU_FILE_BATCH_BASES"] = str(batch_bases)
    os.environ.update(env)
    try:
        rc, got = device_records(path)
    finally:
        for k in env:
            del os.environ[k]
    assert rc == 0, mg.lib().mgLastError()
    assert len(got) == len(want)
    for i, (a, b) in enumerate(zip(got, want)):
        assert np.array_equal(a, b), (i, len(a), len(b))


def test_device_parser_declines_what_it_does_not_take(golden_dir, tmp_path):
    """gzip, FASTQ, an unterminated last line, a missing file: -2, so that mgAddSequenceFile goes to the host parser"""
    for name in ("mixed.fa.gz", "mixed.fq", "unterminated.fa"):
        rc, _ = device_records(os.path.join(golden_dir, name))
        assert rc == -2, name
    rc, _ = device_records(str(tmp_path / "nope.fa"))
    assert rc == -2


@pytest.mark.parametrize("fname", ["mixed.fa", "many.fa", "reads.fa", "mixed.fa.gz", "unterminated.fa", "mixed.fq"])
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
