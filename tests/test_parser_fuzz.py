"""The parser loop closed against the REFERENCE: 640 random FASTA / FASTQ shapes (tests/parser_fuzz.py) whose
outcome under the reference program -- `modutils_ref -c 24 k w 17 -a <file>`: the "added N sequences total length L total hashes H, new
max M" line, the exit code, the messages -- is the fixture tests/golden/parser_fuzz.json (tests/golden/make_parser_fuzz.py).

  not gpu   the HOST parser (mg_seqio.c) on every trial: N and L from its records, H and M from the oracle's scan of them (the oracle is
            the checker), "incomplete sequence record line n" on stderr, and the fatal cases in a process of their own;
  gpu       mgAddSequenceFile on every trial through the DEVICE parser (mg_textgpu.hip) and through the host parser: the line it prints
            is the reference's, byte for byte; the fatal cases die with the reference's message.
Before this the device parser was only ever compared with the host parser (tests/test_gpu_text.py), and the host parser with the
reference on five files (tests/test_seqio.py): a shared misreading of an odd shape would have passed."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import modimizer_amd as mg
from oracle import pyoracle as po
from tests import parser_fuzz as pf, util
from tests.test_seqio import parse_file

FIX = json.load(open(os.path.join(util.ROOT, "tests", "golden", "parser_fuzz.json")))["trials"]


def fatal(t):
    return [e for e in t["err"] if e.startswith("FATAL ERROR")]


def write_trial(t, tmp_path):
    kind, text = pf.make_text(t["seed"])
    assert kind == t["kind"] and pf.digest(text) == t["sha1"], "tests/parser_fuzz.py no longer makes the text the fixture was made from"
    path = str(tmp_path / ("t%d.%s" % (t["seed"], "fq" if kind == "fastq" else "fa")))
    open(path, "wb").write(text)
    return path


def test_fixture_covers_the_shapes():
    assert len(FIX) >= 500
    kinds = {k: sum(t["kind"] == k for t in FIX) for k in ("fasta", "fastq")}
    assert min(kinds.values()) > 150
    msgs = " ".join(e for t in FIX for e in t["err"])
    for m in ("incomplete sequence record", "missing + FASTQ line", "qual not same length as seq", "no initial @ for FASTQ", "failed to open sequence file"):
        assert m in msgs, m


def test_host_parser_equals_the_reference_program(tmp_path, capfd):
    L = mg.lib()
    checked = 0
    for t in FIX:
        if t["rc"] != 0 or t["added"] is None:
            continue
        path = write_trial(t, tmp_path)
        capfd.readouterr()
        names, seqs = parse_file(path, 1 << 40, 1 + t["seed"] % 4)
        err = capfd.readouterr().err
        n, tot = len(seqs), sum(len(s) for s in seqs)
        oh = po.Hasher(t["k"], t["w"], 17)
        if tot:
            bases = np.concatenate(seqs) if seqs else np.zeros(0, np.uint8)
            if t["kind"] == "fastq":
                bases = bases & 3                              # (ACGTN only in these files: nothing is kept as -2)
            offs = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.int64)
            km = util.oracle_scan_batch(oh, bases, offs)[0]
        else:
            km = np.zeros(0, np.uint64)
        line = "added %d sequences total length %d total hashes %d, new max %d" % (n, tot, len(km), len(np.unique(km)))
        assert line == t["added"], (t["seed"], t["kind"])
        want_inc = [e for e in t["err"] if e.startswith("incomplete")]
        got_inc = [l for l in err.splitlines() if l.startswith("incomplete")]
        assert got_inc == want_inc, (t["seed"], got_inc, want_inc)
        os.remove(path)
        checked += 1
    assert checked > 450


def test_host_parser_fails_as_the_reference_program_fails(tmp_path):
    """what the reference refuses: text that is neither FASTA nor FASTQ (seqIOopenRead gives up: modutils says "failed to open sequence
    file") is not opened; a broken FASTQ record ends the process with the reference's message and code"""
    L = mg.lib()
    n_open = n_die = 0
    for t in FIX:
        f = fatal(t)
        if not f:
            continue
        path = write_trial(t, tmp_path)
        if "failed to open sequence file" in f[0]:
            assert not L.mgSeqOpen(path.encode()), t["seed"]
            n_open += 1
        elif "hashTableSize" not in f[0]:
            code = ("import ctypes as C, modimizer_amd as mg; L = mg.lib(); r = L.mgSeqOpen(%r.encode()); b = mg.MgSeqBatch()\n"
                    "while L.mgSeqNextBatch(r, 1 << 30, C.byref(b)): L.mgSeqBatchFree(C.byref(b))" % path)
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=util.ROOT, env=dict(os.environ, MODGPU_NO_TORCH="1"))
            assert r.returncode == 255 and f[0] in r.stderr, (t["seed"], f[0], r.stderr[-300:])
            n_die += 1
        os.remove(path)
    assert n_open > 10 and n_die > 20


@pytest.mark.gpu
@pytest.mark.parametrize("parser", ["device", "host"])
def test_add_sequence_file_prints_what_the_reference_prints(parser, tmp_path, capfd):
    L = mg.lib()
    out = str(tmp_path / "o.txt")
    checked = 0
    with mg.knobs(TEXT_HOST="1" if parser == "host" else "0"):
        for t in FIX:
            if t["rc"] != 0 or t["added"] is None:
                continue
            path = write_trial(t, tmp_path)
            sh = mg.seqhashCreate(t["k"], t["w"], 17)
            ms = mg.modsetCreate(sh, t["bits"])
            capfd.readouterr()
            with mg.CFile(out, "w") as f:
                rc = L.mgAddSequenceFile(ms, path.encode(), f)
            err = capfd.readouterr().err
            assert rc == 0 and open(out).read().strip() == t["added"], (t["seed"], t["kind"], parser)
            assert [l for l in err.splitlines() if l.startswith("incomplete")] == [e for e in t["err"] if e.startswith("incomplete")], t["seed"]
            L.modsetDestroy(ms); L.mgSeqhashDestroy(sh)
            os.remove(path)
            checked += 1
    assert checked > 450


@pytest.mark.gpu
def test_add_sequence_file_fails_as_the_reference_program_fails(tmp_path):
    L = mg.lib()
    n = 0
    for t in FIX:
        f = fatal(t)
        if not f or "hashTableSize" in f[0]:
            continue
        path = write_trial(t, tmp_path)
        if "failed to open sequence file" in f[0]:
            sh = mg.seqhashCreate(t["k"], t["w"], 17); ms = mg.modsetCreate(sh, 20)
            with mg.CFile(os.devnull, "w") as fo:
                assert L.mgAddSequenceFile(ms, path.encode(), fo) != 0, t["seed"]          # (modutils then says "failed to open sequence file")
            L.modsetDestroy(ms)
        else:
            code = ("import ctypes as C, modimizer_amd as mg; L = mg.lib(); sh = mg.seqhashCreate(%d, %d, 17); ms = mg.modsetCreate(sh, 22)\n"
                    "with mg.CFile('/dev/null', 'w') as f: L.mgAddSequenceFile(ms, %r.encode(), f)" % (t["k"], t["w"], path))
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=util.ROOT,
                               env=dict(os.environ, MODGPU_NO_TORCH="1", MODGPU_TEXT_HOST="0"))
            assert r.returncode == 255 and f[0] in r.stderr, (t["seed"], f[0], r.stderr[-300:])
        os.remove(path)
        n += 1
    assert n > 40
