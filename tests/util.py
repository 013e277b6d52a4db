"""Shared helpers for the parity tests (checker side: may use oracle/)."""
import hashlib
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

_scan = None


def scan_vectors():
    global _scan
    if _scan is None:
        _scan = np.load(os.path.join(GOLDEN, "scan_vectors.npz"))
    return _scan


def scan_configs():
    v = scan_vectors()
    return [tuple(int(x) for x in row) for row in v["configs"]]


def scan_cases(ci):
    """yield (name, bases, kmer, pos, isF) for config ci"""
    v = scan_vectors()
    for name in v["c%d_names" % ci]:
        name = str(name)
        yield (name, v["c%d_%s_in" % (ci, name)], v["c%d_%s_kmer" % (ci, name)],
               v["c%d_%s_pos" % (ci, name)], v["c%d_%s_isf" % (ci, name)])


def minimizer_case(ci, name):
    v = scan_vectors()
    key = "c%d_%s_min_hash" % (ci, name)
    if key not in v.files:
        return None
    return v[key], v["c%d_%s_min_pos" % (ci, name)], v["c%d_%s_min_isf" % (ci, name)]


def golden_text(name):
    return open(os.path.join(GOLDEN, name)).read()


def check_dump(text, golden_name):
    """compare a -wt dump with the golden full text, or with its digest (sha256 + head/tail)"""
    full = os.path.join(GOLDEN, golden_name)
    if os.path.exists(full):
        assert text == open(full).read()
        return
    dig = json.load(open(full.replace(".txt", ".digest.json")))
    lines = text.splitlines()
    assert len(lines) == dig["lines"]
    assert lines[:40] == dig["head"] and lines[-5:] == dig["tail"]
    assert hashlib.sha256(text.encode()).hexdigest() == dig["sha256"]


def concat_reads(reads):
    """list of uint8 arrays -> (bases, offsets int64)"""
    offs = np.zeros(len(reads) + 1, np.int64)
    if reads:
        offs[1:] = np.cumsum([len(r) for r in reads])
    bases = np.concatenate(reads) if reads else np.zeros(0, np.uint8)
    return bases.astype(np.uint8), offs


def oracle_scan_batch(hasher, bases, offs):
    ks, ps, fs, st = [], [], [], [0]
    for r in range(len(offs) - 1):
        a, b, c = hasher.scan(bases[offs[r]:offs[r + 1]])
        ks.append(a); ps.append(b); fs.append(c); st.append(st[-1] + len(a))
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
    return cat(ks, np.uint64), cat(ps, np.int32), cat(fs, np.uint8), np.array(st, np.int64)


MODUTILS_TAGS = {"k21d64": (20, 21, 64, 17), "k31d4": (22, 31, 4, 17), "k19d31": (20, 19, 31, 17)}
MODMAP_TAGS = {"k21d64": (21, 64), "k15d8": (15, 8), "k19d31": (19, 31)}


def bench_lines(stdout):
    """bench.py's stdout -> (the driver's line, the whole result): the LAST line is the compact one the driver parses,
    an earlier `BENCH_DETAIL {...}` line carries everything"""
    lines = stdout.splitlines()
    compact = [l for l in lines if l.startswith("{")]
    detail = [l for l in lines if l.startswith("BENCH_DETAIL ")]
    return (json.loads(compact[-1]) if compact else None, json.loads(detail[-1][len("BENCH_DETAIL "):]) if detail else None, compact)
