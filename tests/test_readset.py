"""modasm's read ingest (readsetFileRead + invBuild + readsetStats + the .readset file): oracle and
GPU path against the reference program's own output (tests/golden/asm_*: modutils -c 20 k w 17 -a reads.fa
-s 2 3 5 -w src.mod; modasm -m src.mod -f reads2.fa -S -w stem)."""
import ctypes as C
import gzip
import os

import numpy as np
import pytest

import modimizer_amd as mg
from modimizer_amd import fasta
from tests import util

TAGS = {"k21d16": (21, 16), "k17d31": (17, 31)}


def readset_mask(raw):
    """.readset bytes with the in-memory addresses (Array.base, Read.hit, Read.dx) zeroed"""
    b = bytearray(raw)
    at = 16
    b[at + 8:at + 16] = bytes(8)
    dim = int.from_bytes(b[at + 16:at + 20], "little")
    size = int.from_bytes(b[at + 20:at + 24], "little")
    assert size == 72
    at += 32
    for i in range(dim):
        b[at + 72 * i + 8:at + 72 * i + 24] = bytes(16)
    return bytes(b)


def mod_mask(raw, bits=20):
    b = bytearray(raw)
    at = 8 + 4 + 4 + 8 + 80 + 4 * (1 << bits)
    b[at:at + 8] = bytes(8)                          # value[0]: never initialised by the reference
    return bytes(b)


def rs_lines(text):
    return [l for l in text.splitlines() if l.startswith("RS ")]


@pytest.mark.parametrize("tag", list(TAGS))
def test_oracle_readset_vs_reference_program(tag, golden_dir, tmp_path):
    from oracle import pyoracle as orc
    k, w = TAGS[tag]
    h = orc.Hasher(k, w, 17)
    ms = orc.Modset(h, 20)
    for s in fasta.read_fasta_list(os.path.join(golden_dir, "reads.fa")):
        ms.add_sequence(s)
    ms.set_copy(2, 3, 5)
    p = str(tmp_path / "src.mod"); ms.write_mod(p)
    assert mod_mask(open(p, "rb").read()) == mod_mask(gzip.open(os.path.join(golden_dir, "asm_%s_src.mod" % tag)).read())
    rs = orc.Readset(ms)
    rs.read(fasta.read_fasta_list(os.path.join(golden_dir, "reads2.fa")))
    assert rs_lines(rs.stats_text(str(tmp_path / "s.txt"))) == rs_lines(util.golden_text("asm_%s.stdout.txt" % tag))
    out = str(tmp_path / "o.readset"); rs.write(out)
    assert readset_mask(open(out, "rb").read()) == readset_mask(gzip.open(os.path.join(golden_dir, "asm_%s.readset" % tag)).read())
    ms.write_mod(p)                                  # depth rebuilt from the reads (modasm.c:158,174)
    assert mod_mask(open(p, "rb").read()) == mod_mask(gzip.open(os.path.join(golden_dir, "asm_%s.mod" % tag)).read())
    a = rs.arrays()
    # inverse lists: per mod, the reads that hit it, in read order, one entry per hit
    d = ms.depths()
    assert int(a["invStart"][-1]) == int(d[(d > 0) & (d < 65535)].sum()) == a["totHit"]
    for i in np.flatnonzero(d[1:] > 0)[:200] + 1:
        got = a["invSpace"][int(a["invStart"][i]):int(a["invStart"][i]) + int(d[i])]
        want = [r + 1 for r in range(len(a["nHit"])) for hh in a["hit"][int(a["hitStart"][r]):int(a["hitStart"][r + 1])]
                if (int(hh) & 0x7fffffff) == i]
        assert list(got) == want
    rs.close(); ms.close()


# ---- the library (mg_readset.c) ----

def lib_arrays(rs):
    r = rs.contents
    n, tot, m = r.nReads, int(r.totHit), r.ms.contents.max
    as_np = lambda p, k, dt, off=0: np.ctypeslib.as_array(p, (max(k + off, 1),))[off:k + off].astype(dt).copy()
    return {"len": as_np(r.len, n, np.int64, 1), "nHit": as_np(r.nHit, n, np.int64, 1), "nMiss": as_np(r.nMiss, n, np.int64, 1),
            "nCopy": np.array([list(r.nCopy[i]) for i in range(1, n + 1)]).reshape(n, 4),
            "hitStart": as_np(r.hitStart, n + 1, np.uint64, 1), "hit": as_np(r.hit, tot, np.uint32),
            "dx": as_np(r.dx, tot, np.uint16), "totHit": tot,
            "invStart": as_np(r.invStart, m + 2, np.uint64), "invSpace": as_np(r.invSpace, int(r.invStart[m + 1]), np.uint32)}


def same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


def stats_text(rs, tmp):
    with mg.CFile(tmp, "w") as f:
        mg.lib().mgReadsetStats(rs, f)
    return open(tmp).read()


@pytest.mark.parametrize("tag", list(TAGS))
def test_load_write_readset_files(tag, golden_dir, tmp_path):
    """the reference's own <stem>.mod + <stem>.readset: load, print the stats, write back the same bytes"""
    L = mg.lib()
    stem = os.path.join(golden_dir, "asm_%s" % tag)
    rs = L.mgReadsetLoad(stem.encode())
    assert rs.contents.nReads == 82
    got = stats_text(rs, str(tmp_path / "s.txt")).splitlines()
    assert got == util.golden_text("asm_%s.stdout.txt" % tag).splitlines()[-len(got):] and len(got) == 9
    out = str(tmp_path / "again")
    L.mgReadsetWrite(rs, out.encode())
    assert gzip.open(out + ".mod").read() == gzip.open(stem + ".mod").read()
    assert readset_mask(gzip.open(out + ".readset").read()) == readset_mask(gzip.open(stem + ".readset").read())
    rs2 = L.mgReadsetLoad(out.encode())
    same(lib_arrays(rs), lib_arrays(rs2))
    L.mgReadsetDestroy(rs); L.mgReadsetDestroy(rs2)


def load_mod_gz(path, tmp):
    open(tmp, "wb").write(gzip.open(path).read())
    with mg.CFile(tmp, "r") as f:
        return mg.lib().modsetRead(f)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(TAGS))
def test_readset_file_read_gpu(tag, golden_dir, tmp_path):
    """modasm -m src.mod -f reads2.fa -S -w stem with the scan + lookups on the GPU: same statistics,
    same .readset and .mod bytes (depth rebuilt from the reads)"""
    L = mg.lib()
    ms = load_mod_gz(os.path.join(golden_dir, "asm_%s_src.mod" % tag), str(tmp_path / "src.mod"))
    rs = L.mgReadsetCreate(ms)
    os.environ["MODGPU_FILE_BATCH_MBP"] = "1"; L.mgReloadKnobs()        # several batches: reads2.fa holds 120 kb... one batch; the knob is exercised below
    try:
        assert L.mgReadsetFileRead(rs, os.path.join(golden_dir, "reads2.fa").encode()) == 0
    finally:
        del os.environ["MODGPU_FILE_BATCH_MBP"]; L.mgReloadKnobs()
    got = stats_text(rs, str(tmp_path / "s.txt")).splitlines()
    assert got == util.golden_text("asm_%s.stdout.txt" % tag).splitlines()[-len(got):]
    out = str(tmp_path / "gpu")
    L.mgReadsetWrite(rs, out.encode())
    stem = os.path.join(golden_dir, "asm_%s" % tag)
    assert mod_mask(gzip.open(out + ".mod").read()) == mod_mask(gzip.open(stem + ".mod").read())
    assert readset_mask(gzip.open(out + ".readset").read()) == readset_mask(gzip.open(stem + ".readset").read())
    L.mgReadsetDestroy(rs)


@pytest.mark.gpu
def test_readset_vs_oracle_saturation_and_batches(tmp_path):
    """a k-mer hit more than 65535 times (depth saturates, no inverse list: modasm.c:174,266,278), empty
    and short reads, reads on both strands; the same reads in one call and through a file in batches"""
    from oracle import pyoracle as orc
    rng = np.random.default_rng(11)
    k, w = 15, 1
    g = rng.integers(0, 4, 30000).astype(np.uint8)
    reads = [np.zeros(70000, np.uint8), g[:9000], (3 - g[2000:12000][::-1]).astype(np.uint8), np.zeros(0, np.uint8),
             g[100:110], rng.integers(0, 4, 5000).astype(np.uint8), np.zeros(3000, np.uint8), g[20000:30000]]
    h = orc.Hasher(k, w, 17); oms = orc.Modset(h, 20)
    for s in (reads[0][:100], g):
        oms.add_sequence(s)
    oms.set_copy(1, 2, 3)
    ors = orc.Readset(oms); ors.read(reads)
    want = ors.arrays()
    assert int(oms.depths().max()) == 65535
    # the same modset for the library: through a .mod file
    p = str(tmp_path / "m.mod"); oms.write_mod(p)
    L = mg.lib()
    for mode in ("memory", "file"):
        with mg.CFile(p, "r") as f:
            ms = L.modsetRead(f)
        rs = L.mgReadsetCreate(ms)
        if mode == "memory":
            bases, offs = util.concat_reads(reads)
            assert L.mgReadsetRead(rs, bases.ctypes.data, offs.ctypes.data, len(reads)) == 0
        else:
            fa = str(tmp_path / "r.fa")
            with open(fa, "w") as f:
                for i, s in enumerate(reads):
                    f.write(">r%d\n%s\n" % (i, "".join("ACGT"[b] for b in s)))
            os.environ["MODGPU_FILE_BATCH_MBP"] = "1"
            os.environ["MODGPU_FILE_BATCH_BASES"] = "9500"; L.mgReloadKnobs()
            try:
                assert L.mgReadsetFileRead(rs, fa.encode()) == 0
            finally:
                del os.environ["MODGPU_FILE_BATCH_MBP"]; del os.environ["MODGPU_FILE_BATCH_BASES"]; L.mgReloadKnobs()
        same(want, lib_arrays(rs))
        assert np.array_equal(np.ctypeslib.as_array(ms.contents.depth, (ms.contents.max + 1,)), oms.depths())
        assert rs_lines(stats_text(rs, str(tmp_path / "s.txt"))) == rs_lines(ors.stats_text(str(tmp_path / "o.txt")))
        L.mgReadsetDestroy(rs)
    ors.close(); oms.close()
