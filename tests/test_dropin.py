"""The drop-in boundary proven with the reference's own callers.

oracle/Makefile (test infrastructure, built only where /root/reference exists; the binaries travel to the
GPU box in oracle/_ref/) links the reference's UNMODIFIED modutils.c and modmap.c — with its own seqio.c,
utils.c, array.c, dict.c, hash.c — against libmodgpu.so in the place of seqhash.c + modset.c:
  _ref/modutils_dropin, _ref/modmap_dropin
and the reference's modutils.c with examples/modutils_batch.patch applied on the fly (the per-read loop of
modutils.c:41-45 replaced by one mgAddSequenceBatch call):
  _ref/modutils_batch
The GPU tests run them on the golden inputs and compare with what the reference program itself printed
(tests/golden/*.stdout.txt and the -wt / -H files, made by tests/golden/make_golden.py).
"""
import os
import subprocess
import sys

import pytest

from tests import util

REFDIR = os.path.join(util.ROOT, "oracle", "_ref")
INC = os.path.join(util.ROOT, "include")
REFSRC = "/root/reference"


def strip_timing(text):
    return "\n".join(l for l in text.splitlines() if not l.startswith("user\t") and "resources used" not in l) + "\n"


def need(exe):
    p = os.path.join(REFDIR, exe)
    if not os.path.exists(p):
        pytest.skip("oracle/_ref/%s not built (needs the reference tree at build time)" % exe)
    return p


# ---- headers (CPU) -------------------------------------------------------------------------------

COMPAT_USER = r"""
#include "modgpu.h"
int use (Modset *ms, Seqhash *sh, SeqhashRCiterator *si)
{ msSetCopy1 (ms, 1); msSetCopyM (ms, 2); msSetMinor (ms, 3); msSetRepeat (ms, 3); msSetInternal (ms, 3); msSetRDNA (ms, 3);
  msSetCopy0 (ms, 4); msSetCopy2 (ms, 5);
  int c = msCopy (ms, 1) + msIsCopy0 (ms, 1) + msIsCopy1 (ms, 1) + msIsCopy2 (ms, 1) + msIsCopyM (ms, 1)
          + msIsMinor (ms, 3) + msIsRepeat (ms, 3) + msIsInternal (ms, 3) + msIsRDNA (ms, 3) + MS_MINOR + MS_REPEAT + MS_INTERNAL + MS_RDNA;
  char *s = seqhashString (sh, seqhash (sh, 5)); (void) s;
  seqhashRCiteratorDestroy (si); seqhashDestroy (sh);
  return c + (int) mgModsetDeviceSlots (ms);
}
"""


def test_compat_header_carries_the_header_inline_api(tmp_path):
    """without the reference tree: modgpu.h brings modgpu_compat.h, incl. msSet*/msIs*/msCopy, MS_*, seqhashString,
    seqhashDestroy, seqhashRCiteratorDestroy (modset.h:49-69, seqhash.h:37,54-60); plain C99, warnings on"""
    src = tmp_path / "u.c"
    src.write_text(COMPAT_USER)
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", INC, "-c", str(src), "-o", str(tmp_path / "u.o")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.skipif(not os.path.isdir(REFSRC), reason="reference tree absent")
def test_layer2_header_compiles_after_the_reference_headers(tmp_path):
    """inside the reference tree: #include "modset.h" (the reference's) and then "modgpu.h" in one translation unit"""
    src = tmp_path / "v.c"
    src.write_text('#include "modset.h"\n#include "seqio.h"\n#include "modgpu.h"\n'
                   'I64 f (Modset *ms, char *b, int64_t *o, int n) { msSetCopy1 (ms, 1); return mgAddSequenceBatch (ms, b, o, n) + (I64) sizeof (MgReference); }\n')
    r = subprocess.run(["gcc", "-std=gnu11", "-w", "-I", REFSRC, "-I", INC, "-c", str(src), "-o", str(tmp_path / "v.o")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # and the struct layouts the two sides see are the same
    probe = tmp_path / "p.c"
    probe.write_text('#include <stdio.h>\n#include <stddef.h>\n#ifdef REF\n#include "modset.h"\n#else\n#include "modgpu.h"\n#endif\n'
                     'int main (void) { printf ("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof (Seqhash), sizeof (SeqhashRCiterator), sizeof (Modset), '
                     'offsetof (Seqhash, factor1), offsetof (Seqhash, patternRC), offsetof (SeqhashRCiterator, hashBuf), offsetof (Modset, index), offsetof (Modset, max)); return 0; }\n')
    outs = []
    for flags in (["-DREF", "-I", REFSRC], ["-I", INC]):
        exe = str(tmp_path / ("p" + str(len(outs))))
        r = subprocess.run(["gcc", "-std=gnu11", "-w"] + flags + [str(probe), "-o", exe], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        outs.append(subprocess.run([exe], capture_output=True, text=True).stdout)
    assert outs[0] == outs[1] and outs[0].split()[:3] == ["80", "72", "72"]


def test_batch_patch_is_small_and_applies(tmp_path):
    patch = os.path.join(util.ROOT, "examples", "modutils_batch.patch")
    lines = open(patch).read().splitlines()
    added = [l for l in lines if l.startswith("+") and not l.startswith("+++")]
    assert len(added) <= 24
    if os.path.isdir(REFSRC):
        r = subprocess.run(["patch", "-s", "--dry-run", "-o", os.devnull, os.path.join(REFSRC, "modutils.c"), patch], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr


# ---- the reference's own programs on libmodgpu.so (GPU) -------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("exe", ["modutils_dropin", "modutils_batch"])
@pytest.mark.parametrize("tag", ["k21d64", "k19d31"])
def test_reference_modutils_runs_on_the_library(exe, tag, golden_dir, tmp_path):
    """modutils -c B k w s -a reads.fa -a reads2.fa -wt dump -H hist -p 2 40 -H hist2 -wt dump2: the reference's main()
    and option loop, its seqio, its inlined msSet*/seqhashRCiteratorDestroy — scan and modset from libmodgpu.so"""
    B, k, w, s = util.MODUTILS_TAGS[tag]
    prog = need(exe)
    dump, hist, pdump, phist = (str(tmp_path / n) for n in ("d.txt", "h.txt", "pd.txt", "ph.txt"))
    r = subprocess.run([prog, "-c", str(B), str(k), str(w), str(s), "-a", os.path.join(golden_dir, "reads.fa"),
                        "-a", os.path.join(golden_dir, "reads2.fa"), "-wt", dump, "-H", hist, "-p", "2", "40", "-H", phist, "-wt", pdump],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert strip_timing(r.stdout) == util.golden_text("modutils_%s.stdout.txt" % tag)
    assert open(hist).read() == util.golden_text("modutils_%s.hist.txt" % tag)
    assert open(phist).read() == util.golden_text("modutils_%s.pruned_hist.txt" % tag)
    util.check_dump(open(dump).read(), "modutils_%s.dump.txt" % tag)
    util.check_dump(open(pdump).read(), "modutils_%s.pruned_dump.txt" % tag)


@pytest.mark.gpu
def test_reference_modutils_writes_and_reads_mod_files_on_the_library(golden_dir, tmp_path):
    """-w / -r / -m through the reference's fzopen + the library's modsetWrite / modsetRead / modsetMerge"""
    prog = need("modutils_dropin")
    a, b = str(tmp_path / "a.mod"), str(tmp_path / "b.mod")
    args = ["-c", "20", "21", "64", "17"]
    r1 = subprocess.run([prog] + args + ["-a", os.path.join(golden_dir, "reads.fa"), "-w", a], capture_output=True, text=True, timeout=900)
    r2 = subprocess.run([prog] + args + ["-a", os.path.join(golden_dir, "reads2.fa"), "-w", b], capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0 and r2.returncode == 0, r1.stderr + r2.stderr
    import gzip
    open(b + ".raw", "wb").write(gzip.open(b).read())              # -m reads with fopen, not fzopen (modutils.c:231)
    dump = str(tmp_path / "m.txt")
    r3 = subprocess.run([prog, "-r", a, "-m", b + ".raw", "-wt", dump], capture_output=True, text=True, timeout=900)
    assert r3.returncode == 0, r3.stderr
    # merging the two files' sets = adding the second file after the first (same first-occurrence order, depths add)
    assert open(dump).read() == util.golden_text("modutils_k21d64.dump.txt")
    assert "number of entries 2144 total count 6439" in r3.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["k21d64", "k15d8", "k19d31"])
def test_reference_modmap_runs_on_the_library(tag, golden_dir):
    """modmap -f ref.fa -q queries.fa: referenceFastaRead, referencePack and queryProcess are the reference's own code;
    every modRCiterator / modRCnext / modsetIndexFind / modsetPack under them is the library's"""
    k, w = util.MODMAP_TAGS[tag]
    prog = need("modmap_dropin")
    r = subprocess.run([prog, "-K", str(k), "-W", str(w), "-S", "17", "-B", "20", "-f", os.path.join(golden_dir, "ref.fa"),
                        "-q", os.path.join(golden_dir, "queries.fa")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert strip_timing(r.stdout) == util.golden_text("modmap_%s.stdout.txt" % tag)


@pytest.mark.gpu
def test_reference_modmap_reads_a_reference_the_library_wrote(golden_dir, tmp_path):
    """modmap -w on the library, then the same program -r: .mod/.ref written through the library's modsetWrite"""
    prog = need("modmap_dropin")
    stem = str(tmp_path / "stem")
    base = ["-K", "21", "-W", "64", "-S", "17", "-B", "20"]
    r = subprocess.run([prog] + base + ["-f", os.path.join(golden_dir, "ref.fa"), "-w", stem], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([prog, "-r", stem, "-q", os.path.join(golden_dir, "queries.fa")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    assert strip_timing(r.stdout) == util.golden_text("modmap_k21d64_files.stdout.txt")


# ---- modmap -v (modmap.c:218-229) ------------------------------------------------------------------

VERBOSE_CODE = r"""
import ctypes as C, os, sys
import modimizer_amd as mg
from modimizer_amd import fasta
L = mg.lib()
k, w, gd, host = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4] == "files"
out = C.c_void_p.in_dll(C.CDLL(None), "stdout")
sh = mg.seqhashCreate(k, w, 17); ms = mg.modsetCreate(sh, 20)
ref = L.mgReferenceCreate(ms, 1 << 26)
L.mgSetVerbose(1)
if host:
    assert L.mgReferenceFastaRead(ref, os.path.join(gd, "ref.fa").encode(), True, out) == 0
    assert L.mgQueryFile(ref, os.path.join(gd, "queries.fa").encode(), out) == 0
else:
    def load(p):
        names, b, o = fasta.read_fasta(p)
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        return b, o, arr, len(names)
    b, o, arr, n = load(os.path.join(gd, "ref.fa"))
    assert L.mgReferenceRead(ref, b.ctypes.data, o.ctypes.data, n, arr, True, out) == 0
    b, o, arr, n = load(os.path.join(gd, "queries.fa"))
    assert L.mgQueryProcess(ref, b.ctypes.data, o.ctypes.data, n, arr, out) == 0
C.CDLL(None).fflush(None)
"""


@pytest.mark.gpu
@pytest.mark.parametrize("tag,mode", [("k21d64", "arrays"), ("k15d8", "files")])
def test_verbose_seed_lines_vs_reference(tag, mode, golden_dir):
    """mgSetVerbose(1): the per-seed lines, interleaved with the Q / M lines exactly as `modmap -v` prints them"""
    k, w = util.MODMAP_TAGS[tag]
    r = subprocess.run([sys.executable, "-c", VERBOSE_CODE, str(k), str(w), golden_dir, mode], capture_output=True, text=True,
                       cwd=util.ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    want = util.golden_text("modmap_%s.verbose.stdout.txt" % tag).splitlines()[1:]      # without the "initialised" line
    assert r.stdout.splitlines() == want


@pytest.mark.gpu
def test_reference_modmap_verbose_on_the_library(golden_dir):
    prog = need("modmap_dropin")
    r = subprocess.run([prog, "-K", "21", "-W", "64", "-S", "17", "-B", "20", "-v", "-f", os.path.join(golden_dir, "ref.fa"),
                        "-q", os.path.join(golden_dir, "queries.fa")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert strip_timing(r.stdout) == util.golden_text("modmap_k21d64.verbose.stdout.txt")
