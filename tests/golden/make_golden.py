#!/usr/bin/env python3
"""Generate tests/golden/* by running the REFERENCE ITSELF (compiled from /root/reference into
oracle/_ref/ by oracle/Makefile) in this container.  The outputs are data: inputs and the
reference's answers.  Re-run with:  python tests/golden/make_golden.py
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po          # noqa: E402
from modimizer_amd import fasta, synth     # noqa: E402

assert po.have_ref(), "needs oracle/_ref (only buildable where /root/reference exists)"
R = po.ref()
REFDIR = os.path.join(ROOT, "oracle", "_ref")


def edge_reads(k, rng):
    """name -> bases; each case exists because the cited reference line makes it special."""
    pal = np.array([0, 1, 2, 3] * 40, np.uint8)                 # ACGT repeats: even-k palindromes, hashF==hashR ties (seqhash.c:66-67)
    reads = {
        "empty": np.zeros(0, np.uint8),
        "len1": np.array([2], np.uint8),
        "len_k_minus_1": rng.integers(0, 4, max(k - 1, 0)).astype(np.uint8),   # seqhash.c:162
        "len_k": rng.integers(0, 4, k).astype(np.uint8),                       # exactly one k-mer
        "len_k_plus_1": rng.integers(0, 4, k + 1).astype(np.uint8),
        "polyA": np.zeros(300, np.uint8),                                      # all-N reads look like this (N->0)
        "polyT": np.full(300, 3, np.uint8),
        "polyC": np.full(97, 1, np.uint8),
        "palindromic": pal,
        "dinuc": np.array([0, 3] * 150, np.uint8),
        "random_63": rng.integers(0, 4, 63).astype(np.uint8),
        "random_64": rng.integers(0, 4, 64).astype(np.uint8),
        "random_65": rng.integers(0, 4, 65).astype(np.uint8),
        "random_1000": rng.integers(0, 4, 1000).astype(np.uint8),
        "random_5000": rng.integers(0, 4, 5000).astype(np.uint8),
        "xorshift_10k_prefix": synth.xorshift_read(3000)[0],                   # SURVEY §8(c) known-answer read
    }
    return reads


def gen_scan_vectors():
    out = {}
    configs = [(21, 64, 17), (31, 4, 17), (19, 31, 17), (16, 32, 0), (1, 1, 17), (2, 3, 5), (31, 1, 17), (8, 6, 17), (21, 64, 3)]
    out["configs"] = np.array(configs, np.int32)
    for ci, (k, w, seed) in enumerate(configs):
        sh = R.seqhashCreate(k, w, seed)
        out["c%d_factor1" % ci] = np.array([sh.contents.factor1, sh.contents.factor2], np.uint64)
        rng = np.random.default_rng(1000 * k + w)
        reads = edge_reads(k, rng)
        out["c%d_names" % ci] = np.array(list(reads.keys()))
        for name, bases in reads.items():
            km, pos, isf = po.ref_scan(sh, bases)
            out["c%d_%s_in" % (ci, name)] = bases
            out["c%d_%s_kmer" % (ci, name)] = km
            out["c%d_%s_pos" % (ci, name)] = pos
            out["c%d_%s_isf" % (ci, name)] = isf
            if len(bases) >= k and w <= 64:
                hm, pm, fm = po.ref_scan(sh, bases, minimizer=True)
                out["c%d_%s_min_hash" % (ci, name)] = hm
                out["c%d_%s_min_pos" % (ci, name)] = pm
                out["c%d_%s_min_isf" % (ci, name)] = fm
    np.savez_compressed(os.path.join(HERE, "scan_vectors.npz"), **out)
    print("scan_vectors.npz", len(out), "arrays")


def make_reads_fasta():
    """A small genome, reads drawn from it with errors (so depths > 1 appear), plus edge reads."""
    genome = synth.iid_bases(40000, 4242)
    starts, offsets, strands = synth.ont_read_plan(300000, len(genome), 9, n50=3000, sigma=0.5, lo=50, hi=9000)
    bases = synth.reads_from_genome(genome, starts, offsets, strands, 0.01, 31)
    names, seqs = [], []
    for r in range(len(starts)):
        names.append("read%d" % r)
        seqs.append(bases[int(offsets[r]):int(offsets[r + 1])])
    # edge reads: short, exactly k, with Ns and lower case
    names += ["short", "exact21", "withN"]
    seqs += ["ACGTACGT", "ACGTTGCAAGGCTTAACCGGA",
             "acgtNNNNacgtacgtacgtTTGACCANNGTAGGACCATTTACGGATTACAGGATTTACCCAGGATTACAGGGTTTAAACCCGGGTTTACGATCGATCGGGATATTAGC"]
    path = os.path.join(HERE, "reads.fa")
    fasta.write_fasta(path, names, seqs)
    # second file for a second -a
    starts2, offsets2, strands2 = synth.ont_read_plan(120000, len(genome), 10, n50=2000, sigma=0.5, lo=50, hi=6000)
    bases2 = synth.reads_from_genome(genome, starts2, offsets2, strands2, 0.02, 32)
    fasta.write_fasta(os.path.join(HERE, "reads2.fa"), ["b%d" % r for r in range(len(starts2))],
                      [bases2[int(offsets2[r]):int(offsets2[r + 1])] for r in range(len(starts2))])
    return genome


def strip_timing(text):
    """drop the getrusage lines (utils.c:187-193): timing noise"""
    return "\n".join(l for l in text.splitlines() if not l.startswith("user\t") and "resources used" not in l
                     and not l.startswith("total resources")) + "\n"


def run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=HERE)
    assert r.returncode == 0, (cmd, r.stderr[-2000:])
    return r.stdout


def gen_modutils():
    mu = os.path.join(REFDIR, "modutils_ref")
    for tag, (B, k, w, s) in {"k21d64": (20, 21, 64, 17), "k31d4": (22, 31, 4, 17), "k19d31": (20, 19, 31, 17)}.items():
        out = run([mu, "-c", str(B), str(k), str(w), str(s), "-a", "reads.fa", "-a", "reads2.fa",
                   "-wt", "modutils_%s.dump.txt" % tag, "-H", "modutils_%s.hist.txt" % tag,
                   "-p", "2", "40", "-H", "modutils_%s.pruned_hist.txt" % tag,
                   "-wt", "modutils_%s.pruned_dump.txt" % tag])
        open(os.path.join(HERE, "modutils_%s.stdout.txt" % tag), "w").write(strip_timing(out))
        print("modutils", tag, len(out.splitlines()), "lines")


def make_modmap_inputs(genome_seed=777):
    """Reference with a duplicated segment (copy-2 modimizers) and a triplicated one (copy-M);
    queries: clean reads, chimeric reads (two loci), reverse-strand reads, a junk read."""
    rng = np.random.default_rng(5)
    g = synth.iid_bases(60000, genome_seed)
    chrA = g[:30000].copy()
    chrB = g[30000:60000].copy()
    chrB[5000:8000] = chrA[10000:13000]          # duplicated segment -> copy 2
    chrB[15000:16500] = chrA[20000:21500]        # triplicated (with chrC) -> copy M
    chrC = np.concatenate([synth.iid_bases(4000, 999), chrA[20000:21500], synth.iid_bases(3000, 998)])
    fasta.write_fasta(os.path.join(HERE, "ref.fa"), ["chrA", "chrB", "chrC"], [chrA, chrB, chrC])
    qn, qs = [], []

    def mutate(s, rate, seed):
        r = np.random.default_rng(seed)
        s = s.copy()
        m = r.random(len(s)) < rate
        s[m] = (s[m] + 1 + r.integers(0, 3, int(m.sum()))) & 3
        return s
    qn.append("cleanA"); qs.append(chrA[2000:9000])
    qn.append("cleanB_rc"); qs.append(3 - chrB[20000:27000][::-1])
    qn.append("noisyA"); qs.append(mutate(chrA[12000:20000], 0.03, 1))
    qn.append("chimeraAB"); qs.append(np.concatenate([chrA[1000:6000], chrB[22000:28000]]))
    qn.append("chimeraBA_rc"); qs.append(np.concatenate([chrB[9000:15000], 3 - chrA[24000:29000][::-1]]))
    qn.append("dupregion"); qs.append(chrA[9000:14500])
    qn.append("tripregion"); qs.append(chrA[19000:23000])
    qn.append("junk"); qs.append(synth.iid_bases(5000, 31337))
    qn.append("tiny"); qs.append(chrA[100:110])
    qn.append("three_way"); qs.append(np.concatenate([chrA[3000:7000], chrC[500:3500], chrB[1000:4500]]))
    for i in range(6):
        a = int(rng.integers(0, 22000)); L = int(rng.integers(3000, 8000))
        qn.append("rand%d" % i); qs.append(mutate(chrA[a:a + L] if i % 2 else chrB[a:a + L], 0.01 * i, 50 + i))
    fasta.write_fasta(os.path.join(HERE, "queries.fa"), qn, qs)


def gen_modmap():
    mm = os.path.join(REFDIR, "modmap_ref")
    for tag, (k, w) in {"k21d64": (21, 64), "k15d8": (15, 8), "k19d31": (19, 31)}.items():
        out = run([mm, "-K", str(k), "-W", str(w), "-S", "17", "-B", "20", "-f", "ref.fa", "-q", "queries.fa"])
        open(os.path.join(HERE, "modmap_%s.stdout.txt" % tag), "w").write(strip_timing(out))
        print("modmap", tag, sum(l.startswith("M\t") for l in out.splitlines()), "M lines,",
              sum(l.startswith("Q\t") for l in out.splitlines()), "Q lines")


def gen_modmap_verbose():
    """modmap -v: the per-seed lines of modmap.c:218-229 (printf, interleaved with the Q / M lines on stdout)"""
    mm = os.path.join(REFDIR, "modmap_ref")
    for tag, (k, w) in {"k21d64": (21, 64), "k15d8": (15, 8)}.items():
        out = run([mm, "-K", str(k), "-W", str(w), "-S", "17", "-B", "20", "-v", "-f", "ref.fa", "-q", "queries.fa"])
        open(os.path.join(HERE, "modmap_%s.verbose.stdout.txt" % tag), "w").write(strip_timing(out))
        print("modmap -v", tag, sum(l.startswith("  ") and "\t" in l for l in out.splitlines()), "seed lines")


def gen_modmap_files():
    """modmap -f ref.fa -w stem, then -r stem -q queries.fa: the reference's own .mod/.ref pair (gzip
    streams, utils.c:107-127) and what it prints when it reads them back."""
    mm = os.path.join(REFDIR, "modmap_ref")
    for tag, (k, w) in {"k21d64": (21, 64), "k15d8": (15, 8)}.items():
        stem = "modmap_%s_files" % tag
        run([mm, "-K", str(k), "-W", str(w), "-S", "17", "-B", "20", "-f", "ref.fa", "-w", stem])
        out = run([mm, "-r", stem, "-q", "queries.fa"])
        open(os.path.join(HERE, "%s.stdout.txt" % stem), "w").write(strip_timing(out))
        print("modmap files", tag, [os.path.getsize(os.path.join(HERE, stem + e)) for e in (".mod", ".ref")],
              sum(l.startswith("M\t") for l in out.splitlines()), "M lines")


def gen_modmap_many():
    """1100 short reference sequences: the DICT table doubles twice (dict.c:166) and the length Array
    once (array.c:144) before modmap -w dumps them"""
    mm = os.path.join(REFDIR, "modmap_ref")
    g = synth.iid_bases(1100 * 64, 4242)
    seqs = [g[i * 64:i * 64 + 40 + (i * 7) % 24] for i in range(1100)]
    fasta.write_fasta(os.path.join(HERE, "many.fa"), ["ctg%d_%s" % (i, "x" * (i % 5)) for i in range(1100)], seqs)
    run([mm, "-K", "15", "-W", "8", "-S", "17", "-B", "20", "-f", "many.fa", "-w", "modmap_many_files"])
    print("modmap many", [os.path.getsize(os.path.join(HERE, "modmap_many_files" + e)) for e in (".mod", ".ref")])


def make_seqio_inputs():
    """text files that exercise seqio.c's FASTA/FASTQ parsing (seqio.c:296-343) and dna2indexConv
    (seqio.c:643-652): wrapped lines of several widths, lower case, N, IUPAC codes and junk (dropped in
    FASTA), CR LF line ends, blank lines, descriptions, an empty record, a '>' inside a line."""
    rng = np.random.default_rng(99)
    g = synth.iid_bases(40000, 2024)
    letters = np.array(list("ACGT"))

    def text(b, lower_rate=0.0, n_rate=0.0, junk=None, junk_rate=0.0):
        t = letters[b].copy()
        r = rng.random(len(t))
        t[r < n_rate] = "N"
        if junk:
            m = (r >= n_rate) & (r < n_rate + junk_rate)
            t[m] = rng.choice(list(junk), int(m.sum()))
        low = rng.random(len(t)) < lower_rate
        t[low] = np.char.lower(t[low])
        return "".join(t)

    def wrap(sq, width, eol="\n"):
        return eol.join(sq[i:i + width] for i in range(0, len(sq), width)) + eol if sq else ""
    recs = []
    recs.append(">plain60 first record\n" + wrap(text(g[0:3000]), 60))
    recs.append(">lower\tmixed case, tab before the description\n" + wrap(text(g[3000:6500], lower_rate=0.5), 80))
    recs.append(">withN\n" + wrap(text(g[6500:9000], n_rate=0.02), 70))
    recs.append(">iupac codes and junk are dropped\n" + wrap(text(g[9000:12000], junk="RYKMSWBDHVryk*-.0123 ", junk_rate=0.03), 61))
    recs.append(">crlf\r\n" + wrap(text(g[12000:14000]), 50, eol="\r\n"))
    recs.append(">empty\n")
    recs.append(">blank_lines\n" + wrap(text(g[14000:15000]), 100) + "\n\n" + wrap(text(g[15000:16000]), 100))
    recs.append(">gt_inside\n" + text(g[16000:16050]) + ">notaheader " + text(g[16050:16100]) + "\n" + wrap(text(g[16100:17000]), 60))
    recs.append(">oneline\n" + text(g[17000:25000]) + "\n")
    recs.append(">short\nACGTAC\n")
    for i in range(40):
        a = int(rng.integers(0, 38000)); L = int(rng.integers(30, 1500))
        recs.append(">r%d len=%d\n" % (i, L) + wrap(text(g[a:a + L], lower_rate=0.1, n_rate=0.005), int(rng.integers(20, 120))))
    open(os.path.join(HERE, "mixed.fa"), "w", newline="").write("".join(recs))
    open(os.path.join(HERE, "unterminated.fa"), "w", newline="").write("".join(recs[:3]) + ">last\n" + text(g[100:400]))
    # a file whose LAST line is a header: the reference reports the record as incomplete and does not return it (seqio.c:213-217,314)
    open(os.path.join(HERE, "header_last.fa"), "w", newline="").write(">a\nACGTACGTACGTACGTACGTAAACCCGGGTTT\n>b\nACGTTTGACCGATAGACCAGATAGGGAC\n>lonely\n")
    import gzip as _gz
    with _gz.GzipFile(os.path.join(HERE, "mixed.fa.gz"), "wb", mtime=0) as f:
        f.write("".join(recs).encode())
    fq = []
    for i in range(60):
        a = int(rng.integers(0, 38000)); L = int(rng.integers(25, 400))
        sq = text(g[a:a + L], lower_rate=0.2, n_rate=0.01)
        fq.append("@fq%d some description\n%s\n+%s\n%s\n" % (i, sq, "fq%d" % i if i % 2 else "", "".join(rng.choice(list("!#5:?I@+"), L))))
    fq.append("@fqempty\n\n+\n\n")
    open(os.path.join(HERE, "mixed.fq"), "w", newline="").write("".join(fq))


def gen_seqio():
    mu = os.path.join(REFDIR, "modutils_ref")
    for name in ("mixed.fa", "mixed.fa.gz", "unterminated.fa", "mixed.fq", "header_last.fa"):
        tag = name.replace(".", "_")
        r = subprocess.run([mu, "-c", "20", "15", "4", "17", "-a", name, "-wt", "seqio_%s.dump.txt" % tag],
                           capture_output=True, text=True, cwd=HERE)
        assert r.returncode == 0, r.stderr[-500:]
        open(os.path.join(HERE, "seqio_%s.stdout.txt" % tag), "w").write(strip_timing(r.stdout))
        open(os.path.join(HERE, "seqio_%s.stderr.txt" % tag), "w").write(
            "".join(l + "\n" for l in r.stderr.splitlines() if not l.startswith("COMMAND")))
        print("seqio", name, [l for l in r.stdout.splitlines() if l.startswith("added")], r.stderr.splitlines()[-1:])
    # the gzip'd copy must give the same set: keep one dump
    a, b = (os.path.join(HERE, "seqio_mixed_fa%s.dump.txt" % x) for x in ("", "_gz"))
    assert open(a).read() == open(b).read()
    os.remove(b)


def gen_modasm():
    """modasm's read ingest: a modset with copy classes (modutils -s), then modasm -m .. -f .. -S -w:
    the stats it prints and its <stem>.mod + <stem>.readset files (gzip streams)."""
    mu, ma = os.path.join(REFDIR, "modutils_ref"), os.path.join(REFDIR, "modasm_ref")
    for tag, (k, w) in {"k21d16": (21, 16), "k17d31": (17, 31)}.items():
        run([mu, "-c", "20", str(k), str(w), "17", "-a", "reads.fa", "-s", "2", "3", "5",
             "-w", "asm_%s_src.mod" % tag])
        out = run([ma, "-m", "asm_%s_src.mod" % tag, "-f", "reads2.fa", "-S", "-w", "asm_%s" % tag])
        open(os.path.join(HERE, "asm_%s.stdout.txt" % tag), "w").write(strip_timing(out))
        print("modasm", tag, [os.path.getsize(os.path.join(HERE, "asm_%s%s" % (tag, e))) for e in ("_src.mod", ".mod", ".readset")])


def gen_modset_ops():
    """modsetMerge / modsetDepthPrune / modsetPack / modsetWrite through the reference library."""
    out = {}
    names, bases, offs = fasta.read_fasta(os.path.join(HERE, "reads.fa"))
    names2, bases2, offs2 = fasta.read_fasta(os.path.join(HERE, "reads2.fa"))
    k, w, seed, B = 21, 16, 17, 20
    sh = R.seqhashCreate(k, w, seed)
    a = R.modsetCreate(sh, B, 0)
    b = R.modsetCreate(sh, B, 0)
    for r in range(len(names)):
        po.ref_add_sequence(a, bases[offs[r]:offs[r + 1]])
    for r in range(len(names2)):
        po.ref_add_sequence(b, bases2[offs2[r]:offs2[r + 1]])
    # give b some copy bits so the merge's info arithmetic is exercised (modset.c:124-125)
    for i in range(1, b.contents.max + 1):
        b.contents.info[i] = (i % 4) | ((i % 3 == 0) * 8)
    for i in range(1, a.contents.max + 1):
        a.contents.info[i] = ((i // 2) % 4) | ((i % 5 == 0) * 16)

    def snap(ms, tag):
        n = ms.contents.max + 1
        out[tag + "_value"] = np.ctypeslib.as_array(ms.contents.value, (n,)).copy()
        out[tag + "_depth"] = np.ctypeslib.as_array(ms.contents.depth, (n,)).copy()
        out[tag + "_info"] = np.ctypeslib.as_array(ms.contents.info, (n,)).copy()
        idx = np.ctypeslib.as_array(ms.contents.index, (1 << B,))
        nz = np.nonzero(idx)[0]
        out[tag + "_index_pos"] = nz.astype(np.uint32)
        out[tag + "_index_val"] = idx[nz].copy()
        out[tag + "_summary"] = np.frombuffer(po.ref_text(R.modsetSummary, ms, "/tmp/_golden_sum.txt"), np.uint8)
    snap(a, "a"); snap(b, "b")
    modbytes = po.ref_text(R.modsetWrite, a, "/tmp/_golden_a.mod")
    import hashlib
    # value[0] is uninitialised memory in the reference (modset.c:27): zero it before hashing
    n = a.contents.max + 1
    voff = 8 + 4 + 4 + 8 + 80 + 4 * (1 << B)
    mb = bytearray(modbytes); mb[voff:voff + 8] = b"\0" * 8
    out["a_mod_sha256"] = np.frombuffer(hashlib.sha256(bytes(mb)).digest(), np.uint8)
    out["a_mod_len"] = np.array([len(modbytes)], np.int64)
    out["a_mod_header"] = np.frombuffer(modbytes[:8 + 4 + 4 + 8 + 80], np.uint8)
    assert R.modsetMerge(a, b)
    snap(a, "merged")
    R.modsetDepthPrune(a, 2, 30)
    snap(a, "pruned")
    R.modsetPack(a)
    out["packed_size"] = np.array([a.contents.size], np.uint32)
    out["params"] = np.array([k, w, seed, B], np.int32)
    np.savez_compressed(os.path.join(HERE, "modset_ops.npz"), **out)
    print("modset_ops.npz", {k_: v.shape for k_, v in out.items() if k_.endswith("_value")})


def digest_dumps():
    """the -wt dumps are large: keep the k21d64 ones in full, the others as sha256 + head/tail"""
    import hashlib
    import json
    for f in sorted(os.listdir(HERE)):
        if f.endswith("dump.txt") and "k21d64" not in f:
            txt = open(os.path.join(HERE, f)).read()
            lines = txt.splitlines()
            json.dump({"sha256": hashlib.sha256(txt.encode()).hexdigest(), "lines": len(lines),
                       "head": lines[:40], "tail": lines[-5:]},
                      open(os.path.join(HERE, f.replace(".txt", ".digest.json")), "w"), indent=0)
            os.remove(os.path.join(HERE, f))


if __name__ == "__main__":
    gen_scan_vectors()
    make_reads_fasta()
    gen_modutils()
    make_modmap_inputs()
    gen_modmap()
    gen_modmap_verbose()
    gen_modmap_files()
    gen_modmap_many()
    gen_modasm()
    make_seqio_inputs()
    gen_seqio()
    gen_modset_ops()
    digest_dumps()
    for fn in ("/tmp/_golden_sum.txt", "/tmp/_golden_a.mod"):
        if os.path.exists(fn):
            os.remove(fn)
    print("sizes:", {f: os.path.getsize(os.path.join(HERE, f)) for f in sorted(os.listdir(HERE))})
