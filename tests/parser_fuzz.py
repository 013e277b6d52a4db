"""Random FASTA / FASTQ shapes for the parser loop against the REFERENCE (seqio.c:234-346): the text of trial `seed`
is a pure function of the seed, so the fixture (tests/golden/parser_fuzz.json, made by tests/golden/make_parser_fuzz.py where the
compiled reference is) holds seeds, a digest of the text, and what `modutils_ref -c B k w 17 -a <file>` said about it.

Shapes: CR LF, blank lines, '>' inside header and sequence lines and at the start of a sequence line, lower case, N, IUPAC codes and junk
bytes (FASTA drops them: seqio.c:321-323), empty records, headers back to back, tabs, no final newline, a header as the last line, one very
long line; FASTQ with repeated ids on the '+' line, '@' and '+' at the start of quality lines -- and, as error cases, text that is
neither, a missing '+', a quality line of another length, a file cut in the middle of a record, a blank line between records.
FASTQ sequence lines hold only ACGTN in either case: any other byte is kept by the reference as -2 (seqio.c:328-331) and indexes
patternRC[-2] when it is hashed (seqhash.c:62) -- undefined there, not a behaviour to reproduce."""
import hashlib

import numpy as np

KW = [(21, 64), (15, 8), (19, 31), (11, 3), (31, 4), (16, 32)]
N_TRIALS = 640


def params(seed):
    k, w = KW[seed % len(KW)]
    return 24, k, w


def make_text(seed):
    rng = np.random.default_rng(1_000_003 * seed + 17)
    fastq = rng.random() < 0.4
    return ("fastq", _fastq(rng)) if fastq else ("fasta", _fasta(rng))


def digest(text):
    return hashlib.sha1(text).hexdigest()


def _fasta(rng):
    letters = np.frombuffer(b"ACGTacgtNnRYKMSWBDHVxX*-. \t>1", np.uint8)
    wt = np.array([20, 20, 20, 20, 5, 5, 5, 5, 2, 1] + [0.4] * 8 + [0.5] * 11, float)[:len(letters)]; wt /= wt.sum()
    n_rec = int(rng.choice([1, 2, 3, 7, 40, 300]))
    crlf_file = rng.random() < 0.15
    lines = []
    for r in range(n_rec):
        eol = b"\r" if (crlf_file or rng.random() < 0.05) else b""
        hdr = [b">r%d" % r, b">r%d desc with > and\ttab" % r, b">%d|x|y" % r, b">r%d " % r][int(rng.integers(0, 4))]
        lines.append(hdr + eol)
        shape = rng.random()
        n = 0 if shape < 0.08 else int(rng.integers(1, 40)) if shape < 0.3 else int(rng.integers(40, 4000)) if shape < 0.97 else 150_000
        seq = letters[rng.choice(len(letters), n, p=wt)].tobytes()
        width = int(rng.choice([0, 1, 7, 60, 80, 4096]))
        body = [seq] if (width == 0 or not seq) else [seq[i:i + width] for i in range(0, len(seq), width)]
        for l in body:
            lines.append(l + eol)
            if rng.random() < 0.03:
                lines.append(eol)                                      # a blank line inside the record
    text = b"\n".join(lines) + b"\n"
    end = rng.random()
    if end < 0.12:
        text = text[:-1]                                               # no final newline
    elif end < 0.2:
        text += b">last header" + (b"\n" if rng.random() < 0.5 else b"")      # a header as the last line: an incomplete record
    elif end < 0.24:
        text = b"\n" + text                                            # does not start with '>': not FASTA at all
    elif end < 0.27:
        text = b"ACGT\n" + text
    return text


def _fastq(rng):
    n_rec = int(rng.choice([1, 2, 5, 60, 1500]))
    recs = []
    sl = np.frombuffer(b"ACGTacgtNn", np.uint8); sw = np.array([22, 22, 22, 22, 2, 2, 2, 2, 2, 2], float); sw /= sw.sum()
    ql = np.frombuffer(b"!#+5@FIJ~>", np.uint8)
    for r in range(n_rec):
        shape = rng.random()
        n = 0 if shape < 0.03 else int(rng.integers(1, 30)) if shape < 0.2 else int(rng.choice([100, 150, 151, 250])) if shape < 0.9 else int(rng.integers(300, 30000))
        seq = sl[rng.choice(len(sl), n, p=sw)].tobytes()
        qual = ql[rng.integers(0, len(ql), n)].tobytes()
        name = b"r%d" % r if rng.random() < 0.7 else b"r%d 1:N:0:ACGT+TTTT" % r
        plus = b"+" if rng.random() < 0.8 else b"+" + name
        recs.append([b"@" + name, seq, plus, qual])
    bad = rng.random()
    if bad < 0.05 and n_rec > 1:
        recs[int(rng.integers(0, n_rec))][2] = b"-"                    # no '+'
    elif bad < 0.1:
        i = int(rng.integers(0, n_rec)); recs[i][3] = recs[i][3] + b"I"      # quality of another length
    text = b"\n".join(b"\n".join(r) for r in recs) + b"\n"
    if 0.1 <= bad < 0.16:
        cut = int(rng.integers(1, len(text)))
        text = text[:cut]                                              # the file ends somewhere
    elif 0.16 <= bad < 0.2 and n_rec > 1:
        i = int(rng.integers(1, n_rec)); parts = [b"\n".join(r) for r in recs]
        text = b"\n".join(parts[:i]) + b"\n\n" + b"\n".join(parts[i:]) + b"\n"      # a blank line between two records
    elif 0.2 <= bad < 0.24:
        text = text[:-1]                                               # no final newline
    return text
