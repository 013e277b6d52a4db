"""Minimal FASTA reader mirroring what the reference's seqio hands to the hot path
(seqio.c:234-346 with dna2indexConv, seqio.c:643-652, after the callers' N->0 patch
modmap.c:97 / modutils.c:39): a/c/g/t -> 0..3 (either case), n -> 0, every other character is
dropped from the sequence.  Host-side convenience for tests and drivers; not an accelerated path."""
import numpy as np

_CONV = np.full(256, 255, np.uint8)
for _c, _v in (("a", 0), ("c", 1), ("g", 2), ("t", 3), ("n", 0)):
    _CONV[ord(_c)] = _v
    _CONV[ord(_c.upper())] = _v


def encode(seq_bytes):
    a = _CONV[np.frombuffer(seq_bytes, np.uint8)]
    return a[a != 255]


def read_fasta(path):
    """Returns (names, bases uint8 concatenated, offsets int64[n+1])."""
    names, chunks = [], []
    cur = []
    with open(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                if names:
                    chunks.append(encode(b"".join(cur)))
                names.append(line[1:].split()[0].decode() if len(line) > 1 else "")
                cur = []
            elif names:
                cur.append(line)
    if names:
        chunks.append(encode(b"".join(cur)))
    offsets = np.zeros(len(chunks) + 1, np.int64)
    if chunks:
        offsets[1:] = np.cumsum([len(c) for c in chunks])
    bases = np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)
    return names, bases, offsets


def write_fasta(path, names, seqs, width=70):
    with open(path, "w") as f:
        for n, s in zip(names, seqs):
            f.write(">%s\n" % n)
            txt = "".join("ACGT"[b] for b in s) if not isinstance(s, str) else s
            for i in range(0, len(txt), width):
                f.write(txt[i:i + width] + "\n")


def read_fasta_list(path):
    """the sequences of a FASTA file as a list of uint8 arrays"""
    _, bases, offs = read_fasta(path)
    return [bases[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
