"""Deterministic synthetic inputs; numpy mirror of csrc/mg_synth.hip.

Not part of the reference: these define the benchmark's and the parity tests' inputs.  Everything
is counter-based (splitmix64 of a global ordinal) so the device generator and this host mirror
produce identical bases without any file.
"""
import numpy as np

GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + GOLD
        x = (x ^ (x >> np.uint64(30))) * _M1
        x = (x ^ (x >> np.uint64(27))) * _M2
        return x ^ (x >> np.uint64(31))


def iid_bases(n, seed, start=0):
    """base g = splitmix64(seed ^ g*GOLD) >> 62 for g in [start, start+n) — mgSynthGenome."""
    g = np.arange(start, start + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return (splitmix64(np.uint64(seed) ^ (g * GOLD)) >> np.uint64(62)).astype(np.uint8)


def xorshift_read(n, state=0x9E3779B97F4A7C15):
    """The generator behind the known answers of SURVEY §8(c): x^=x<<13; x^=x>>7; x^=x<<17;
    base = x>>62.  Returns (bases, next state)."""
    out = np.empty(n, np.uint8)
    x = state
    M = (1 << 64) - 1
    for i in range(n):
        x ^= (x << 13) & M
        x ^= x >> 7
        x ^= (x << 17) & M
        out[i] = x >> 62
    return out, x


def ont_read_plan(total_bases, genome_bases, seed, n50=20000, sigma=0.6, lo=500, hi=200000):
    """Log-normal read lengths whose length-weighted median is n50 (mu = ln n50 - sigma^2),
    clamped to [lo, hi]; uniform starts; random strands.  Returns (starts u64, offsets u64
    [nReads+1], strands u8); the last read is trimmed so offsets[-1] == total_bases."""
    rng = np.random.default_rng(seed)
    mu = np.log(n50) - sigma * sigma
    mean_len = np.exp(mu + sigma * sigma / 2)
    lens = []
    got = 0
    while got < total_bases:
        m = int((total_bases - got) / mean_len * 1.1) + 16
        l = np.clip(rng.lognormal(mu, sigma, m), lo, min(hi, genome_bases)).astype(np.int64)
        lens.append(l)
        got += int(l.sum())
    lens = np.concatenate(lens)
    cs = np.cumsum(lens)
    n = int(np.searchsorted(cs, total_bases)) + 1
    lens = lens[:n].copy()
    lens[-1] -= cs[n - 1] - total_bases
    if lens[-1] <= 0:
        lens = lens[:-1]
    offsets = np.zeros(len(lens) + 1, np.uint64)
    offsets[1:] = np.cumsum(lens).astype(np.uint64)
    starts = (rng.random(len(lens)) * (genome_bases - lens + 1)).astype(np.uint64)
    strands = (rng.random(len(lens)) < 0.5).astype(np.uint8)
    return starts, offsets, strands


def fixed_read_plan(n_reads, read_len, genome_bases, seed):
    """n_reads reads of one length (the Illumina-like config)."""
    rng = np.random.default_rng(seed)
    offsets = (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len))
    starts = (rng.random(n_reads) * (genome_bases - read_len + 1)).astype(np.uint64)
    strands = (rng.random(n_reads) < 0.5).astype(np.uint8)
    return starts, offsets, strands


def reads_from_genome(genome, starts, offsets, strands, err_rate, seed):
    """Host mirror of mgSynthReads: returns the concatenated bases (uint8 0..3)."""
    total = int(offsets[-1])
    out = np.empty(total, np.uint8)
    for r in range(len(starts)):
        a, b = int(offsets[r]), int(offsets[r + 1])
        L = b - a
        seg = genome[int(starts[r]):int(starts[r]) + L]
        out[a:b] = (3 - seg[::-1]) if strands[r] else seg
    q = np.arange(total, dtype=np.uint64)
    with np.errstate(over="ignore"):
        u = splitmix64(np.uint64(seed) ^ (q * GOLD))
    thresh = 0 if err_rate <= 0 else int(err_rate * 18446744073709551616.0)
    hit = u < np.uint64(thresh)
    sub = ((u & np.uint64(0xFFFF)) % np.uint64(3)).astype(np.uint8)
    out[hit] = (out[hit] + 1 + sub[hit]) & 3
    return out


def repeat_genome(n_bases, seed, alu=0.10, satellite=0.03, ca=0.01):
    """A genome with the repeat structure real ones have, by share of its length: an Alu-like family (300 b, 10 % divergence
    between copies) each copy with a poly-A tail of 15-45 bases, satellite arrays (200-3000 copies of a 171-base monomer, 2 %
    divergence), (CA)n microsatellites, unique sequence in between.  Not from the reference: the workload behind
    bench.py's other_configs.realistic (poly-A and monomer k-mers with 1e5-1e6 copies: one table bucket each)."""
    rng = np.random.default_rng(seed)

    def mutate(x, rate):
        x = x.copy(); m = rng.random(len(x)) < rate
        x[m] = (x[m] + rng.integers(1, 4, int(m.sum()))) & 3
        return x
    alu_seq = rng.integers(0, 4, 300).astype(np.uint8); mono = rng.integers(0, 4, 171).astype(np.uint8)
    share = {"alu": alu, "sat": satellite, "ca": ca}
    have = dict.fromkeys(share, 0)
    parts, n = [], 0
    while n < n_bases:
        kind = next((k for k in share if have[k] < share[k] * n), None) if n else None
        if kind == "alu":
            p = np.concatenate([mutate(alu_seq, 0.10), np.zeros(int(rng.integers(15, 45)), np.uint8)])
        elif kind == "sat":
            p = np.concatenate([mutate(mono, 0.02) for _ in range(int(rng.integers(200, 3000)))])
        elif kind == "ca":
            p = np.tile(np.array([1, 0], np.uint8), int(rng.integers(10, 60)))
        else:
            p = rng.integers(0, 4, int(rng.integers(200, 6000))).astype(np.uint8)
        if kind:
            have[kind] += len(p)
        parts.append(p); n += len(p)
    return np.concatenate(parts)[:n_bases]
