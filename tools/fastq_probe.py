#!/usr/bin/env python3
"""dev probe: short-read FASTQ file -> modset end to end (k=31 d=4 as BASELINE config 5)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import modimizer_amd as mg
n_reads = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 4_000_000
path = "/dev/shm/probe.fq"
rng = np.random.default_rng(1)
letters = np.frombuffer(b"ACGT", np.uint8)
rec = 150
seqs = letters[rng.integers(0, 4, (n_reads, rec))]
hdr = np.frombuffer(b"@r\n", np.uint8); plus = np.frombuffer(b"\n+\n", np.uint8); q = np.full((n_reads, rec), ord("I"), np.uint8); nl = np.full((n_reads, 1), 10, np.uint8)
block = np.concatenate([np.tile(hdr, (n_reads, 1)), seqs, np.tile(plus, (n_reads, 1)), q, nl], axis=1)
block.tofile(path); size = os.path.getsize(path); del block, seqs, q
L = mg.lib()
for threads in (16,):
    os.environ["MODGPU_PARSE_THREADS"] = str(threads)
    mg.lib().mgReloadKnobs()
    t0 = time.time(); r = L.mgSeqOpen(path.encode()); b = mg.MgSeqBatch(); tot = 0
    while L.mgSeqNextBatch(r, 512_000_000, C.byref(b)):
        tot += b.total; L.mgSeqBatchFree(C.byref(b))
    L.mgSeqClose(r); dt = time.time() - t0
    print("FASTQ parse, %2d threads: %.2f s  %.2f GB/s text  %.2f Gbp/s" % (threads, dt, size / dt / 1e9, tot / dt / 1e9))
del os.environ["MODGPU_PARSE_THREADS"]
mg.lib().mgReloadKnobs()
if L.mgDeviceCount() > 0:
    sh = mg.seqhashCreate(31, 4, 17); ms = mg.modsetCreate(sh, 32)
    for host in ("1", "0", "0", "1", "0"):
        os.environ["MODGPU_TEXT_HOST"] = host                      # 1: the host parser; 0: the text parsed on the device
        mg.lib().mgReloadKnobs()
        L.mgModsetClear(ms, None); t0 = time.time()
        with mg.CFile("/dev/null", "w") as f:
            rc = L.mgAddSequenceFile(ms, path.encode(), f)
            assert rc == 0, L.mgLastError().decode()
        dt = time.time() - t0
        print("FASTQ file -> modset (%s parser): %.3f s  %.2f Gbp/s (max %d)" % ("host" if host == "1" else "device", dt, n_reads * rec / dt / 1e9, ms.contents.max))
# the same text as ordinary gzip (one zlib stream, the reference's path) and as blocked gzip (inflated by the pool)
import gzip, struct, zlib
raw = open(path, "rb").read()[: 300 * (1 << 20)]
raw = raw[: raw.rindex(b"\n@r\n") + 1]
open(path + ".gz", "wb").write(gzip.compress(raw, 1))
with open(path + ".bgz", "wb") as f:
    for i in range(0, len(raw), 65280):
        ch = raw[i:i + 65280]; c = zlib.compressobj(1, zlib.DEFLATED, -15); pay = c.compress(ch) + c.flush()
        f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(pay) + 25) + pay + struct.pack("<II", zlib.crc32(ch), len(ch)))
    f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
for ext in (".gz", ".bgz"):
    t0 = time.time(); r = L.mgSeqOpen((path + ext).encode()); b = mg.MgSeqBatch(); tot = 0
    while L.mgSeqNextBatch(r, 512_000_000, C.byref(b)):
        tot += b.total; L.mgSeqBatchFree(C.byref(b))
    L.mgSeqClose(r); dt = time.time() - t0
    print("FASTQ%s parse: %.2f s  %.2f GB/s text  %.2f Gbp/s  (%d MB compressed)" % (ext, dt, len(raw) / dt / 1e9, tot / dt / 1e9, os.path.getsize(path + ext) >> 20))
    os.remove(path + ext)
os.remove(path)
