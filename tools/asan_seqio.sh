#!/bin/bash
# dev: the text parser (mg_seqio.c) under AddressSanitizer + UBSan on the CPU, over the golden text files
# and a generated short-read FASTQ file.  usage: bash tools/asan_seqio.sh
set -e
cd "$(dirname "$0")/.."
B=gpurun_out/asan; mkdir -p $B
cat > $B/main.c <<'EOC'
#include <stdio.h>
#include <stdlib.h>
#include "modgpu.h"
/* the parser's callers live in other units: not exercised here */
int64_t mgAddSequenceBatch (Modset *ms, const char *b, const int64_t *o, int n) { (void) ms; (void) b; (void) o; (void) n; return 0; }
int mgReferenceRead (MgReference *r, const char *b, const int64_t *o, int n, const char **nm, bool a, FILE *f) { (void) r; (void) b; (void) o; (void) n; (void) nm; (void) a; (void) f; return 0; }
int mgQueryProcess (MgReference *r, const char *b, const int64_t *o, int n, const char **nm, FILE *f) { (void) r; (void) b; (void) o; (void) n; (void) nm; (void) f; return 0; }
#include "mg_internal.h"
int mgTextForEachBatchDevice (const char *fn, MgTextBatchFn f, void *c, U64 bb, U64 br, U64 *a, U64 *b, U64 *d, U64 *e) { (void) fn; (void) f; (void) c; (void) bb; (void) br; (void) a; (void) b; (void) d; (void) e; return -2; }
int mgQueryProcessDevice (MgReference *r, const U32 *p, U64 t, const U64 *o, int n, const char **nm, FILE *f) { (void) r; (void) p; (void) t; (void) o; (void) n; (void) nm; (void) f; return 0; }
MgQueryPipe *mgQueryPipeOpen (MgReference *r, FILE *f) { (void) r; (void) f; return 0; }
int mgQueryPipePush (MgQueryPipe *p, const U32 *d, U64 t, const U64 *o, int n, const char *ib, const U64 *io) { (void) p; (void) d; (void) t; (void) o; (void) n; (void) ib; (void) io; return 0; }
void mgQueryPipeClose (MgQueryPipe *p) { (void) p; }
int mgReferenceAddDevice (MgReference *r, const U32 *p, U64 t, const U64 *o, int n, const char **nm, bool a) { (void) r; (void) p; (void) t; (void) o; (void) n; (void) nm; (void) a; return 0; }
void mgReferenceFinish (MgReference *r, U64 t, bool a, FILE *f) { (void) r; (void) t; (void) a; (void) f; }
int mgAddSequenceFileDevice (Modset *ms, const char *fn, U64 *a, U64 *b, U64 *c, U64 *d, U64 *e) { (void) ms; (void) fn; (void) a; (void) b; (void) c; (void) d; (void) e; return -2; }
int main (int argc, char **argv)
{
  for (int a = 2 ; a < argc ; ++a)
    { MgSeqReader *r = mgSeqOpen (argv[a]); if (!r) continue;
      MgSeqBatch b; unsigned long long sum = 0, ids = 0, n = 0;
      while (mgSeqNextBatch (r, atoll (argv[1]), &b))
        { for (int i = 0 ; i < b.nSeq ; ++i) { for (const char *p = b.names[i] ; *p ; ++p) ids = ids * 31 + (unsigned char) *p; ids = ids * 31 + (unsigned long long) (b.offsets[i + 1] - b.offsets[i]); }
          for (int64_t i = 0 ; i < b.total ; ++i) sum = sum * 131 + (unsigned char) b.bases[i];
          n += b.nSeq; mgSeqBatchFree (&b);
        }
      mgSeqClose (r);
      printf ("%s: %llu records, checksums %llx %llx\n", argv[a], n, ids, sum);
    }
  return 0;
}
EOC
gcc -g -O1 -std=gnu11 -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude -Imodimizer_amd/csrc \
    $B/main.c modimizer_amd/csrc/mg_seqio.c modimizer_amd/csrc/mg_knobs.c -o $B/seqio_asan -lz -lpthread
python3 - <<'EOP'
import numpy as np
rng = np.random.default_rng(2)
with open("gpurun_out/asan/short.fq", "wb") as f:
    for i in range(120000):
        l = int(rng.integers(0, 200)); s = np.frombuffer(b"ACGTNacgtRY", np.uint8)[rng.integers(0, 11, l)].tobytes()
        f.write(b"@r%d x\n" % i + s + b"\n+\n" + b"I" * l + b"\n")
    f.write(b"@cut\nACG")
import struct, zlib
raw = open("gpurun_out/asan/short.fq", "rb").read()
with open("gpurun_out/asan/short.fq.bgz", "wb") as f:                 # the same text as blocked gzip
    for i in range(0, len(raw), 60000):
        ch = raw[i:i + 60000]; c = zlib.compressobj(1, zlib.DEFLATED, -15); pay = c.compress(ch) + c.flush()
        f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(pay) + 25) + pay + struct.pack("<II", zlib.crc32(ch), len(ch)))
    f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
# records whose raw sequence text is an exact multiple of the 1 MiB conversion unit (FASTA: newlines included)
L = np.frombuffer(b"ACGT", np.uint8)
with open("gpurun_out/asan/mult.fa", "wb") as f:
    for i, lines in enumerate((16384, 3, 32768, 16383)):
        s = L[rng.integers(0, 4, lines * 63)]
        f.write(b">m%d\n" % i + b"".join(s[j:j + 63].tobytes() + b"\n" for j in range(0, len(s), 63)))
with open("gpurun_out/asan/mult.fq", "wb") as f:
    for i, n in enumerate((1 << 20, 7, 2 << 20, (1 << 20) + 1)):
        f.write(b"@m%d\n" % i + L[rng.integers(0, 4, n)].tobytes() + b"\n+\n" + b"I" * n + b"\n")
EOP
G=tests/golden
for mb in 3000 777777 1000000000; do
  for t in 1 7; do
    MODGPU_PARSE_THREADS=$t $B/seqio_asan $mb $G/mixed.fa $G/mixed.fa.gz $G/mixed.fq $G/unterminated.fa $G/many.fa $B/short.fq $B/short.fq.bgz $B/mult.fa $B/mult.fq 2>&1 | md5sum
  done
done
echo "(all six digests above must be equal)"
