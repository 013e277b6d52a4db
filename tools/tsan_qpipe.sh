#!/bin/bash
# dev: mgQueryFile's three-stage output pipeline (mg_callers.c mgQueryPipe*: device half on the caller's thread, formatter team, writer)
# under ThreadSanitizer, then under AddressSanitizer + UBSan, on the CPU build with the device half stubbed: the stub hands out
# deterministic tallies and blocks, the harness pushes 60 batches of 1 .. 50000 reads (small ones format on one thread, large ones on the
# team) and compares the file with lines printed by plain fprintf -- so the order of the batches, the hand-over of the ids and the
# hand-written "%.2f" are all checked, and the queues, the page-locked pool and the frees run under the sanitizers.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); D=${TMPDIR:-/tmp}/modgpu_tsan_qpipe; mkdir -p $D
cat > $D/stubs.c <<'EOS'
#include <stdlib.h>
#include <string.h>
#include "modgpu.h"
#include "mg_internal.h"
/* "device" memory is host memory here */
MgStatus mgMemcpyD2H (void *d, const void *s, size_t n, void *st) { (void) st; memcpy (d, s, n); return MG_OK; }
MgStatus mgMemcpyH2D (void *d, const void *s, size_t n, void *st) { (void) st; memcpy (d, s, n); return MG_OK; }
MgStatus mgCopyOutPinned (void *d, const void *s, size_t n, void *st) { (void) st; memcpy (d, s, n); return MG_OK; }
void *mgPinnedAlloc (size_t n) { return malloc (n ? n : 16); } void mgPinnedFree (void *p) { free (p); }
void mgChainScratchKeep (int on) { (void) on; } void mgChainReleaseBuffers (void) {} void mgChainForget (const MgReference *r) { (void) r; }
MgStatus modsetSyncToHost (Modset *ms, int w) { (void) ms; (void) w; return MG_OK; }
const char *mgLastError (void) { return "stub"; }
/* what the harness recomputes: tallies and blocks as a function of the read's length */
void stubQ (U64 len, MgChainQ *q, MgChainM *m)
{ q->nSeeds = (U32) (len / 7 + 1); q->missed = (U32) (len % 5); if (q->missed > q->nSeeds) q->missed = q->nSeeds; q->copy1 = (U32) (len % 11 + 1); q->copy2 = (U32) (len % 3); q->copyM = (U32) (len % 2);
  q->nM = len % 13 == 0 ? (U32) (len % 4) : 0;
  for (U32 k = 0 ; k < q->nM ; ++k)
    { m[k].pos0 = (U32) (len + k); m[k].posN = (U32) (len + 100 * k + 7); m[k].id0 = (U32) ((len + k) % 3); m[k].off0 = (U32) (3 * len); m[k].offN = (U32) (3 * len + 50 + k);
      m[k].n1 = (int) (len % 9 + k); m[k].n2 = (int) (len % 6); m[k].span = (U32) (len % 17 + 1);
    }
}
int mgChainQueryDevice (const MgReference *ref, const U32 *dPacked, U64 total, const U64 *dOff, U32 nReads, MgChainQ *hQ, MgChainM **hMOut, U32 maxM, int pinned)
{ (void) ref; (void) dPacked; (void) total; (void) maxM; (void) pinned;
  MgChainM *m = (MgChainM *) malloc (((size_t) nReads * 4 + 1) * sizeof (MgChainM)); size_t nm = 0;
  for (U32 r = 0 ; r < nReads ; ++r) { stubQ (dOff[r + 1] - dOff[r], &hQ[r], m + nm); nm += hQ[r].nM; }
  if (nm) *hMOut = m; else { free (m); *hMOut = 0; }
  return 0;
}
/* never reached by the harness */
MgStatus mgQueryReadsDevice (Modset *ms, const U32 *p, U64 t, const U64 *o, U32 n, U32 *a, U32 *b, U32 *c, U64 cap, U64 *nn, void *s) { (void) ms; (void) p; (void) t; (void) o; (void) n; (void) a; (void) b; (void) c; (void) cap; (void) nn; (void) s; abort (); }
MgStatus mgInsertReadsDevice (Modset *ms, const U32 *p, U64 t, const U64 *o, U32 n, U32 *a, U32 *b, U32 *c, U64 cap, U64 *nn, void *s) { (void) ms; (void) p; (void) t; (void) o; (void) n; (void) a; (void) b; (void) c; (void) cap; (void) nn; (void) s; abort (); }
int64_t mgAddSequenceBatch (Modset *ms, const char *b, const int64_t *o, int n) { (void) ms; (void) b; (void) o; (void) n; abort (); }
MgStatus mgDeviceAlloc (void **p, size_t n) { (void) p; (void) n; abort (); } MgStatus mgDeviceFree (void *p) { (void) p; abort (); }
MgStatus mgStreamSynchronize (void *s) { (void) s; abort (); }
size_t mgPackedWords (U64 n) { (void) n; abort (); }
MgStatus mgUploadPack (const char *b, U64 n, U32 *d, void *s) { (void) b; (void) n; (void) d; (void) s; abort (); }
bool modsetPack (Modset *ms) { (void) ms; abort (); } Modset *modsetRead (FILE *f) { (void) f; abort (); } void modsetWrite (Modset *ms, FILE *f) { (void) ms; (void) f; abort (); }
char *seqString (U64 k, int len) { (void) k; (void) len; abort (); }
MgStatus mgRefBuildAppend (MgReference *ref, const U32 *a, const U32 *b, const U32 *c, U64 n, U32 idBase, U32 *appended) { (void) ref; (void) a; (void) b; (void) c; (void) n; (void) idBase; (void) appended; abort (); }
MgStatus mgRefBuildFinish (MgReference *ref, U32 *a, U32 *b, U32 *c, U32 *d, U32 *e, U32 *f, U8 *g, U32 t[3]) { (void) ref; (void) a; (void) b; (void) c; (void) d; (void) e; (void) f; (void) g; (void) t; abort (); }
int mgRefPackedTallies (MgReference *ref, U32 t[3]) { (void) ref; (void) t; return 0; }
EOS
cat > $D/main.c <<'EOS'
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "modgpu.h"
#include "mg_internal.h"
void stubQ (U64 len, MgChainQ *q, MgChainM *m);
int main (int argc, char **argv)
{ const char *path = argv[1], *want = argv[2]; (void) argc;
  static char *refNames[3] = { "chr1", "a_longer_sequence_name", "x" };
  MgReference ref; memset (&ref, 0, sizeof (ref)); ref.names = refNames; ref.nSeq = 3;
  Modset ms; memset (&ms, 0, sizeof (ms)); ref.ms = &ms;
  FILE *out = fopen (path, "w"), *exp = fopen (want, "w");
  fputs ("first line\n", out); fputs ("first line\n", exp);
  MgQueryPipe *p = mgQueryPipeOpen (&ref, out);
  unsigned long long x = 88172645463325252ull; long long readNo = 0;
  for (int b = 0 ; b < 60 ; ++b)
    { x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      const int n = b % 7 == 3 ? 20000 + (int) (x % 30000) : 1 + (int) (x % 3000);
      U64 *off = malloc (((size_t) n + 1) * 8), *idOff = malloc ((size_t) n * 8); char *idBytes = malloc ((size_t) n * 24), *ip = idBytes;
      off[0] = 0;
      for (int r = 0 ; r < n ; ++r)
        { x ^= x << 13; x ^= x >> 7; x ^= x << 17;
          off[r + 1] = off[r] + (r % 97 == 0 ? 0 : x % 400);
          idOff[r] = (U64) (ip - idBytes); ip += sprintf (ip, "read_%lld%s", readNo++, r % 5 ? "" : "/1") + 1;
        }
      for (int r = 0 ; r < n ; ++r)                       /* what modmap.c:208-210,262-266 print */
        { MgChainQ q; MgChainM m[4]; const U64 len = off[r + 1] - off[r]; stubQ (len, &q, m);
          fprintf (exp, "Q\t%s\t%llu\t%d miss, %d copy1, %d copy2, %d multi, %.2f hit\n", idBytes + idOff[r], (unsigned long long) len, (int) q.missed, (int) q.copy1, (int) q.copy2,
                   (int) q.copyM, ((int) q.nSeeds - (int) q.missed) / (double) (int) q.nSeeds);
          for (U32 k = 0 ; k < q.nM ; ++k)
            fprintf (exp, "M\t%s\t%d\t%d\t%d\t%s\t%d\t%d\t%d %d\t%.2f\t%.2f\n", idBytes + idOff[r], (int) m[k].pos0, (int) m[k].posN, (int) (m[k].posN - m[k].pos0), refNames[m[k].id0],
                     (int) m[k].off0, (int) m[k].offN, m[k].n1, m[k].n2, (m[k].n1 + m[k].n2) / (double) m[k].span, m[k].n1 / (double) (int) q.copy1);
        }
      if (mgQueryPipePush (p, (const U32 *) off, off[n], off, n, idBytes, idOff)) { fprintf (stderr, "push failed\n"); return 1; }
      memset (idBytes, '#', (size_t) (ip - idBytes)); free (idBytes); free (idOff); free (off);      /* the parser moves on */
    }
  mgQueryPipeClose (p);
  fputs ("last line\n", out); fputs ("last line\n", exp);
  fclose (out); fclose (exp);
  mgQueryReleaseBuffers ();
  printf ("qpipe ok: %lld reads\n", readNo); return 0; }
EOS
for san in thread address,undefined; do
  gcc -g -O1 -fsanitize=$san -fno-sanitize-recover=all -std=gnu11 -I$R/include -I$R/modimizer_amd/csrc -o $D/t $D/main.c $D/stubs.c \
      $R/modimizer_amd/csrc/mg_callers.c $R/modimizer_amd/csrc/mg_pgzip.c $R/modimizer_amd/csrc/mg_knobs.c -lpthread -lm -lz
  $D/t $D/out.txt $D/want.txt
  cmp $D/out.txt $D/want.txt && echo "  -fsanitize=$san: output identical ($(wc -l < $D/out.txt) lines)"
done
