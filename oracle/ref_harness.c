/* oracle/ref_harness.c — TEST INFRASTRUCTURE ONLY.
 * Timing harness around the UNMODIFIED reference (compiled from /root/reference by oracle/Makefile
 * into oracle/_ref/ref_bench): the CPU baseline bench.py reports next to the GPU numbers.
 * It calls the reference's own modRCiterator/modRCnext/modsetIndexFind; the only logic restated
 * here is the caller loop of modutils.c:19-31 (a static function there).
 *
 * usage: ref_bench <sample.bin> k w seed tableBits
 *   sample.bin = u64 nReads, u64 totalBases, i64 offsets[nReads+1], u8 bases[totalBases] (0..3)
 * prints one JSON line.
 */
#include "modset.h"
#include <time.h>

static double now (void)
{ struct timespec t; clock_gettime (CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main (int argc, char **argv)
{
  if (argc < 6) { fprintf (stderr, "usage: ref_bench sample.bin k w seed bits\n"); return 2; }
  FILE *f = fopen (argv[1], "rb");
  if (!f) { fprintf (stderr, "cannot open %s\n", argv[1]); return 2; }
  U64 nReads, total;
  if (fread (&nReads, 8, 1, f) != 1 || fread (&total, 8, 1, f) != 1) return 2;
  I64 *off = (I64 *) malloc ((nReads + 1) * sizeof (I64));
  char *bases = (char *) malloc (total + 1);
  if (fread (off, 8, nReads + 1, f) != nReads + 1 || fread (bases, 1, total, f) != total) return 2;
  fclose (f);
  int k = atoi (argv[2]), w = atoi (argv[3]), seed = atoi (argv[4]), bits = atoi (argv[5]);

  Seqhash *sh = seqhashCreate (k, w, seed);
  /* scan only */
  double t0 = now ();
  U64 nScan = 0, x = 0;
  for (U64 r = 0 ; r < nReads ; ++r)
    { SeqhashRCiterator *mi = modRCiterator (sh, bases + off[r], (int) (off[r + 1] - off[r]));
      U64 kmer; int pos;
      while (modRCnext (mi, &kmer, &pos, 0)) { ++nScan; x ^= kmer + pos; }
      seqhashRCiteratorDestroy (mi);
    }
  double tScan = now () - t0;

  /* scan + sketch (modutils.c:19-31), including modsetCreate as the reference pays it */
  t0 = now ();
  Modset *ms = modsetCreate (sh, bits, 0);
  U64 nHash = 0;
  for (U64 r = 0 ; r < nReads ; ++r)
    { SeqhashRCiterator *mi = modRCiterator (sh, bases + off[r], (int) (off[r + 1] - off[r]));
      U64 kmer; int pos;
      while (modRCnext (mi, &kmer, &pos, 0))
        { U32 index = modsetIndexFind (ms, kmer, true);
          U16 *di = &ms->depth[index]; ++*di; if (!*di) *di = U16MAX;
          ++nHash;
        }
      seqhashRCiteratorDestroy (mi);
    }
  double tSketch = now () - t0;
  printf ("{\"bases\": %llu, \"reads\": %llu, \"scan_s\": %.4f, \"sketch_s\": %.4f, \"scan_mbps\": %.3f, "
          "\"sketch_mbps\": %.3f, \"hashes\": %llu, \"entries\": %u, \"xor\": \"%llx\"}\n",
          (unsigned long long) total, (unsigned long long) nReads, tScan, tSketch,
          total / tScan / 1e6, total / tSketch / 1e6, (unsigned long long) nHash, ms->max,
          (unsigned long long) x);
  return nScan == nHash ? 0 : 1;
}
