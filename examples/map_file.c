/* examples/map_file.c — what `modmap -K k -W w -S s -B bits -f ref.fa [-w stem] -q queries.fa` does (modmap.c:346-374), written against
 * include/modgpu.h in plain C: the reference FASTA is parsed, scanned, inserted and packed on the GPU (mgReferenceFastaRead: modmap.c:93-134 +
 * 74-91), the query file is parsed, scanned, looked up, tallied and chained there (mgQueryFile: modmap.c:188-281), and the files of `-w` are the
 * reference's own format (modmap.c:136-156), which its `-r` reads back.
 *
 *   gcc -O2 -I include examples/map_file.c -o map_file -L modimizer_amd -lmodgpu -Wl,-rpath,$PWD/modimizer_amd -Wl,-rpath,/opt/rocm/lib
 *   ./map_file 28 19 31 17 ref.fa queries.fa [stem]        (stem: also write stem.mod + stem.ref, then load them again and query from the copy)
 */
#include <stdio.h>
#include <stdlib.h>
#include "modgpu.h"

int main (int argc, char **argv)
{
  if (argc < 7)
    { fprintf (stderr, "usage: %s <table bits> <k> <w> <seed> <ref.fa> <queries.fa> [stem to write and re-read]\n", argv[0]); return 2; }
  const int bits = atoi (argv[1]), k = atoi (argv[2]), w = atoi (argv[3]), seed = atoi (argv[4]);
  Seqhash *sh = seqhashCreate (k, w, seed);                                            /* modmap.c:358 */
  printf ("  modmap initialised with k = %d, w = %d, random seed = %d\n", k, w, seed);  /* modmap.c:359-360 */
  Modset *ms = modsetCreate (sh, bits, 0);                                             /* modmap.c:361 */
  MgReference *ref = mgReferenceCreate (ms, 1 << 26);                                  /* modmap.c:362 */
  if (mgReferenceFastaRead (ref, argv[5], true, stdout)) { fprintf (stderr, "FATAL ERROR: %s\n", mgLastError ()); return 1; }
  if (argc > 7)
    { mgReferenceWrite (ref, argv[7]);                                                 /* -w stem */
      mgReferenceDestroy (ref); modsetDestroy (ms);
      ref = mgReferenceLoad (argv[7]);                                                 /* -r stem: the Modset comes with it */
      ms = ref->ms;
    }
  if (mgQueryFile (ref, argv[6], stdout)) { fprintf (stderr, "FATAL ERROR: %s\n", mgLastError ()); return 1; }      /* -q */
  mgReferenceDestroy (ref);
  modsetDestroy (ms);
  return 0;
}
