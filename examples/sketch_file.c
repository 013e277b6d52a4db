/* examples/sketch_file.c — what `modutils -c B k w s -a reads.fa [-a more.fa] -H hist -wt dump` does
 * (modutils.c:139-157,220-224,240-244,191-199), written against include/modgpu.h in plain C:
 * the file is parsed by the library's thread pool, scanned and sketched on the GPU.
 *
 *   gcc -O2 -I include examples/sketch_file.c -o sketch_file -L modimizer_amd -lmodgpu \
 *       -Wl,-rpath,$PWD/modimizer_amd -Wl,-rpath,/opt/rocm/lib
 *   ./sketch_file 20 21 64 17 hist.txt dump.txt reads.fa [more.fa ...]
 */
#include <stdio.h>
#include <stdlib.h>
#include "modgpu.h"

int main (int argc, char **argv)
{
  if (argc < 8)
    { fprintf (stderr, "usage: %s <table bits> <k> <w> <seed> <hist out> <dump out> <reads.fa> [...]\n", argv[0]); return 2; }
  Seqhash *sh = seqhashCreate (atoi (argv[2]), atoi (argv[3]), atoi (argv[4]));      /* modutils.c:146 */
  Modset *ms = modsetCreate (sh, atoi (argv[1]), 0);                                   /* modutils.c:150 */
  seqhashReport (sh, stdout);
  for (int i = 7 ; i < argc ; ++i)
    { if (mgAddSequenceFile (ms, argv[i], stdout))                                     /* modutils.c:220-223 */
        { fprintf (stderr, "FATAL ERROR: failed to open sequence file %s (%s)\n", argv[i], mgLastError ()); return 1; }
      modsetSummary (ms, stdout);
    }
  FILE *f = fopen (argv[5], "w");
  if (!f) { perror (argv[5]); return 1; }
  mgDepthHistogram (ms, f);                                                            /* modutils.c:240-244 */
  fclose (f);
  if (!(f = fopen (argv[6], "w"))) { perror (argv[6]); return 1; }
  mgModsetWriteText (ms, f);                                                           /* modutils.c:191-199 */
  fclose (f);
  modsetDestroy (ms);
  mgSeqhashDestroy (sh);
  return 0;
}
