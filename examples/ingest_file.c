/* examples/ingest_file.c — what `modasm -m src.mod -f reads.fa -S -w stem` does (modasm.c:1558-1581), written against include/modgpu.h in
 * plain C: the modset is read from its file (modsetRead, modset.c:90-104), every read of the FASTA / FASTQ file is scanned and looked up on
 * the GPU and its hit list kept (mgReadsetFileRead: readsetFileRead + invBuild, modasm.c:151-191,258-287 -- depth[] is rebuilt from these
 * reads), the statistics are the reference's lines (mgReadsetStats: modasm.c:193-253) and stem.mod + stem.readset its files
 * (mgReadsetWrite: modasm.c:108-126), which `modasm -r stem` reads back.
 *
 *   gcc -O2 -I include examples/ingest_file.c -o ingest_file -L modimizer_amd -lmodgpu -Wl,-rpath,$PWD/modimizer_amd -Wl,-rpath,/opt/rocm/lib
 *   ./ingest_file src.mod reads.fa [stem]
 */
#include <stdio.h>
#include <stdlib.h>
#include "modgpu.h"

int main (int argc, char **argv)
{
  if (argc < 3) { fprintf (stderr, "usage: %s <mod file> <reads.fa|.fq> [stem to write]\n", argv[0]); return 2; }
  FILE *f = mgFzOpen (argv[1], "r");                                                   /* -m: fzopen, gzip or plain (modasm.c:1559) */
  if (!f) { fprintf (stderr, "FATAL ERROR: failed to open mod file %s\n", argv[1]); return 1; }
  Modset *ms = modsetRead (f); fclose (f);
  if (ms->max >= 0x80000000u) { fprintf (stderr, "FATAL ERROR: too many entries in modset\n"); return 1; }      /* modasm.c:1562 */
  modsetSummary (ms, stdout);                                                          /* modasm.c:1563 */
  MgReadset *rs = mgReadsetCreate (ms);                                                /* -f (modasm.c:1566-1571) */
  if (mgReadsetFileRead (rs, argv[2])) { fprintf (stderr, "FATAL ERROR: failed to read %s: %s\n", argv[2], mgLastError ()); return 1; }
  mgReadsetStats (rs, stdout);                                                         /* -S */
  if (argc > 3) mgReadsetWrite (rs, argv[3]);                                          /* -w stem */
  mgReadsetDestroy (rs);
  modsetDestroy (ms);
  return 0;
}
