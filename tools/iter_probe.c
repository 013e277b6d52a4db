/* tools/iter_probe.c -- dev: what one modRCiterator call costs by read length, on the host leg (mgIterScanHost), through the
 * kernel (MODGPU_ITER_HOST_BELOW=0) and in the compiled reference (oracle/_ref/libmodref.so when present): the crossover
 * mg_host.c's MG_ITER_HOST_BELOW_DEFAULT is set from.
 *   gcc -O2 -I include -o /tmp/iter_probe tools/iter_probe.c -Lmodimizer_amd -lmodgpu -Wl,-rpath,$PWD/modimizer_amd -ldl */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <dlfcn.h>
#include "modgpu.h"

static double now (void) { struct timespec t; clock_gettime (CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

typedef SeqhashRCiterator *(*IterFn) (Seqhash *, char *, int);
typedef bool (*NextFn) (SeqhashRCiterator *, U64 *, int *, bool *);

static double loop (IterFn mk, NextFn nx, Seqhash *sh, char *bases, int len, int nReads, U64 *sum)
{
  double t0 = now ();
  for (int r = 0 ; r < nReads ; ++r)
    { SeqhashRCiterator *si = mk (sh, bases + (size_t) r * len, len);
      U64 km; int pos; bool isF;
      while (nx (si, &km, &pos, &isF)) *sum += km + pos;
      free (si->hashBuf); free (si->fBuf); free (si);          /* seqhash.h:54-55, as the reference's callers do */
    }
  return (now () - t0) / nReads * 1e6;
}

int main (int argc, char **argv)
{
  int k = argc > 1 ? atoi (argv[1]) : 21, w = argc > 2 ? atoi (argv[2]) : 64;
  int haveGpu = mgDeviceCount () > 0;
  void *ref = dlopen ("oracle/_ref/libmodref.so", RTLD_NOW | RTLD_LOCAL);
  IterFn rmk = ref ? (IterFn) dlsym (ref, "modRCiterator") : 0;
  NextFn rnx = ref ? (NextFn) dlsym (ref, "modRCnext") : 0;
  Seqhash *(*rcreate) (int, int, int) = ref ? (Seqhash * (*) (int, int, int)) dlsym (ref, "seqhashCreate") : 0;
  Seqhash *sh = seqhashCreate (k, w, 17), *rsh = rcreate ? rcreate (k, w, 17) : 0;
  static const int lens[] = { 50, 150, 500, 1000, 2000, 4000, 6000, 8000, 12000, 16000, 24000, 32000, 64000, 128000, 250000 };
  printf ("k %d w %d  device %s  reference %s\n%8s %12s %12s %12s   (us per modRCiterator + replay + destroy)\n", k, w,
          haveGpu ? "yes" : "no", ref ? "yes" : "no", "len", "host leg", "kernel", "reference");
  for (unsigned i = 0 ; i < sizeof (lens) / sizeof (lens[0]) ; ++i)
    { int len = lens[i], nReads = 20000000 / len; if (nReads > 20000) nReads = 20000; if (nReads < 20) nReads = 20;
      char *bases = malloc ((size_t) len * nReads);
      U64 x = 88172645463325252ull;
      for (size_t j = 0 ; j < (size_t) len * nReads ; ++j) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; bases[j] = (char) (x >> 62); }
      U64 s1 = 0, s2 = 0, s3 = 0; double tH = -1, tK = -1, tR = -1;
      if (haveGpu)
        { mgIterHostBelow ((1 << 30) - 1); loop (modRCiterator, modRCnext, sh, bases, len, nReads > 100 ? 100 : nReads, &s1); s1 = 0;
          tH = loop (modRCiterator, modRCnext, sh, bases, len, nReads, &s1);
          mgIterHostBelow (0); loop (modRCiterator, modRCnext, sh, bases, len, nReads > 100 ? 100 : nReads, &s2); s2 = 0;
          int nk = nReads > 4000 ? 4000 : nReads;
          tK = loop (modRCiterator, modRCnext, sh, bases, len, nk, &s2);
          if (nk != nReads) { s1 = 0; loop (modRCiterator, modRCnext, sh, bases, len, nk, &s1); }
          if (s1 != s2) { printf ("MISMATCH at len %d\n", len); return 1; }
        }
      else
        { double t0 = now ();
          for (int r = 0 ; r < nReads ; ++r) { U64 *b = mgIterScanHost (sh, bases + (size_t) r * len, len); s1 += b[0]; free (b); }
          tH = (now () - t0) / nReads * 1e6;
        }
      if (rmk) tR = loop (rmk, rnx, rsh, bases, len, nReads, &s3);
      printf ("%8d %12.2f %12.2f %12.2f\n", len, tH, tK, tR);
      free (bases);
    }
  return 0;
}
