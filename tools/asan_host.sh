#!/bin/bash
# dev: the scalar leg of modRCiterator (mg_host.c mgIterScanHost) and the replay under AddressSanitizer + UBSan on the CPU build (GPU
# sanitizers are not available on the pool): every k in a spread of 1..31 x d in {1, 2, 3, 4, 31, 64, 96, 1000, 2^20}, lengths 0..3000,
# with a homopolymer run.  The device hooks are stubbed (the kernel leg is not reached: the crossover is set above every length).
set -e
R=$(cd "$(dirname "$0")/.." && pwd); D=${TMPDIR:-/tmp}/modgpu_asan_host; mkdir -p $D
cat > $D/stubs.c <<'EOS'
#include "modgpu.h"
#include "mg_internal.h"
volatile int mgLiveDeviceModsets = 0;
void mgHookDestroy (Modset *ms) {} void mgHookHostRewrote (Modset *ms) {} void mgHookNeedHost (Modset *ms, int w) {} void mgHookNeedHostAll (Modset *ms, int w) {}
int mgHookHasDevice (Modset *ms) { return 0; } int mgHookMergeDevice (Modset *a, Modset *b) { return -1; } int mgHookMergeDeviceArrays (Modset *a, const U64 *v, const U16 *d, const U8 *i, U32 n) { return -1; } int mgHookPruneDevice (Modset *m, int a, int b) { return -1; }
int mgIterScan (Seqhash *sh, const char *s, int len, U64 **blk) { return -1; } int mgIterRequireDevice (void) { return 0; } void mgIterReleaseBuffers (void) {}
int mgIterMinScan (Seqhash *sh, const char *s, int len, U64 **rec, U64 *n) { return -1; } const char *mgLastError (void) { return "stub"; }
EOS
cat > $D/main.c <<'EOS'
#include <stdio.h>
#include <stdlib.h>
#include "modgpu.h"
int main (void)
{ unsigned long long x = 88172645463325252ull, tot = 0;
  int ks[] = { 1, 2, 5, 15, 16, 17, 19, 21, 27, 31 }, ws[] = { 1, 2, 3, 4, 31, 64, 96, 1000, 1 << 20 };
  for (unsigned a = 0 ; a < 10 ; ++a) for (unsigned b = 0 ; b < 9 ; ++b)
    { Seqhash *sh = seqhashCreate (ks[a], ws[b], 17);
      for (int len = 0 ; len < 3000 ; len += (len < 70 ? 1 : 97))
        { char *s = malloc (len + 1);
          for (int i = 0 ; i < len ; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; s[i] = (char) (x >> 62); }
          if (len > 200) for (int i = 50 ; i < 150 ; ++i) s[i] = 0;
          mgIterHostBelow (1 << 20);
          SeqhashRCiterator *it = modRCiterator (sh, s, len);
          U64 km; int pos; bool f;
          while (modRCnext (it, &km, &pos, &f)) tot += km + pos + f;
          mgSeqhashRCiteratorDestroy (it); free (s);
        }
      mgSeqhashDestroy (sh);
    }
  printf ("asan_host ok %llu\n", tot); return 0; }
EOS
gcc -g -O1 -fsanitize=address,undefined -fno-sanitize-recover=all -std=gnu11 -I$R/include -I$R/modimizer_amd/csrc -o $D/t $D/main.c $D/stubs.c \
    $R/modimizer_amd/csrc/mg_host.c $R/modimizer_amd/csrc/mg_knobs.c -lpthread -lm
$D/t
