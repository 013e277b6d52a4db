#!/bin/bash
# dev: the parallel gzip writer and reader (mg_pgzip.c: worker threads, the ordered hand-over of members, the reader's window) under
# ThreadSanitizer and then AddressSanitizer + UBSan, CPU only: files of 0 bytes, 1 byte, one member exactly, many members written in small
# fwrites and in one large one, read back through mgGzipOpenRead in small and large freads, and decompressed by zlib's own gzread (what the
# reference's fzopen uses) for comparison.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); D=${TMPDIR:-/tmp}/modgpu_tsan_pgzip; mkdir -p $D
cat > $D/harness.c <<'EOS'
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include "modgpu.h"
static unsigned long long rs = 88172645463325252ull;
static unsigned rnd (void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (unsigned) (rs >> 33); }
int main (void)
{
  const size_t sizes[] = { 0, 1, 4095, (size_t) 16 << 20, ((size_t) 16 << 20) + 1, (size_t) 100 << 20, ((size_t) 67 << 20) + 12345 };
  for (unsigned c = 0 ; c < sizeof (sizes) / sizeof (sizes[0]) ; ++c)
    { const size_t n = sizes[c];
      unsigned char *a = (unsigned char *) malloc (n + 1), *b = (unsigned char *) malloc (n + 1);
      for (size_t i = 0 ; i < n ; ++i) a[i] = (i >> 12) & 1 ? (unsigned char) rnd () : (unsigned char) (i >> 16);      /* runs and noise in turn */
      const char *name = "/tmp/modgpu_tsan_pgzip/t.gz";
      FILE *f = mgGzipOpenWrite (name); if (!f) { puts ("open for writing failed"); return 1; }
      if (c & 1) { for (size_t at = 0 ; at < n ; ) { size_t k = 1 + rnd () % 300000; if (k > n - at) k = n - at; if (fwrite (a + at, 1, k, f) != k) return 2; at += k; } }
      else if (n && fwrite (a, 1, n, f) != n) return 2;
      if (fclose (f)) { puts ("close failed"); return 3; }
      f = mgGzipOpenRead (name); if (!f) { puts ("not recognised as this writer's file"); return 4; }
      size_t got = 0;
      if (c & 1) got = fread (b, 1, n + 1, f);
      else for (;;) { size_t k = 1 + rnd () % 700000; if (k > n + 1 - got) k = n + 1 - got; size_t r = fread (b + got, 1, k, f); got += r; if (r < k || got == n + 1) break; }
      fclose (f);
      if (got != n || memcmp (a, b, n)) { printf ("case %u: read back %zu of %zu bytes, or different\n", c, got, n); return 5; }
      gzFile z = gzopen (name, "r"); size_t zg = 0; int r;
      while ((r = gzread (z, b + zg, (unsigned) ((n + 1 - zg) > (1u << 30) ? (1u << 30) : (n + 1 - zg)))) > 0) zg += (size_t) r;
      gzclose (z);
      if (zg != n || memcmp (a, b, n)) { printf ("case %u: zlib's gzread gives %zu of %zu bytes, or different\n", c, zg, n); return 6; }
      printf ("case %u: %zu bytes ok\n", c, n);
      free (a); free (b);
    }
  puts ("PGZIP_SAN_OK");
  return 0;
}
EOS
for san in thread address,undefined; do
  echo "== -fsanitize=$san"
  gcc -O1 -g -fsanitize=$san -fno-omit-frame-pointer -pthread -I$R/include -I$R/modimizer_amd/csrc $D/harness.c $R/modimizer_amd/csrc/mg_pgzip.c $R/modimizer_amd/csrc/mg_knobs.c -o $D/h_$san -lz
  MODGPU_GZIP_THREADS=6 $D/h_$san 2>&1 | tail -12
done
