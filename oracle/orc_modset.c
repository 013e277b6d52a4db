/* oracle/orc_modset.c — TEST INFRASTRUCTURE ONLY.
 * CPU restatement of the modset table (reference modset.c) and of the caller-side bookkeeping
 * that defines "sketch" and "hit list" (reference modutils.c:19-63, modmap.c:49-134,188-281).
 * Pinned against the compiled reference (oracle/_ref) and tests/golden/.
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

/* modset.c:15-31: 20 <= bits <= 34; default capacity (tableSize/4)-1; index zeroed,
 * depth/info zeroed, value left as allocated. */
OrcModset *orcModsetCreate (const OrcHasher *h, int bits, uint32_t size)
{
  if (bits < 20 || bits > 34) return 0;
  OrcModset *ms = (OrcModset *) calloc (1, sizeof (OrcModset));
  ms->hasher = *h;
  ms->tableBits = bits;
  ms->tableSize = (uint64_t) 1 << bits;
  ms->tableMask = ms->tableSize - 1;
  if (size >= (ms->tableSize >> 2)) { free (ms); return 0; }
  ms->size = size ? size : (uint32_t) ((ms->tableSize >> 2) - 1);
  ms->index = (uint32_t *) calloc (ms->tableSize, sizeof (uint32_t));
  ms->value = (uint64_t *) malloc ((size_t) ms->size * sizeof (uint64_t));
  ms->depth = (uint16_t *) calloc (ms->size, sizeof (uint16_t));
  ms->info  = (uint8_t *) calloc (ms->size, 1);
  return ms;
}

void orcModsetDestroy (OrcModset *ms)
{ if (!ms) return; free (ms->index); free (ms->value); free (ms->depth); free (ms->info); free (ms); }

/* modset.c:45-62: primary slot = hash & tableMask; on collision step by
 * ((hash >> tableBits) & tableMask) | 1 (computed once); empty slot ends the search;
 * insertion takes index ++max and the capacity test comes after the slot write. */
uint32_t orcModsetFind (OrcModset *ms, uint64_t kmer, int isAdd)
{
  uint64_t hash = orcHash (&ms->hasher, kmer);
  uint64_t slot = hash & ms->tableMask;
  uint64_t step = ((hash >> ms->tableBits) & ms->tableMask) | 1;
  uint32_t ix;
  while ((ix = ms->index[slot]) != 0 && ms->value[ix] != kmer)
    slot = (slot + step) & ms->tableMask;
  if (ix || !isAdd) return ix;
  ix = ++ms->max;
  ms->index[slot] = ix;
  if (ms->max >= ms->size) { ms->overflow = 1; --ms->max; ms->index[slot] = 0; return 0; }
  ms->value[ix] = kmer;
  return ix;
}

static void shrinkArrays (OrcModset *ms, uint32_t newSize)
{
  ms->value = (uint64_t *) realloc (ms->value, (size_t) newSize * sizeof (uint64_t));
  ms->depth = (uint16_t *) realloc (ms->depth, (size_t) newSize * sizeof (uint16_t));
  ms->info  = (uint8_t *) realloc (ms->info, (size_t) newSize);
  ms->size = newSize;
}

/* modset.c:36-43 */
int orcModsetPack (OrcModset *ms)
{
  if (ms->size == ms->max + 1) return 0;
  shrinkArrays (ms, ms->max + 1);
  return 1;
}

/* modset.c:64-77: rebuild keeping entries with min <= depth (< max when max != 0), in old
 * index order, carrying depth and info. */
void orcModsetDepthPrune (OrcModset *ms, int min, int max)
{
  uint32_t n = ms->max;
  ms->max = 0;
  memset (ms->index, 0, ms->tableSize * sizeof (uint32_t));
  for (uint32_t i = 1 ; i <= n ; ++i)
    if (ms->depth[i] >= min && (!max || ms->depth[i] < max))
      { uint32_t j = orcModsetFind (ms, ms->value[i], 1);
        ms->info[j] = ms->info[i];
        ms->depth[j] = ms->depth[i];
      }
}

/* modset.c:106-128: hashers must agree on (w,k,factor1); ms1 arrays grow to
 * min(max1+max2+1, tableSize/4 - 1); ms2 entries inserted in ms2 index order;
 * depth adds saturate at 65535; copy bits add and saturate at 3 while the other info bits
 * of the ms1 entry are cleared (info &= 0x3 then |= c). */
int orcModsetMerge (OrcModset *a, OrcModset *b)
{
  if (a->hasher.w != b->hasher.w || a->hasher.k != b->hasher.k ||
      a->hasher.factor1 != b->hasher.factor1) return 0;
  uint64_t newSize = (uint64_t) a->max + b->max + 1;
  if (newSize >= (a->tableSize >> 2)) newSize = (a->tableSize >> 2) - 1;
  { uint32_t old = a->size, nw = (uint32_t) newSize;
    uint64_t *v = (uint64_t *) malloc ((size_t) nw * sizeof (uint64_t));
    uint16_t *d = (uint16_t *) malloc ((size_t) nw * sizeof (uint16_t));
    uint8_t  *f = (uint8_t *) malloc ((size_t) nw);
    uint32_t keep = old < nw ? old : nw;
    memcpy (v, a->value, (size_t) keep * sizeof (uint64_t));
    memcpy (d, a->depth, (size_t) keep * sizeof (uint16_t));
    memcpy (f, a->info, (size_t) keep);
    /* the reference leaves the grown tail uninitialised; a fresh entry's depth/info are then
       whatever malloc returned.  Zero it here: the reference only reads it after `+=`. */
    if (nw > keep)
      { memset (d + keep, 0, (size_t) (nw - keep) * sizeof (uint16_t));
        memset (f + keep, 0, (size_t) (nw - keep));
      }
    free (a->value); free (a->depth); free (a->info);
    a->value = v; a->depth = d; a->info = f; a->size = nw;
  }
  for (uint32_t i = 1 ; i <= b->max ; ++i)
    { uint32_t j = orcModsetFind (a, b->value[i], 1);
      if (!j) return 0;
      uint32_t d = (uint32_t) a->depth[j] + b->depth[i];
      a->depth[j] = d > 0xffff ? 0xffff : (uint16_t) d;
      int c = (a->info[j] & 3) + (b->info[i] & 3);
      if (c > 3) c = 3;
      a->info[j] = (uint8_t) ((a->info[j] & 3) | c);
    }
  return 1;
}

/* modutils.c:53-63 */
void orcDepthHistogram (const OrcModset *ms, uint64_t *hist)
{
  memset (hist, 0, 65536 * sizeof (uint64_t));
  for (uint32_t i = 1 ; i <= ms->max ; ++i) ++hist[ms->depth[i]];
}

void orcDepthHistogramPrint (const OrcModset *ms, FILE *f)
{
  uint64_t *hist = (uint64_t *) malloc (65536 * sizeof (uint64_t));
  orcDepthHistogram (ms, hist);
  for (uint32_t d = 0 ; d < 65536 ; ++d)
    if (hist[d]) fprintf (f, "DP\t%u\t%u\n", d, (uint32_t) hist[d]);
  free (hist);
}

/* modset.c:130-153 (with seqhash.c:55-56 for the first line).
 * The histogram array in the reference starts 256 long and grows to maxDepth+1. */
void orcModsetSummary (OrcModset *ms, FILE *f)
{
  fprintf (f, "SH k %d  w/m %d  s %d\n", ms->hasher.k, ms->hasher.w, ms->hasher.seed);
  fprintf (f, "MS table bits %d size %llu number of entries %u",
           ms->tableBits, (unsigned long long) ms->tableSize, ms->max);
  if (!ms->max) { fputc ('\n', f); return; }
  uint64_t *hist = (uint64_t *) malloc (65536 * sizeof (uint64_t));
  orcDepthHistogram (ms, hist);
  uint32_t copy[4] = { 0, 0, 0, 0 }, top = 0;
  for (uint32_t i = 1 ; i <= ms->max ; ++i)
    { ++copy[ms->info[i] & 3]; if (ms->depth[i] > top) top = ms->depth[i]; }
  /* arrayMax(h): the reference's array only registers growth through array(), which is used
     when depth >= arrayMax; so arrayMax ends as maxDepth+1 (0 when every depth is 0 and ... the
     first element goes through array() too, giving arrayMax 1). */
  uint32_t hmax = top + 1;
  uint64_t sum = 0, tot = 0;
  for (uint32_t i = 0 ; i < hmax ; ++i)          /* i*arr(h,i,U32): 32-bit product (modset.c:144) */
    { sum += (uint32_t) hist[i]; tot += (uint32_t) (i * (uint32_t) hist[i]); }
  int64_t half = (int64_t) (tot / 2);
  uint32_t n50;
  for (n50 = 0 ; n50 < hmax ; ++n50)
    { /* i*arr(h,i,U32) is a 32-bit product in the reference */
      half -= (uint32_t) (n50 * (uint32_t) hist[n50]);
      if (half < 0) break;
    }
  fprintf (f, " total count %llu\nMS average depth %.1f N50 depth %u",
           (unsigned long long) tot, tot / (double) sum, n50);
  if (copy[0] < ms->max)
    fprintf (f, " copy0 %u copy1 %u copy2 %u copyM %u", copy[0], copy[1], copy[2], copy[3]);
  fputc ('\n', f);
  free (hist);
}

/* modset.c:79-88 + seqhash.c:41-44.  The Seqhash struct is written raw (80 bytes):
 * seed,k,w (3 ints) pad4 mask(8) shift1,shift2 (2 ints) factor1 factor2 patternRC[4]. */
int orcModsetWrite (OrcModset *ms, FILE *f)
{
  uint32_t n = ms->max + 1;
  if (fwrite ("MSHSTv2", 8, 1, f) != 1) return 0;
  if (fwrite (&ms->tableBits, sizeof (int), 1, f) != 1) return 0;
  if (fwrite (&n, sizeof (uint32_t), 1, f) != 1) return 0;
  if (fwrite ("SQHSHv2", 8, 1, f) != 1) return 0;
  struct { int seed, k, w, pad; uint64_t mask; int shift1, shift2;
           uint64_t factor1, factor2, patternRC[4]; } raw;
  memset (&raw, 0, sizeof (raw));
  raw.seed = ms->hasher.seed; raw.k = ms->hasher.k; raw.w = ms->hasher.w;
  raw.mask = ms->hasher.mask; raw.shift1 = ms->hasher.shift1; raw.shift2 = 2 * ms->hasher.k;
  raw.factor1 = ms->hasher.factor1; raw.factor2 = ms->hasher.factor2;
  for (int b = 0 ; b < 4 ; ++b) raw.patternRC[b] = (uint64_t) (3 - b) << (2 * (ms->hasher.k - 1));
  if (fwrite (&raw, sizeof (raw), 1, f) != 1) return 0;
  if (fwrite (ms->index, sizeof (uint32_t), ms->tableSize, f) != ms->tableSize) return 0;
  if (fwrite (ms->value, sizeof (uint64_t), n, f) != n) return 0;
  if (fwrite (ms->depth, sizeof (uint16_t), n, f) != n) return 0;
  if (fwrite (ms->info, 1, n, f) != n) return 0;
  return 1;
}

/* modutils.c:194-198 */
void orcModsetWriteText (const OrcModset *ms, FILE *f)
{
  fprintf (f, "modset bits %d size %d k %d w %d seed %d\n",
           ms->tableBits, (int) (ms->max + 1), ms->hasher.k, ms->hasher.w, ms->hasher.seed);
  for (uint32_t i = 1 ; i <= ms->max ; ++i)
    fprintf (f, "%d\t%s\t%d\t%d\n", (int) i, orcSeqString (ms->value[i], ms->hasher.k),
             (int) ms->depth[i], (int) ms->info[i]);
}

/* modutils.c:19-31: depth is a U16 that is bumped and, on wrap to 0, pinned at 65535. */
int64_t orcAddSequence (OrcModset *ms, const uint8_t *s, int64_t len)
{
  /* per-thread scratch that only grows: a malloc/free pair per read turns into mmap/munmap for
     long reads and serialises threads on the kernel's mm lock */
  static __thread uint64_t *km = 0;
  static __thread int64_t kmCap = 0;
  const OrcHasher *h = &ms->hasher;
  if (len < h->k) return 0;
  int64_t cap = len - h->k + 1;
  if (cap > kmCap) { free (km); kmCap = cap + cap / 2; km = (uint64_t *) malloc ((size_t) kmCap * sizeof (uint64_t)); }
  int64_t n = orcScanRead (h, s, len, km, 0, 0, cap);
  for (int64_t i = 0 ; i < n ; ++i)
    { uint32_t ix = orcModsetFind (ms, km[i], 1);
      if (!ix) break;
      uint16_t d = (uint16_t) (ms->depth[ix] + 1);
      ms->depth[ix] = d ? d : 0xffff;
    }
  return n;
}

int64_t orcScanMany (const OrcHasher *h, const uint8_t *bases, const int64_t *offsets, int64_t nReads,
                     OrcModset *ms)
{
  int64_t total = 0;
  for (int64_t r = 0 ; r < nReads ; ++r)
    { const uint8_t *s = bases + offsets[r];
      int64_t len = offsets[r + 1] - offsets[r];
      total += ms ? orcAddSequence (ms, s, len) : orcScanRead (h, s, len, 0, 0, 0, 0);
    }
  return total;
}

/**************** modmap.c restatement ****************/

/* modmap.c:49-64 */
OrcReference *orcReferenceCreate (OrcModset *ms, uint32_t size)
{
  if (!ms || !ms->size || !size) return 0;
  OrcReference *ref = (OrcReference *) calloc (1, sizeof (OrcReference));
  ref->ms = ms;
  ref->size = size;
  ref->depth = (uint32_t *) calloc (ms->max ? ms->max : ms->size, sizeof (uint32_t));
  ref->index = (uint32_t *) malloc ((size_t) size * sizeof (uint32_t));
  ref->offset = (uint32_t *) malloc ((size_t) size * sizeof (uint32_t));
  ref->id = (uint32_t *) malloc ((size_t) size * sizeof (uint32_t));
  return ref;
}

void orcReferenceDestroy (OrcReference *ref)
{
  if (!ref) return;
  free (ref->depth); free (ref->rev); free (ref->loc);
  free (ref->index); free (ref->offset); free (ref->id); free (ref);
}

/* modmap.c:106-118: each modimizer occurrence that has (or gets) a modset index is appended
 * as (index, pos, seqId) and bumps the per-index occurrence count. */
int orcReferenceAddSequence (OrcReference *ref, const uint8_t *s, int64_t len, int isAdd)
{
  const OrcHasher *h = &ref->ms->hasher;
  uint32_t id = (uint32_t) ref->nSeq++;
  ref->totLen += len;
  if (len < h->k) return 1;
  int64_t cap = len - h->k + 1;
  uint64_t *km = (uint64_t *) malloc ((size_t) cap * sizeof (uint64_t));
  int32_t *ps = (int32_t *) malloc ((size_t) cap * sizeof (int32_t));
  int64_t n = orcScanRead (h, s, len, km, ps, 0, cap);
  int ok = 1;
  for (int64_t i = 0 ; i < n ; ++i)
    { uint32_t ix = orcModsetFind (ref->ms, km[i], isAdd);
      if (!ix) continue;
      if (ref->max + 1 >= ref->size) { ok = 0; break; }        /* "reference size overflow" */
      ref->index[ref->max] = ix;
      ++ref->depth[ix];
      ref->offset[ref->max] = (uint32_t) ps[i];
      ref->id[ref->max] = id;
      ++ref->max;
    }
  free (km); free (ps);
  return ok;
}

/* modmap.c:120-133 then referencePack modmap.c:74-91 */
void orcReferenceFinish (OrcReference *ref, int isAdd)
{
  OrcModset *ms = ref->ms;
  ref->n1 = ref->n2 = ref->nM = 0;
  for (uint32_t i = 1 ; i <= ms->max ; ++i)
    { uint32_t d = ref->depth[i];
      if (d == 1) { ms->info[i] = (uint8_t) ((ms->info[i] & 0xfc) | 1); ++ref->n1; }
      else if (d == 2) { ms->info[i] = (uint8_t) ((ms->info[i] & 0xfc) | 2); ++ref->n2; }
      else { ms->info[i] |= 3; ++ref->nM; }
    }
  if (isAdd) orcModsetPack (ms);
  /* referencePack: trim arrays, loc = exclusive prefix sum of per-index counts, rev = the
     occurrences grouped by index in occurrence order. */
  ref->depth = (uint32_t *) realloc (ref->depth, (size_t) (ms->max + 1) * sizeof (uint32_t));
  uint32_t n = ref->max ? ref->max : 1;
  ref->index = (uint32_t *) realloc (ref->index, (size_t) n * sizeof (uint32_t));
  ref->offset = (uint32_t *) realloc (ref->offset, (size_t) n * sizeof (uint32_t));
  ref->id = (uint32_t *) realloc (ref->id, (size_t) n * sizeof (uint32_t));
  ref->size = ref->max;
  ref->rev = (uint32_t *) malloc ((size_t) n * sizeof (uint32_t));
  ref->loc = (uint32_t *) malloc ((size_t) (ms->max + 1) * sizeof (uint32_t));
  ref->loc[0] = 0;
  for (uint32_t i = 1 ; i <= ms->max ; ++i) ref->loc[i] = ref->loc[i - 1] + ref->depth[i - 1];
  memset (ref->depth, 0, (size_t) (ms->max + 1) * sizeof (uint32_t));
  for (uint32_t i = 0 ; i < ref->max ; ++i)
    { uint32_t ix = ref->index[i];
      ref->rev[ref->loc[ix] + ref->depth[ix]++] = i;
    }
}

/* One end-of-block test of modmap.c:232-241 (and its copy at :245-254). */
static int blockEnds (const OrcReference *ref, uint32_t loc, uint32_t loc0, uint32_t locN,
                      uint32_t i0, uint32_t iN, int checkUnset)
{
  if (checkUnset && !loc0) return 1;
  if (ref->id[loc] != ref->id[loc0]) return 1;
  int end = 0;
  if (loc0 < locN)
    { if (loc < locN) end = 1;
      int32_t d = (int32_t) (locN - loc0 - iN + i0);
      if (d > 50 || d < -50) end = 1;
    }
  else if (loc0 > locN)
    { if (loc > locN) end = 1;
      int32_t d = (int32_t) (loc0 - locN - iN + i0);
      if (d > 50 || d < -50) end = 1;
    }
  return end;
}

static void printM (const OrcReference *ref, FILE *f, const char *readName, const char **refNames,
                    const uint32_t *seedPos, uint32_t i0, uint32_t iN, uint32_t loc0, uint32_t locN,
                    int n1, int n2, int copy1)
{
  fprintf (f, "M\t%s\t%d\t%d\t%d\t%s\t%d\t%d\t%d %d\t%.2f\t%.2f\n",
           readName, (int) seedPos[i0], (int) seedPos[iN], (int) (seedPos[iN] - seedPos[i0]),
           refNames[ref->id[loc0]], (int) ref->offset[loc0], (int) ref->offset[locN],
           n1, n2, (n1 + n2) / (double) ((locN > loc0) ? (locN - loc0) : (loc0 - locN)),
           n1 / (double) copy1);
}

/* modmap.c:197-276 for one read.  Seeds include misses (index 0).  Chaining keeps the
 * reference's quirks: occurrence number 0 doubles as "no block open" (modmap.c:232), a block
 * is printed when it ends only if it holds more than two copy-1 seeds (:256), and the block
 * still open at the end of the read is printed only if it holds more than two copy-2 seeds
 * (:269). */
int64_t orcQueryRead (OrcReference *ref, const char *readName, const uint8_t *s, int64_t len,
                      const char **refNames, FILE *f,
                      uint32_t *seedIndex, uint32_t *seedPos, int64_t seedCap)
{
  OrcModset *ms = ref->ms;
  const OrcHasher *h = &ms->hasher;
  int64_t cap = len >= h->k ? len - h->k + 1 : 1;
  uint64_t *km = (uint64_t *) malloc ((size_t) cap * sizeof (uint64_t));
  int32_t *ps = (int32_t *) malloc ((size_t) cap * sizeof (int32_t));
  uint32_t *six = (uint32_t *) malloc ((size_t) cap * sizeof (uint32_t));
  uint32_t *spos = (uint32_t *) malloc ((size_t) cap * sizeof (uint32_t));
  int64_t n = orcScanRead (h, s, len, km, ps, 0, cap);
  int missed = 0, copy[4] = { 0, 0, 0, 0 };
  for (int64_t i = 0 ; i < n ; ++i)
    { uint32_t ix = orcModsetFind (ms, km[i], 0);
      six[i] = ix; spos[i] = (uint32_t) ps[i];
      if (ix) ++copy[ms->info[ix] & 3]; else ++missed;
      if (i < seedCap) { if (seedIndex) seedIndex[i] = ix; if (seedPos) seedPos[i] = spos[i]; }
    }
  if (f)
    fprintf (f, "Q\t%s\t%llu\t%d miss, %d copy1, %d copy2, %d multi, %.2f hit\n",
             readName, (unsigned long long) len, missed, copy[1], copy[2], copy[3],
             (n - missed) / (double) n);

  uint32_t loc0 = 0, locN = 0, i0 = 0, iN = 0;
  int n1 = 0, n2 = 0;
  for (int64_t i = 0 ; f && i < n ; ++i)
    { uint32_t ix = six[i];
      if (!ix || (ms->info[ix] & 3) == 3) continue;             /* misses and multi-copy ignored */
      uint32_t loc = ref->rev[ref->loc[ix]];
      int is1 = (ms->info[ix] & 3) == 1;
      int end = blockEnds (ref, loc, loc0, locN, i0, iN, 1);
      if (end && loc0 && !is1)                                   /* try the second copy */
        { loc = ref->rev[ref->loc[ix] + 1];
          end = blockEnds (ref, loc, loc0, locN, i0, iN, 0);
        }
      if (end)
        { if (n1 > 2) printM (ref, f, readName, refNames, spos, i0, iN, loc0, locN, n1, n2, copy[1]);
          n1 = n2 = 0; loc0 = loc; i0 = (uint32_t) i;
        }
      if (is1) ++n1; else ++n2;
      locN = loc; iN = (uint32_t) i;
    }
  if (f && n2 > 2) printM (ref, f, readName, refNames, spos, i0, iN, loc0, locN, n1, n2, copy[1]);
  free (km); free (ps); free (six); free (spos);
  return n;
}

/* ---------------------------------------------------------------------------------------------
 * First occurrences and counts of a stream of k-mers: what sequential modsetIndexFind (..., true) + ++depth over the
 * stream produce (modset.c:56-57: entry ++max at first sight; modutils.c:26), stated without a table layout: flag[i] = 1
 * where km[i] has not occurred before, and cntAt[i] = number of occurrences of km[i] in the whole stream (written at first
 * occurrences only).  Then value[1..] = km[flag] in order, depth[1..] = min (65535, cntAt[flag]).  The full-size parity
 * tests use it to pin every entry of a 1e8-entry modset in seconds: nThreads threads each own the k-mers of one hash
 * class (all read the whole stream; no sharing, no locks). */
#include <pthread.h>
typedef struct { const uint64_t *km; uint64_t n; int t, T; uint8_t *flag; uint32_t *cntAt; uint32_t *firstAt; int ok; } OrcFoJob;

static inline uint64_t orcFoMix (uint64_t x)
{ x ^= x >> 31; x *= 0x7fb5d329728ea185ull; x ^= x >> 27; x *= 0x81dadef4bc2dd44dull; x ^= x >> 33; return x; }

static void *orcFoWorker (void *arg)
{
  OrcFoJob *j = (OrcFoJob *) arg;
  uint64_t mine = 0;
  for (uint64_t i = 0 ; i < j->n ; ++i) if (orcFoMix (j->km[i]) % (uint64_t) j->T == (uint64_t) j->t) ++mine;
  uint64_t cap = 1024; while (cap * 7 < mine * 10 + 10) cap <<= 1;       /* load <= 0.7 even if every one of them is distinct */
  uint64_t *key = (uint64_t *) malloc (cap * 8);
  uint64_t *first = (uint64_t *) malloc (cap * 8);      /* index of the first occurrence + 1; 0 = empty slot */
  if (!key || !first) { free (key); free (first); j->ok = 0; return 0; }
  memset (first, 0, cap * 8);
  for (uint64_t i = 0 ; i < j->n ; ++i)
    { const uint64_t x = j->km[i], h = orcFoMix (x);
      if (h % (uint64_t) j->T != (uint64_t) j->t) continue;
      uint64_t s = (h / (uint64_t) j->T) & (cap - 1);
      while (first[s] && key[s] != x) s = (s + 1) & (cap - 1);
      if (!first[s]) { first[s] = i + 1; key[s] = x; j->flag[i] = 1; j->cntAt[i] = 1; }
      else ++j->cntAt[first[s] - 1];
      if (j->firstAt) j->firstAt[i] = (uint32_t) (first[s] - 1);
    }
  free (key); free (first);
  j->ok = 1;
  return 0;
}

/* flag[n] and cntAt[n] must be zeroed by the caller; returns the number of distinct k-mers, -1 on allocation failure.
   firstAt (may be 0; n < 2^32): for EVERY i the position of the first occurrence of km[i] -- with it, the index modsetIndexFind hands
   occurrence i is (number of flags up to firstAt[i]), without sorting anything (tests/fullsize_whole.py c3ref) */
int64_t orcFirstOccurrencesAt (const uint64_t *km, uint64_t n, int nThreads, uint8_t *flag, uint32_t *cntAt, uint32_t *firstAt);
int64_t orcFirstOccurrences (const uint64_t *km, uint64_t n, int nThreads, uint8_t *flag, uint32_t *cntAt)
{ return orcFirstOccurrencesAt (km, n, nThreads, flag, cntAt, 0); }
int64_t orcFirstOccurrencesAt (const uint64_t *km, uint64_t n, int nThreads, uint8_t *flag, uint32_t *cntAt, uint32_t *firstAt)
{
  if (nThreads < 1) nThreads = 1;
  if (nThreads > 64) nThreads = 64;
  OrcFoJob job[64]; pthread_t th[64];
  for (int t = 0 ; t < nThreads ; ++t)
    { job[t].km = km; job[t].n = n; job[t].t = t; job[t].T = nThreads; job[t].flag = flag; job[t].cntAt = cntAt; job[t].firstAt = firstAt; job[t].ok = 0;
      if (pthread_create (&th[t], 0, orcFoWorker, &job[t])) { orcFoWorker (&job[t]); th[t] = 0; }
    }
  int ok = 1;
  for (int t = 0 ; t < nThreads ; ++t) { if (th[t]) pthread_join (th[t], 0); ok &= job[t].ok; }
  if (!ok) return -1;
  int64_t u = 0;
  for (uint64_t i = 0 ; i < n ; ++i) u += flag[i];
  return u;
}

/* referencePack (modmap.c:74-91) as the reference writes it: from the modset index of every occurrence (index[n], values 1 .. U), depth[U + 1]
 * (occurrences per index), loc[U + 1] (loc[0] = 0, loc[i] = loc[i - 1] + depth[i - 1]: modmap.c:82-84) and rev[n] (the occurrences grouped by
 * index in occurrence order: modmap.c:86-90, `rev[loc[*ri] + depth[*ri]++] = i`).  One sequential pass each. */
void orcReferencePack (const uint32_t *index, uint64_t n, uint32_t U, uint32_t *depth, uint32_t *loc, uint32_t *rev)
{
  memset (depth, 0, ((size_t) U + 1) * sizeof (uint32_t));
  for (uint64_t i = 0 ; i < n ; ++i) ++depth[index[i]];                       /* modmap.c:113 */
  loc[0] = 0;
  for (uint32_t i = 1 ; i <= U ; ++i) loc[i] = loc[i - 1] + depth[i - 1];
  uint32_t *fill = (uint32_t *) calloc ((size_t) U + 1, sizeof (uint32_t));
  for (uint64_t i = 0 ; i < n ; ++i) { const uint32_t x = index[i]; rev[loc[x] + fill[x]++] = (uint32_t) i; }
  free (fill);
}

/* What modsetIndexFind (ms, kmer, false) (modset.c:45-62) returns for each of m k-mers, given the set's k-mers sorted
 * (sorted[n]) with the index each holds (idxOfSorted[n]): the index, or 0 for a k-mer that is not in the set.  Binary
 * searches on nThreads threads -- the full-size parity run of config 3's queries looks 1.56e8 seeds up (tests/fullsize_whole.py). */
typedef struct { const uint64_t *sorted; const uint32_t *idx; uint64_t n; const uint64_t *keys; uint64_t lo, hi; uint32_t *out; } OrcLookJob;
static void *lookWorker (void *arg)
{
  OrcLookJob *j = (OrcLookJob *) arg;
  for (uint64_t i = j->lo ; i < j->hi ; ++i)
    { const uint64_t k = j->keys[i];
      uint64_t a = 0, b = j->n;
      while (a < b) { const uint64_t mid = (a + b) >> 1; if (j->sorted[mid] < k) a = mid + 1; else b = mid; }
      j->out[i] = (a < j->n && j->sorted[a] == k) ? j->idx[a] : 0;
    }
  return 0;
}
void orcSortedLookupMany (const uint64_t *sorted, const uint32_t *idxOfSorted, uint64_t n, const uint64_t *keys, uint64_t m, int nThreads, uint32_t *out)
{
  if (nThreads < 1) nThreads = 1;
  if (nThreads > 64) nThreads = 64;
  OrcLookJob job[64]; pthread_t th[64];
  for (int t = 0 ; t < nThreads ; ++t)
    { job[t] = (OrcLookJob) { sorted, idxOfSorted, n, keys, m * (uint64_t) t / nThreads, m * (uint64_t) (t + 1) / nThreads, out };
      pthread_create (&th[t], 0, lookWorker, &job[t]);
    }
  for (int t = 0 ; t < nThreads ; ++t) pthread_join (th[t], 0);
}
