/* mg_host.c — layer 1 of libmodgpu: the reference's seqhash.h / modset.h API in plain C.
 *
 * Same signatures, struct layouts, messages and exit behaviour as the reference
 * (seqhash.c, modset.c, utils.c:19-30), so modmap/modutils-style callers link unchanged.
 * What differs is where the work happens:
 *   - modRCiterator runs ONE GPU scan over the read (mg_scan.hip) and modRCnext replays it;
 *   - the Modset host arrays are a mirror: batch inserts/lookups run on the device table
 *     (mg_table.hip) and the mgHook* calls below bring the host arrays up to date before any
 *     function here reads them.
 * Scalar modsetIndexFind works on the host arrays, as the transparent struct demands (callers
 * index ms->value/depth/info themselves: modutils.c:26, modmap.c:126-129).
 */
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include "modgpu.h"
#include "mg_internal.h"

/* utils.c:19-30 */
static void die (const char *format, ...)
{
  va_list args;
  va_start (args, format);
  fprintf (stderr, "FATAL ERROR: ");
  vfprintf (stderr, format, args);
  fprintf (stderr, "\n");
  va_end (args);
  exit (-1);
}

static void *xalloc (size_t n, int zero)
{
  void *p = zero ? calloc (n ? n : 1, 1) : malloc (n ? n : 1);
  if (!p) die ("memory allocation failure requesting %llu bytes", (unsigned long long) n);
  return p;
}

/* the Modset arrays are what the first sync from the device writes end to end: ask for huge pages (mg_knobs.c mgHugeHint); the memory
   stays what malloc () / calloc () returned, free ()-able as the reference's modsetDestroy frees it (modset.c:33-34) */
static void *xallocBig (size_t n, int zero) { void *p = xalloc (n, zero); mgHugeHint (p, n); return p; }

/* ------------------------------ seqhash ------------------------------ */

Seqhash *seqhashCreate (int k, int w, int seed)
{
  Seqhash *sh = (Seqhash *) xalloc (sizeof (Seqhash), 1);
  if (k < 1 || k >= 32) die ("seqhash k %d must be between 1 and 32\n", k);
  if (w < 1) die ("seqhash w %d must be positive\n", w);
  sh->k = k; sh->w = w; sh->seed = seed;
  sh->mask = ((U64) 1 << (2 * k)) - 1;
  sh->shift1 = 64 - 2 * k;
  sh->shift2 = 2 * k;
  /* glibc random() stream after srandom(seed); the high word is drawn first (seqhash.c:30-33) */
  srandom ((unsigned) seed);
  U64 hi = (U64) random (), lo = (U64) random ();
  sh->factor1 = (hi << 32) | lo | 1;
  hi = (U64) random (); lo = (U64) random ();
  sh->factor2 = (hi << 32) | lo | 1;
  for (int b = 0 ; b < 4 ; ++b) sh->patternRC[b] = (U64) (3 - b) << (2 * (k - 1));
  return sh;
}

void mgSeqhashDestroy (Seqhash *sh) { free (sh); }

void seqhashWrite (Seqhash *sh, FILE *f)
{
  if (fwrite ("SQHSHv2", 8, 1, f) != 1) die ("failed to write seqhash header");
  if (fwrite (sh, sizeof (Seqhash), 1, f) != 1) die ("failed to write seqhash");
}

Seqhash *seqhashRead (FILE *f)
{
  Seqhash *sh = (Seqhash *) xalloc (sizeof (Seqhash), 0);
  char name[8];
  if (fread (name, 8, 1, f) != 1) die ("failed to read seqhash header");
  if (memcmp (name, "SQHSHv2", 8)) die ("seqhash read mismatch");
  if (fread (sh, sizeof (Seqhash), 1, f) != 1) die ("failed to read seqhash");
  return sh;
}

void seqhashReport (Seqhash *sh, FILE *f)
{ fprintf (f, "SH k %d  w/m %d  s %d\n", sh->k, sh->w, sh->seed); }

char *seqString (U64 kmer, int len)
{
  static char buf[33];
  if (len > 32) len = 32;
  if (len < 0) len = 0;
  buf[len] = 0;
  for (int i = len ; i-- ; kmer >>= 2) buf[i] = "acgt"[kmer & 3];
  return buf;
}

/* Iterator facade.  Layout kept from seqhash.h:25-34; hashBuf is the free()-able block that
 * carries the replay records: [0] = n, then n k-mers, then n packed (pos | isF<<31) words.
 * iMin is the replay cursor. */
static SeqhashRCiterator *iterAlloc (Seqhash *sh, char *s, int len)
{
  SeqhashRCiterator *si = (SeqhashRCiterator *) xalloc (sizeof (SeqhashRCiterator), 1);
  si->sh = sh; si->s = s; si->sEnd = s + len;
  si->fBuf = (bool *) xalloc ((size_t) sh->w * sizeof (bool), 1);
  return si;
}

/* The latency switch of the per-read facade.  A synchronous call cannot hide a kernel launch: launch + poll of the
 * one-launch iterator kernel cost 13-15 us whatever the read's length, and a read shorter than the crossover is
 * scanned in less than that by one host core.  Reads below the crossover are therefore scanned right here, by the
 * library's own scalar loop -- the semantics of seqhash.c:60-79,154-196 restated: roll the forward k-mer and its
 * reverse complement, hash both (seqhash.h:58), the smaller hash picks the strand (ties -> reverse, seqhash.c:66-67),
 * keep the start when hash % w == 0 (seqhash.c:170,190) -- into the same replay block the kernel writes.  Bases are
 * taken modulo 4, as the packer takes them (mg_pack.c).  This is a dispatch by latency inside the scalar facade, not
 * a fallback: modRCiterator still needs a HIP device (mgIterRequireDevice) and every batch entry point runs on it. */
/* measured on the MI355X box (tools/iter_probe.c, profiles/r04_iter_probe.txt; us per call, host leg / kernel leg):
 *   k=21 w=64:  150 b 0.28 / 12.7   4 kb 4.4 / 13.7   12 kb 12.9 / 14.2   16 kb 17.3 / 14.4   64 kb 69 / 22   250 kb 267 / 53
 *   k=31 w=4:   150 b 0.63 / 15.9   4 kb 16 / 29      8 kb 31.7 / 32.1    12 kb 47.7 / 34.5   250 kb 1010 / 326
 *   k=19 w=31:  12 kb 19.2 / 19.7   16 kb 25.6 / 20.1
 * (the reference's own iterator: 0.71 us at 150 b, 56 us at 12 kb, 1144 us at 250 kb).  Dense selections (w < 16) cost the
 * scalar loop more per base (stores, mispredicted branches), so their crossover is lower. */
#define MG_ITER_HOST_BELOW_DEFAULT 12288
#define MG_ITER_HOST_BELOW_DENSE    8192
static int gIterHostBelow = -1;           /* -1: not looked up yet, -2: the defaults by w, >= 0: set (knob or mgIterHostBelow); relaxed atomics: iterators run on any thread */
static int iterHostBelow (const Seqhash *sh)
{
  int v = __atomic_load_n (&gIterHostBelow, __ATOMIC_RELAXED);
  if (v == -1)
    { const long kv = mgKnobs ()->iterHostBelow;             /* tuning knob: 0 = every read through the kernel */
      v = kv != MG_KNOB_UNSET ? (kv < 0 ? 0 : (int) kv) : -2;
      __atomic_store_n (&gIterHostBelow, v, __ATOMIC_RELAXED);
    }
  if (v >= 0) return v;
  return sh && sh->w < 16 ? MG_ITER_HOST_BELOW_DENSE : MG_ITER_HOST_BELOW_DEFAULT;
}
/* mgReloadKnobs () (mg_knobs.c) calls this: the crossover is looked up again, so a MODGPU_ITER_HOST_BELOW set between two calls counts */
void mgIterKnobsReloaded (void) { __atomic_store_n (&gIterHostBelow, -1, __ATOMIC_RELAXED); }

/* the crossover in bases; below < 0 asks (the value for sparse selections when the defaults are in force), below >= 0
   sets it for every hasher, MG_ITER_BELOW_DEFAULTS (1 << 30 and above) goes back to what the environment says (the knob, or the
   defaults by w).  Returns the value in force before the call. */
int mgIterHostBelow (int below)
{
  const int was = iterHostBelow (0);
  if (below >= (1 << 30)) __atomic_store_n (&gIterHostBelow, -1, __ATOMIC_RELAXED);
  else if (below >= 0) __atomic_store_n (&gIterHostBelow, below, __ATOMIC_RELAXED);
  return was;
}

/* the replay block {n, n k-mers, n pos | isF << 31} of one read; malloc()ed, exact size */
U64 *mgIterScanHost (const Seqhash *sh, const char *s, int len)
{
  const int k = sh->k;
  if (len < k)
    { U64 *blk = (U64 *) xalloc (16, 0); blk[0] = 0; return blk; }
  const U64 f1 = sh->factor1, mask = sh->mask;
  const int shift1 = sh->shift1, topShift = 2 * (k - 1);
  const U64 w = (U64) sh->w;
  /* w = wOdd << wShift: hash % w == 0 iff the low wShift bits are zero and (hash >> wShift) * wOdd^-1 mod 2^64
     is at most (2^64 - 1) / wOdd -- exact, no division in the loop */
  int wShift = 0; U64 wOdd = w;
  while (!(wOdd & 1)) { wOdd >>= 1; ++wShift; }
  U64 inv = wOdd;                                        /* Newton: five steps double 3 correct bits to 64 and more */
  for (int i = 0 ; i < 5 ; ++i) inv *= 2 - wOdd * inv;
  const U64 lim = ~(U64) 0 / wOdd, lowMask = ((U64) 1 << wShift) - 1;
  /* room for the expected number of modimizers and a quarter more (on the stack for short reads); a read that holds more -- a
     homopolymer that is a modimizer, d = 1 -- gets room for every start */
  const size_t most = (size_t) (len - k + 1);
  size_t cap = (size_t) len / (size_t) w + (size_t) len / (4 * (size_t) w) + 64;
  if (cap > most) cap = most;
  U64 stackKm[512]; U32 stackPf[512];
  U64 *km = cap <= 512 ? stackKm : (U64 *) xalloc (cap * 8, 0);
  U32 *pf = cap <= 512 ? stackPf : (U32 *) xalloc (cap * 4, 0);
  const unsigned char *b = (const unsigned char *) s;
  U64 h = 0, r = 0;
  for (int i = 0 ; i < k - 1 ; ++i)
    { const U64 x = b[i] & 3;
      h = (h << 2) | x;
      r = (r >> 2) | ((3 - x) << topShift);
    }
  size_t n = 0;
  for (int i = k - 1 ; i < len ; ++i)
    { const U64 x = b[i] & 3;
      h = ((h << 2) & mask) | x;
      r = (r >> 2) | ((3 - x) << topShift);
      const U64 hF = (h * f1) >> shift1, hR = (r * f1) >> shift1;
      const int isF = hF < hR;
      const U64 hash = isF ? hF : hR;
      if (!(hash & lowMask) && ((hash >> wShift) * inv) <= lim)
        { if (n == cap)
            { U64 *nk = (U64 *) xalloc (most * 8, 0); U32 *np = (U32 *) xalloc (most * 4, 0);
              memcpy (nk, km, n * 8); memcpy (np, pf, n * 4);
              if (km != stackKm) { free (km); free (pf); }
              km = nk; pf = np; cap = most;
            }
          km[n] = isF ? h : r;
          pf[n] = (U32) (i - k + 1) | (isF ? MG_FWD_BIT : 0);
          ++n;
        }
    }
  U64 *blk = (U64 *) xalloc ((n + 1) * 8 + n * 4 + 8, 0);
  blk[0] = n;
  memcpy (blk + 1, km, n * 8);
  memcpy (blk + 1 + n, pf, n * 4);
  if (km != stackKm) { free (km); free (pf); }
  return blk;
}

SeqhashRCiterator *modRCiterator (Seqhash *sh, char *s, int len)
{
  SeqhashRCiterator *si = iterAlloc (sh, s, len);
  U64 *blk = 0;
  if (len < iterHostBelow (sh))
    { if (mgIterRequireDevice ()) die ("modRCiterator: %s", mgLastError ());
      blk = mgIterScanHost (sh, s, len);
    }
  else if (mgIterScan (sh, s, len, &blk)) die ("modRCiterator: GPU scan failed: %s", mgLastError ());
  si->hashBuf = blk;                       /* the replay block as the scan wrote it: no second copy */
  si->iMin = 0;
  si->isDone = (blk[0] == 0);
  return si;
}

bool modRCnext (SeqhashRCiterator *si, U64 *kmer, int *pos, bool *isF)
{
  if (si->isDone) return false;
  U64 n = si->hashBuf[0];
  const U64 *km = si->hashBuf + 1;
  const U32 *pf = (const U32 *) (km + n);
  U64 i = (U64) (unsigned) si->iMin;
  if (kmer) *kmer = km[i];
  if (pos) *pos = (int) (pf[i] & MG_POS_MASK);
  if (isF) *isF = (pf[i] & MG_FWD_BIT) != 0;
  si->h = km[i];
  if (++i >= n) si->isDone = true;
  si->iMin = (int) i;
  return true;
}

void mgSeqhashRCiteratorDestroy (SeqhashRCiterator *si)
{ free (si->hashBuf); free (si->fBuf); free (si); }

/* minimizerRCiterator / minimizerRCnext (seqhash.c:83-152): one GPU pass over the read
 * (mg_minimizer.hip: a wavefront walks the reference's chain of windows), replayed like the
 * modimizer iterator: the record block carries the returned hash values instead of k-mers. */
SeqhashRCiterator *minimizerRCiterator (Seqhash *sh, char *s, int len)
{
  SeqhashRCiterator *si = iterAlloc (sh, s, len);
  U64 *rec = 0, n = 0;
  if (mgIterMinScan (sh, s, len, &rec, &n)) die ("minimizerRCiterator: GPU scan failed: %s", mgLastError ());
  U64 *blk = (U64 *) xalloc ((size_t) (n + 1) * 8 + (size_t) n * 4 + 8, 0);
  blk[0] = n;
  if (n)
    { memcpy (blk + 1, rec, (size_t) n * 8);
      memcpy (blk + 1 + n, rec + n, (size_t) n * 4);
    }
  free (rec);
  si->hashBuf = blk;
  si->iMin = 0;
  si->isDone = (n == 0);
  return si;
}

bool minimizerRCnext (SeqhashRCiterator *si, U64 *u, int *pos, bool *isF)
{
  if (si->isDone) return false;
  U64 n = si->hashBuf[0];
  const U64 *hv = si->hashBuf + 1;
  const U32 *pf = (const U32 *) (hv + n);
  U64 i = (U64) (unsigned) si->iMin;
  if (u) *u = hv[i];
  if (pos) *pos = (int) (pf[i] & MG_POS_MASK);
  if (isF) *isF = (pf[i] & MG_FWD_BIT) != 0;
  if (++i >= n) si->isDone = true;
  si->iMin = (int) i;
  return true;
}

/* ------------------------------ modset ------------------------------ */

Modset *modsetCreate (Seqhash *sh, int bits, U32 size)
{
  if (bits < 20 || bits > 34) die ("table bits %d must be between 20 and 34", bits);
  Modset *ms = (Modset *) xalloc (sizeof (Modset), 1);
  ms->hasher = sh;
  ms->tableBits = bits;
  ms->tableSize = (U64) 1 << bits;
  ms->tableMask = ms->tableSize - 1;
  ms->index = (U32 *) xallocBig (ms->tableSize * sizeof (U32), 1);
  if (size >= (ms->tableSize >> 2)) die ("Modset size %u is too big for %d bits", size, bits);
  ms->size = size ? size : (U32) ((ms->tableSize >> 2) - 1);
  ms->value = (U64 *) xallocBig ((size_t) ms->size * sizeof (U64), 0);
  ms->depth = (U16 *) xallocBig ((size_t) ms->size * sizeof (U16), 1);
  ms->info = (U8 *) xallocBig ((size_t) ms->size, 1);
  return ms;
}

void modsetDestroy (Modset *ms)
{
  if (mgLiveDeviceModsets) mgHookDestroy (ms);
  free (ms->index); free (ms->value); free (ms->depth); free (ms->info); free (ms);
}

/* the reference's resize (utils.h): new arrays of newSize entries holding the first min (size, newSize) old ones, the rest of depth / info
   zero.  realloc () does it without touching what is kept (a block of this size shrinks or grows by remapping pages). */
static void regrow (Modset *ms, U32 newSize)
{
  const U32 keep = ms->size < newSize ? ms->size : newSize;
  U64 *v = (U64 *) realloc (ms->value, (size_t) (newSize ? newSize : 1) * sizeof (U64));
  U16 *d = (U16 *) realloc (ms->depth, (size_t) (newSize ? newSize : 1) * sizeof (U16));
  U8 *f = (U8 *) realloc (ms->info, (size_t) (newSize ? newSize : 1));
  if (!v || !d || !f) die ("memory allocation failure requesting %llu bytes", (unsigned long long) newSize * 11);
  if (newSize > keep)
    { memset (d + keep, 0, (size_t) (newSize - keep) * sizeof (U16));
      memset (f + keep, 0, (size_t) (newSize - keep));
    }
  ms->value = v; ms->depth = d; ms->info = f; ms->size = newSize;
}

bool modsetPack (Modset *ms)
{
  if (mgLiveDeviceModsets) mgHookNeedHostAll (ms, 0);
  if (ms->size == ms->max + 1) return false;
  regrow (ms, ms->max + 1);
  return true;
}

/* host-array form of the probe loop; assumes index[] and value[] are current */
static U32 hostFind (Modset *ms, U64 kmer, int isAdd)
{
  U64 hash = seqhash (ms->hasher, kmer);
  U64 at = hash & ms->tableMask;
  U64 hop = 0;
  U32 ix = ms->index[at];
  while (ix && ms->value[ix] != kmer)
    { if (!hop) hop = ((hash >> ms->tableBits) & ms->tableMask) | 1;
      at = (at + hop) & ms->tableMask;
      ix = ms->index[at];
    }
  if (!ix && isAdd)
    { ix = ms->index[at] = ++ms->max;
      if (ms->max >= ms->size) die ("hashTableSize %u is too small for %u", ms->size, ms->max);
      ms->value[ix] = kmer;
    }
  return ix;
}

U32 modsetIndexFind (Modset *ms, U64 kmer, int isAdd)
{
  if (mgLiveDeviceModsets) mgHookNeedHost (ms, 1);
  return hostFind (ms, kmer, isAdd);
}

void modsetDepthPrune (Modset *ms, int min, int max)
{
  if (mgLiveDeviceModsets && mgHookHasDevice (ms))
    { /* the set lives on the device: compact it there (same survivors, same order, same renumbering) */
      U32 before = ms->max;
      if (mgHookPruneDevice (ms, min, max)) die ("modsetDepthPrune on the device failed: %s", mgLastError ());
      fprintf (stderr, "  pruned Modset from %d to %d with min %d <= depth < max %d\n", before, ms->max, min, max);
      return;
    }
  U32 n = ms->max;
  ms->max = 0;
  memset (ms->index, 0, ms->tableSize * sizeof (U32));
  for (U32 i = 1 ; i <= n ; ++i)
    if (ms->depth[i] >= min && (!max || ms->depth[i] < max))
      { hostFind (ms, ms->value[i], 1);
        ms->info[ms->max] = ms->info[i];
        ms->depth[ms->max] = ms->depth[i];
      }
  fprintf (stderr, "  pruned Modset from %d to %d with min %d <= depth < max %d\n", n, ms->max, min, max);
  if (mgLiveDeviceModsets) mgHookHostRewrote (ms);
}

void modsetWrite (Modset *ms, FILE *f)
{
  if (mgLiveDeviceModsets) mgHookNeedHostAll (ms, 1);
  U32 n = ms->max + 1;
  if (fwrite ("MSHSTv2", 8, 1, f) != 1) die ("failed to write modset header");
  if (fwrite (&ms->tableBits, sizeof (int), 1, f) != 1) die ("failed to write bits");
  if (fwrite (&n, sizeof (U32), 1, f) != 1) die ("failed to write size");
  seqhashWrite (ms->hasher, f);
  if (fwrite (ms->index, sizeof (U32), ms->tableSize, f) != ms->tableSize) die ("fail write index");
  if (fwrite (ms->value, sizeof (U64), n, f) != n) die ("failed to write value");
  if (fwrite (ms->depth, sizeof (U16), n, f) != n) die ("failed to write depth");
  if (fwrite (ms->info, sizeof (U8), n, f) != n) die ("failed to write info");
}

Modset *modsetRead (FILE *f)
{
  char name[8];
  int bits; U32 n;
  if (fread (name, 8, 1, f) != 1) die ("failed to read modset header");
  if (memcmp (name, "MSHSTv2", 8)) { name[7] = 0; die ("bad modset header %s != MSHSTv2", name); }
  if (fread (&bits, sizeof (int), 1, f) != 1) die ("failed to read bits");
  if (fread (&n, sizeof (U32), 1, f) != 1) die ("failed to read size");
  Seqhash *sh = seqhashRead (f);
  Modset *ms = modsetCreate (sh, bits, n);
  if (fread (ms->index, sizeof (U32), ms->tableSize, f) != ms->tableSize) die ("failed read index");
  if (fread (ms->value, sizeof (U64), n, f) != n) die ("failed to read value");
  if (fread (ms->depth, sizeof (U16), n, f) != n) die ("failed to read depth");
  if (fread (ms->info, sizeof (U8), n, f) != n) die ("failed to read info");
  ms->max = n - 1;
  return ms;
}

bool modsetMerge (Modset *ms1, Modset *ms2)
{
  Seqhash *a = ms1->hasher, *b = ms2->hasher;
  if (a->w != b->w || a->k != b->k || a->factor1 != b->factor1) return false;
  const bool onDevice = mgLiveDeviceModsets && mgHookHasDevice (ms1);
  if (mgLiveDeviceModsets) { mgHookNeedHostAll (ms1, onDevice ? 0 : 1); mgHookNeedHostAll (ms2, 0); }
  U64 want = (U64) ms1->max + ms2->max + 1;
  if (want >= (ms1->tableSize >> 2)) want = (ms1->tableSize >> 2) - 1;
  regrow (ms1, (U32) want);
  if (onDevice)
    { if (mgHookMergeDevice (ms1, ms2)) die ("modsetMerge on the device failed: %s", mgLastError ());
      return true;
    }
  for (U32 i = 1 ; i <= ms2->max ; ++i)
    { U32 j = hostFind (ms1, ms2->value[i], 1);
      U32 d = (U32) ms1->depth[j] + ms2->depth[i];
      ms1->depth[j] = (U16) (d > 0xffff ? 0xffff : d);
      int c = (ms1->info[j] & 3) + (ms2->info[i] & 3);
      if (c > 3) c = 3;
      ms1->info[j] = (U8) ((ms1->info[j] & 3) | c);
    }
  if (mgLiveDeviceModsets) mgHookHostRewrote (ms1);
  return true;
}

/* modsetMerge where the second set is given as bare arrays (entries 1..n2 at value2[1..n2] etc.):
 * what a rank receives from its peers when per-GPU modsets are merged in rank order. */
bool mgModsetMergeArrays (Modset *ms1, U64 *value2, U16 *depth2, U8 *info2, U32 n2)
{
  Modset view;
  memset (&view, 0, sizeof (view));
  view.hasher = ms1->hasher; view.tableBits = ms1->tableBits; view.size = n2 + 1;
  view.tableSize = ms1->tableSize; view.tableMask = ms1->tableMask;
  view.value = value2; view.depth = depth2; view.info = info2; view.max = n2;
  return modsetMerge (ms1, &view);
}

/* the same with the second set's arrays in DEVICE memory (entry i at [i - 1], i = 1..n2) and ms1 on the device: what the root of
 * mgModsetMergeRankOrder holds after the peers' arrays arrived over xGMI.  false: ms1 has no device table (the caller then
 * brings the arrays to the host and takes mgModsetMergeArrays). */
bool mgModsetMergeDeviceArrays (Modset *ms1, const U64 *dValue2, const U16 *dDepth2, const U8 *dInfo2, U32 n2)
{
  if (!mgLiveDeviceModsets || !mgHookHasDevice (ms1)) return false;
  mgHookNeedHostAll (ms1, 0);
  U64 want = (U64) ms1->max + n2 + 1;
  if (want >= (ms1->tableSize >> 2)) want = (ms1->tableSize >> 2) - 1;
  regrow (ms1, (U32) want);
  if (mgHookMergeDeviceArrays (ms1, dValue2, dDepth2, dInfo2, n2)) die ("modsetMerge on the device failed: %s", mgLastError ());
  return true;
}

void modsetSummary (Modset *ms, FILE *f)
{
  if (mgLiveDeviceModsets) mgHookNeedHostAll (ms, 0);
  seqhashReport (ms->hasher, f);
  fprintf (f, "MS table bits %d size %llu number of entries %u",
           ms->tableBits, (unsigned long long) ms->tableSize, ms->max);
  if (!ms->max) { fputc ('\n', f); return; }
  U32 *h = (U32 *) xalloc (65536 * sizeof (U32), 1);
  U32 copy[4] = { 0, 0, 0, 0 }, bins = 0;
  for (U32 i = 1 ; i <= ms->max ; ++i)
    { U32 d = ms->depth[i];
      ++h[d]; if (d + 1 > bins) bins = d + 1;
      ++copy[ms->info[i] & 3];
    }
  U64 sum = 0, tot = 0;
  for (U32 i = 0 ; i < bins ; ++i) { sum += h[i]; tot += (U32) (i * h[i]); }   /* 32-bit products, modset.c:144 */
  int64_t half = (int64_t) (tot / 2);
  U32 n50 = 0;
  for ( ; n50 < bins ; ++n50) { half -= (U32) (n50 * h[n50]); if (half < 0) break; }
  fprintf (f, " total count %llu\nMS average depth %.1f N50 depth %u",
           (unsigned long long) tot, tot / (double) sum, n50);
  if (copy[0] < ms->max)
    fprintf (f, " copy0 %u copy1 %u copy2 %u copyM %u", copy[0], copy[1], copy[2], copy[3]);
  fputc ('\n', f);
  free (h);
}
