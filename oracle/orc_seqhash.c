/* oracle/orc_seqhash.c — TEST INFRASTRUCTURE ONLY.
 * CPU restatement of the seqhash scan (reference seqhash.c / seqhash.h); see oracle.h.
 * Written as closed-form array code, not as the reference's iterator state machine; pinned
 * bit-for-bit against the compiled reference (oracle/_ref) by tests/test_oracle.py.
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

/* seqhash.c:20-37: k in [1,31], w >= 1; factor1/factor2 drawn from glibc random() after
 * srandom(seed), high word first (the order gcc and clang both produce for seqhash.c:31). */
int orcHasherInit (OrcHasher *h, int k, int w, int seed)
{
  if (k < 1 || k >= 32 || w < 1) return -1;
  memset (h, 0, sizeof (*h));
  h->seed = seed; h->k = k; h->w = w;
  h->mask = ((uint64_t) 1 << (2 * k)) - 1;
  h->shift1 = 64 - 2 * k;
  srandom ((unsigned) seed);
  uint64_t a = (uint64_t) random (), b = (uint64_t) random ();
  h->factor1 = (a << 32) | b | 1;
  a = (uint64_t) random (); b = (uint64_t) random ();
  h->factor2 = (a << 32) | b | 1;
  return 0;
}

uint64_t orcSplitmix64 (uint64_t x)
{
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

/* The generator behind SURVEY §8(c)'s known answers: x^=x<<13; x^=x>>7; x^=x<<17; base = x>>62 */
uint64_t orcXorshiftBases (uint64_t state, uint8_t *out, int64_t n)
{
  for (int64_t i = 0 ; i < n ; ++i)
    { state ^= state << 13; state ^= state >> 7; state ^= state << 17;
      out[i] = (uint8_t) (state >> 62);
    }
  return state;
}

/* Canonical hash of the k-mer starting at each position.
 * seqhash.c:60-79: F is the k bases read left to right as base-4 digits, R is the reverse
 * complement; the canonical hash is the smaller of the two multiply-shift hashes and the
 * strand is forward only when hashF < hashR (ties -> reverse, seqhash.c:66-67). */
typedef struct { uint64_t F, R; } Pair;

static inline Pair firstPair (const OrcHasher *h, const uint8_t *s)
{
  Pair p = { 0, 0 };
  int top = 2 * (h->k - 1);
  for (int j = 0 ; j < h->k ; ++j)
    { p.F = (p.F << 2) | s[j];
      p.R = (p.R >> 2) | ((uint64_t) (3 - s[j]) << top);
    }
  return p;
}

static inline Pair nextPair (const OrcHasher *h, Pair p, uint8_t b)
{
  p.F = ((p.F << 2) & h->mask) | b;
  p.R = (p.R >> 2) | ((uint64_t) (3 - b) << (2 * (h->k - 1)));
  return p;
}

/* seqhash.c:154-196: every k-mer start 0..len-k is examined, including the last; a k-mer is
 * emitted when canonicalHash % w == 0; kmer is F or R according to the strand (seqhash.c:183);
 * pos is the 0-based start (seqhash.c:184, iMin counts advances). */
int64_t orcScanRead (const OrcHasher *h, const uint8_t *s, int64_t len,
                     uint64_t *kmer, int32_t *pos, uint8_t *isF, int64_t cap)
{
  if (len < h->k) return 0;                                   /* seqhash.c:162 */
  int64_t n = 0;
  uint64_t w = (uint64_t) h->w;
  Pair p = firstPair (h, s);
  for (int64_t i = 0 ; ; ++i)
    { uint64_t hF = orcHash (h, p.F), hR = orcHash (h, p.R);
      int fwd = hF < hR;
      uint64_t hash = fwd ? hF : hR;
      if (hash % w == 0)
        { if (n < cap)
            { if (kmer) kmer[n] = fwd ? p.F : p.R;
              if (pos) pos[n] = (int32_t) i;
              if (isF) isF[n] = (uint8_t) fwd;
            }
          ++n;
        }
      if (i + h->k >= len) break;
      p = nextPair (h, p, s[i + h->k]);
    }
  return n;
}

/* seqhash.c:83-152 restated over an explicit array of canonical hashes.
 *
 * The reference keeps a ring of w hashes.  Behaviour reproduced here (each item follows from
 * the cited lines):
 *  - the first window is k-mers 0..w-1; slots past the end of the read hold U64MAX
 *    (advanceHashRC returns U64MAX when the input is exhausted, seqhash.c:77-78);
 *  - hashBuf[0] is never written during set-up (seqhash.c:101 keeps the first hash only in a
 *    local), so if the first minimum is k-mer 0 the value returned for it is 0;
 *  - after returning the minimum at position p the next window is p+1..p+w (seqhash.c:128-139);
 *  - ties are broken by ring slot (position mod w), smallest slot first (seqhash.c:146-147);
 *  - when the new window runs off the end of the read, only a hash strictly smaller than the
 *    previous minimum is accepted, otherwise iteration ends (seqhash.c:142-149);
 *  - if all input had been consumed when a minimum is returned, iteration ends (seqhash.c:125).
 */
int64_t orcMinimizerRead (const OrcHasher *h, const uint8_t *s, int64_t len,
                          uint64_t *hashOut, int32_t *posOut, uint8_t *isFOut, int64_t cap)
{
  const uint64_t NONE = ~(uint64_t) 0;
  int k = h->k, w = h->w;
  if (len < k) return 0;                                       /* seqhash.c:94 */
  int64_t nk = len - k + 1;                                    /* number of k-mers */
  uint64_t *hv = (uint64_t *) malloc ((size_t) nk * sizeof (uint64_t));
  uint8_t  *fv = (uint8_t *) malloc ((size_t) nk);
  Pair p = firstPair (h, s);
  for (int64_t i = 0 ; i < nk ; ++i)
    { uint64_t hF = orcHash (h, p.F), hR = orcHash (h, p.R);
      fv[i] = hF < hR; hv[i] = fv[i] ? hF : hR;
      if (i + k < len) p = nextPair (h, p, s[i + k]);
    }
#define HV(q) ((q) < nk ? hv[q] : NONE)

  int64_t n = 0;
  /* first window: strict '<' scanning slots 1..w-1 against hash 0 => leftmost minimum */
  int64_t cur = 0; uint64_t curHash = hv[0];
  for (int64_t q = 1 ; q < w ; ++q) if (HV(q) < curHash) { curHash = HV(q); cur = q; }
  int64_t consumed = k + (w - 1);            /* bases consumed so far (si->s - start) */
  int first = 1;
  for (;;)
    { /* emit (seqhash.c:120-124) */
      uint64_t ret = (first && cur == 0) ? 0 : curHash;       /* hashBuf[0] quirk */
      if (n < cap)
        { if (hashOut) hashOut[n] = ret;
          if (posOut) posOut[n] = (int32_t) cur;
          if (isFOut) isFOut[n] = fv[cur];
        }
      ++n; first = 0;
      if (consumed >= len) break;                              /* seqhash.c:125 */
      /* refill: window becomes cur+1 .. cur+w; input consumed up to k-mer cur+w */
      consumed = k + cur + w;
      uint64_t bound; int haveFull = HV(cur + w) != NONE;      /* seqhash.c:142 */
      bound = haveFull ? NONE : ret;                           /* 'min = *u', seqhash.c:127 */
      int64_t best = -1; uint64_t bestHash = bound;
      /* scan ring slots 0..w-1 in slot order, strict '<' */
      for (int slot = 0 ; slot < w ; ++slot)
        { /* the position in cur+1..cur+w whose (pos mod w) == slot */
          int64_t base = (cur + 1) - ((cur + 1) % w);
          int64_t q = base + slot; if (q < cur + 1) q += w;
          uint64_t v = HV(q);
          if (v < bestHash) { bestHash = v; best = q; }
        }
      if (best < 0) break;                                     /* seqhash.c:148-149 */
      cur = best; curHash = bestHash;
    }
#undef HV
  free (hv); free (fv);
  return n;
}

/* seqhash.c:198-206 */
const char *orcSeqString (uint64_t kmer, int len)
{
  static char buf[33];
  static const char acgt[4] = { 'a', 'c', 'g', 't' };
  if (len > 32) len = 32;
  buf[len] = 0;
  for (int i = len - 1 ; i >= 0 ; --i) { buf[i] = acgt[kmer & 3]; kmer >>= 2; }
  return buf;
}

/* Whole-stream check for the full-size parity runs (tests/fullsize_whole.py): the reads of one piece of a batch
 * (bases[], offsets[nReads + 1] relative to bases[0]) are scanned by orcScanRead's loop on nThreads threads and compared,
 * modimizer by modimizer, with a stream given as arrays: read r's modimizers are km / posF [first[r], first[r + 1])
 * (posF = pos | isF << 31).  Nothing is sampled.  Returns the number of reads that differ; *firstBad = the lowest such
 * read (or -1), *nChecked = modimizers compared. */
#include <pthread.h>
typedef struct
{ const OrcHasher *h; const uint8_t *bases; const int64_t *offsets; int64_t nReads;
  const int64_t *first; const uint64_t *km; const uint32_t *posF;
  int64_t *next; int64_t bad, firstBad, checked;
} OrcCheckJob;

static int64_t checkRead (const OrcHasher *h, const uint8_t *s, int64_t len, const uint64_t *km, const uint32_t *posF, int64_t want)
{
  int64_t n = 0;
  if (len >= h->k)
    { const uint64_t w = (uint64_t) h->w;
      Pair p = firstPair (h, s);
      for (int64_t i = 0 ; ; ++i)
        { uint64_t hF = orcHash (h, p.F), hR = orcHash (h, p.R);
          int fwd = hF < hR;
          uint64_t hash = fwd ? hF : hR;
          if (hash % w == 0)
            { if (n >= want) return -1;
              if (km[n] != (fwd ? p.F : p.R) || posF[n] != ((uint32_t) i | ((uint32_t) fwd << 31))) return -1;
              ++n;
            }
          if (i + h->k >= len) break;
          p = nextPair (h, p, s[i + h->k]);
        }
    }
  return n == want ? n : -1;
}

static void *checkWorker (void *arg)
{
  OrcCheckJob *j = (OrcCheckJob *) arg;
  for (;;)
    { int64_t lo = __atomic_fetch_add (j->next, 64, __ATOMIC_RELAXED);
      if (lo >= j->nReads) break;
      int64_t hi = lo + 64 < j->nReads ? lo + 64 : j->nReads;
      for (int64_t r = lo ; r < hi ; ++r)
        { int64_t n = checkRead (j->h, j->bases + j->offsets[r], j->offsets[r + 1] - j->offsets[r],
                                 j->km + j->first[r], j->posF + j->first[r], j->first[r + 1] - j->first[r]);
          if (n < 0) { ++j->bad; if (j->firstBad < 0 || r < j->firstBad) j->firstBad = r; }
          else j->checked += n;
        }
    }
  return 0;
}

int64_t orcScanCheckMany (const OrcHasher *h, const uint8_t *bases, const int64_t *offsets, int64_t nReads,
                          const int64_t *first, const uint64_t *km, const uint32_t *posF, int nThreads,
                          int64_t *firstBad, int64_t *nChecked)
{
  if (nThreads < 1) nThreads = 1;
  if (nThreads > 64) nThreads = 64;
  OrcCheckJob job[64]; pthread_t th[64];
  int64_t next = 0;
  for (int t = 0 ; t < nThreads ; ++t)
    { job[t] = (OrcCheckJob) { h, bases, offsets, nReads, first, km, posF, &next, 0, -1, 0 };
      pthread_create (&th[t], 0, checkWorker, &job[t]);
    }
  int64_t bad = 0, fb = -1, chk = 0;
  for (int t = 0 ; t < nThreads ; ++t)
    { pthread_join (th[t], 0);
      bad += job[t].bad; chk += job[t].checked;
      if (job[t].firstBad >= 0 && (fb < 0 || job[t].firstBad < fb)) fb = job[t].firstBad;
    }
  if (firstBad) *firstBad = fb;
  if (nChecked) *nChecked = chk;
  return bad;
}
