/* oracle/oracle.h — TEST INFRASTRUCTURE ONLY (not product code).
 *
 * CPU restatement of modimizer's seqhash + modset hot path, used solely as the
 * checker for the HIP implementation: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load or call anything under oracle/.
 *
 * Every function cites the reference file:line (paths into /root/reference) whose
 * behaviour it restates.  The restatement is pinned against the reference itself:
 * oracle/Makefile compiles the unmodified reference sources into oracle/_ref/ and
 * tests/test_oracle.py + tests/golden/ compare the two bit-for-bit.
 */
#ifndef MOD_ORACLE_H
#define MOD_ORACLE_H

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* seqhash.h:15-23 — hash parameters (only the fields the hash actually uses). */
typedef struct {
  int      seed, k, w;
  int      shift1;          /* 64 - 2k */
  uint64_t mask;            /* 2^(2k) - 1 */
  uint64_t factor1;         /* odd multiplier */
  uint64_t factor2;         /* drawn but never used by any hash (seqhash.c:32) */
} OrcHasher;

/* seqhash.c:20-37.  Returns 0 on success, -1 on the parameter errors the reference die()s on. */
int orcHasherInit (OrcHasher *h, int k, int w, int seed);

/* seqhash.h:58 */
static inline uint64_t orcHash (const OrcHasher *h, uint64_t x)
{ return (x * h->factor1) >> h->shift1; }

/* seqhash.c:154-196 (modRCiterator + modRCnext run to exhaustion over one read).
 * s[len] holds bases 0..3 one per byte.  Emits up to cap modimizers in increasing pos;
 * returns the total number found (may exceed cap; excess is not stored).
 * Any of kmer/pos/isF may be NULL. */
int64_t orcScanRead (const OrcHasher *h, const uint8_t *s, int64_t len,
                     uint64_t *kmer, int32_t *pos, uint8_t *isF, int64_t cap);

/* seqhash.c:83-152 (minimizerRCiterator + minimizerRCnext run to exhaustion).
 * Emits (hash,pos,isF) exactly as successive minimizerRCnext calls would. */
int64_t orcMinimizerRead (const OrcHasher *h, const uint8_t *s, int64_t len,
                          uint64_t *hash, int32_t *pos, uint8_t *isF, int64_t cap);

/* modset.h:17-28 */
typedef struct {
  OrcHasher hasher;
  int       tableBits;
  uint32_t  size;           /* capacity of value/depth/info */
  uint64_t  tableSize, tableMask;
  uint32_t *index;          /* tableSize entries, 0 = empty */
  uint64_t *value;
  uint16_t *depth;
  uint8_t  *info;
  uint32_t  max;            /* entries are 1..max */
  int       overflow;       /* set instead of die() when max >= size (modset.c:58) */
} OrcModset;

OrcModset *orcModsetCreate (const OrcHasher *h, int bits, uint32_t size);     /* modset.c:15-31 */
void       orcModsetDestroy (OrcModset *ms);
uint32_t   orcModsetFind (OrcModset *ms, uint64_t kmer, int isAdd);           /* modset.c:45-62 */
int        orcModsetPack (OrcModset *ms);                                     /* modset.c:36-43 */
void       orcModsetDepthPrune (OrcModset *ms, int min, int max);             /* modset.c:64-77 */
int        orcModsetMerge (OrcModset *a, OrcModset *b);                       /* modset.c:106-128 */
void       orcModsetSummary (OrcModset *ms, FILE *f);                         /* modset.c:130-153 */
int        orcModsetWrite (OrcModset *ms, FILE *f);                           /* modset.c:79-88 (+seqhash.c:41-44) */

/* modutils.c:19-31 — insert + saturating depth for every modimizer of one read; returns #hashes */
int64_t orcAddSequence (OrcModset *ms, const uint8_t *s, int64_t len);
/* modutils.c:53-63 — hist[65536] counts of depth[1..max] */
void    orcDepthHistogram (const OrcModset *ms, uint64_t *hist65536);
/* modutils.c:53-63 — "DP\t<depth>\t<count>\n" lines */
void    orcDepthHistogramPrint (const OrcModset *ms, FILE *f);
/* modutils.c:194-198 — "-wt" text dump */
void    orcModsetWriteText (const OrcModset *ms, FILE *f);

/* modmap.c:35-47 Reference object, modmap.c:93-134 build, :74-91 pack */
typedef struct {
  OrcModset *ms;
  uint32_t   size, max;
  uint32_t  *index, *offset, *id;
  uint32_t  *depth;         /* per modset index */
  uint32_t  *rev, *loc;
  int        nSeq;
  int64_t    totLen;
  uint32_t   n1, n2, nM;    /* copy-class counts printed at modmap.c:130 */
} OrcReference;

OrcReference *orcReferenceCreate (OrcModset *ms, uint32_t size);              /* modmap.c:49-64 */
void          orcReferenceDestroy (OrcReference *ref);
/* modmap.c:106-118 for one sequence (id = its ordinal) */
int           orcReferenceAddSequence (OrcReference *ref, const uint8_t *s, int64_t len, int isAdd);
/* modmap.c:120-133 copy classification + modsetPack + referencePack */
void          orcReferenceFinish (OrcReference *ref, int isAdd);

/* modmap.c:188-281 for one read: writes the "Q" line and any "M" lines to f.
 * seqName(i) naming is supplied by the caller through names[]. Returns number of seeds. */
int64_t orcQueryRead (OrcReference *ref, const char *readName, const uint8_t *s, int64_t len,
                      const char **refNames, FILE *f,
                      uint32_t *seedIndex, uint32_t *seedPos, int64_t seedCap);

/* modasm.c:30-57,79-86: the long-read set that readsetFileRead + invBuild leave behind */
typedef struct { int32_t len, nHit, nMiss, nCopy[4]; } OrcRead;
typedef struct {
  OrcModset *ms;
  int nReads, capReads;
  OrcRead  *reads;          /* [1..nReads]; entry 0 is burnt (modasm.c:95) */
  uint64_t *hitStart;       /* hits of read i: [hitStart[i], hitStart[i+1]) */
  uint32_t *hit;            /* modset index, top bit = forward */
  uint16_t *dx;             /* distance to the previous hit of the read */
  uint64_t  totHit, capHit;
  uint64_t *invStart;       /* [max+2]: reads holding mod i at invSpace[invStart[i] ..] (depth[i] of them, none when saturated) */
  uint32_t *invSpace;
} OrcReadset;
OrcReadset *orcReadsetCreate (OrcModset *ms);
void orcReadsetDestroy (OrcReadset *rs);
void orcReadsetBegin (OrcReadset *rs);                                           /* modasm.c:158 */
void orcReadsetAddRead (OrcReadset *rs, const uint8_t *s, int64_t len);          /* modasm.c:161-188 */
void orcReadsetFinish (OrcReadset *rs);                                          /* modasm.c:258-287 */
void orcReadsetStats (const OrcReadset *rs, FILE *f);                            /* modasm.c:193-253 */
int  orcReadsetWrite (const OrcReadset *rs, FILE *f);                            /* modasm.c:113-125 */

/* seqhash.c:198-206 */
const char *orcSeqString (uint64_t kmer, int len);

/* Deterministic synthetic data shared by tests/bench; not from the reference. */
uint64_t orcSplitmix64 (uint64_t x);
uint64_t orcXorshiftBases (uint64_t state, uint8_t *out, int64_t n);   /* SURVEY §8(c) known-answer reads */

/* Timed CPU baseline helper: scan (+ optional modset add) over nReads reads laid out by
 * offsets[nReads+1] in bases[]; returns total modimizers. Single thread. */
/* first occurrences and counts of a k-mer stream (what modset.c:56-57 + modutils.c:26 make of it), multi-threaded:
   flag[i] = 1 at first occurrences, cntAt[i] = occurrences of km[i] (at first occurrences); both zeroed by the caller */
int64_t orcFirstOccurrences (const uint64_t *km, uint64_t n, int nThreads, uint8_t *flag, uint32_t *cntAt);
int64_t orcFirstOccurrencesAt (const uint64_t *km, uint64_t n, int nThreads, uint8_t *flag, uint32_t *cntAt, uint32_t *firstAt);   /* + the first occurrence's position for every i */
void orcReferencePack (const uint32_t *index, uint64_t n, uint32_t U, uint32_t *depth, uint32_t *loc, uint32_t *rev);              /* modmap.c:74-91 */
/* whole-stream comparison of a given modimizer stream with the scan of every read of a piece, multi-threaded (orc_seqhash.c) */
int64_t orcScanCheckMany (const OrcHasher *h, const uint8_t *bases, const int64_t *offsets, int64_t nReads,
                          const int64_t *first, const uint64_t *km, const uint32_t *posF, int nThreads,
                          int64_t *firstBad, int64_t *nChecked);
void orcSortedLookupMany (const uint64_t *sorted, const uint32_t *idxOfSorted, uint64_t n, const uint64_t *keys, uint64_t m, int nThreads, uint32_t *out);   /* orc_modset.c */
int64_t orcScanMany (const OrcHasher *h, const uint8_t *bases, const int64_t *offsets, int64_t nReads,
                     OrcModset *msOrNull);

#ifdef __cplusplus
}
#endif
#endif
