/* mg_common.h — shared declarations for the HIP side of libmodgpu (gfx950 only). */
#ifndef MG_COMMON_H
#define MG_COMMON_H
#include "mg_knobs.h"

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "modgpu.h"

#define MG_WAVE 64

/* scan tiling: one workgroup walks tiles of the concatenated batch */
#define MG_SCAN_THREADS   256
#define MG_POS_PER_THREAD 64
#define MG_TILE_BASES     (64 * MG_POS_PER_THREAD)                /* 4096 k-mer starts per tile: one wavefront's */
#define MG_TILE_WORDS     (MG_TILE_BASES / 16)                    /* 256 packed words = 1 KiB  */

/* Hash parameters in the form the kernels use. */
struct MgHashParams {
  U64 factor1;
  U64 mask;        /* 2^(2k) - 1 */
  int k;
  int shift1;      /* 64 - 2k */
  U32 d;           /* the modulus sh->w */
  int dShift;      /* d = dOdd << dShift */
  U64 dOddInv;     /* inverse of dOdd mod 2^64 */
  U64 dOddLim;     /* floor((2^64-1) / dOdd) */
  /* the same test in 32-bit arithmetic, for hashes of at most 40 bits (k <= 20) and d < 2^15 (mgDivisibleOdd32): */
  U32 c24;         /* 2^24 mod dOdd */
  U32 inv32;       /* inverse of dOdd mod 2^32 */
  U32 lim32;       /* floor((2^32-1) / dOdd) */
  U32 small32;     /* 1 when the 32-bit test applies */
};

void mgSetError (const char *fmt, ...);
MgStatus mgHipFail (hipError_t e, const char *what);
MgStatus mgEnsureDevice (void);
MgHashParams mgMakeParams (const Seqhash *sh);

enum MgKernelId {
  MG_K_PACK = 0, MG_K_UNPACK, MG_K_TILE_FIRST_READ, MG_K_SCAN, MG_K_TABLE_INSERT, MG_K_TABLE_ASSIGN,
  MG_K_TABLE_FLAG, MG_K_TABLE_FIND, MG_K_TABLE_LOAD, MG_K_TABLE_EXPORT, MG_K_TABLE_HIST,
  MG_K_INDEX_REPLAY, MG_K_INDEX_FINISH, MG_K_SYNTH_GENOME, MG_K_SYNTH_READS, MG_K_MEMSET,
  MG_K_SEG_SCAN, MG_K_SEG_COMPACT, MG_K_PART, MG_K_PART_HIST, MG_K_PART_SCATTER, MG_K_RANK_COUNT, MG_K_RANK_SCAN, MG_K_BUCKET_DEDUP,
  MG_K_BUCKET_MERGE, MG_K_RANK_LOOKUP, MG_K_TABLE_FIND_SEG, MG_K_HOT_REDUCE, MG_K_BUCKET_FIND, MG_K_UNPART, MG_K_CHAIN, MG_K_CHAIN_RESOLVE, MG_K_COUNT
};
void mgProfBegin (int id, hipStream_t st);
void mgProfEnd (int id, hipStream_t st);
#define MG_LAUNCH(id, st, ...) do { mgProfBegin (id, st); hipLaunchKernelGGL (__VA_ARGS__); mgProfEnd (id, st); } while (0)

#define MG_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return mgHipFail (e_, #call); } while (0)

#ifdef __HIPCC__
/* Exact "x % d == 0" without a division (d = dOdd * 2^dShift):
 * low dShift bits zero, and (x >> dShift) * dOdd^-1 mod 2^64 <= floor((2^64-1)/dOdd). */
__device__ __forceinline__ bool mgDivisible (U64 x, const MgHashParams &p)
{
  if (x & (((U64) 1 << p.dShift) - 1)) return false;
  return ((x >> p.dShift) * p.dOddInv) <= p.dOddLim;
}

/* Exact "h % dOdd == 0" for h < 2^40 and odd dOdd < 2^15 in 32-bit arithmetic (seqhash.c:190 is `u % w`):
 * h = a 2^24 + b with a < 2^16, b < 2^24, so h = a c24 + b (mod dOdd) with c24 = 2^24 mod dOdd, and t = a c24 + b < 2^16 2^15 + 2^24
 * < 2^32 does not overflow; t is a multiple of the odd dOdd iff t dOdd^-1 mod 2^32 <= floor((2^32-1) / dOdd).
 * v_alignbit + v_and + v_mad_u32_u24 + v_mul_lo_u32 + v_cmp, in the place of a 64-bit low product (2 v_mul_lo_u32 + the double-cost
 * v_mad_u64_u32 + v_add3) and a 64-bit compare. */
__device__ __forceinline__ bool mgDivisibleOdd32 (U64 h, const MgHashParams &p)
{
  const U32 lo = (U32) h, hi = (U32) (h >> 32);
  const U32 a = __builtin_amdgcn_alignbit (hi, lo, 24);          /* h >> 24 */
  const U32 t = __umul24 (a, p.c24) + (lo & 0xffffffu);
  return t * p.inv32 <= p.lim32;
}

/* reverse complement of 16 bases in one word: reverse the bit order, swap the two bits of every base back, complement.
 * After the bit reversal the swap-and-complement is one three-input bit operation on (x << 1, x >> 1, 0x55555555):
 * out = mask ? ~(x >> 1) : ~(x << 1), truth table 0x27 */
__device__ __forceinline__ U32 mgRevComp16 (U32 w)
{
  const U32 x = __brev (w);
  return __builtin_amdgcn_bitop3_b32 (x << 1, x >> 1, 0x55555555u, 0x27);
}

/* reverse complement of a k-mer held in the low 2k bits: the two words swap places, each reverse-complemented */
__device__ __forceinline__ U64 mgRevComp (U64 f, int shift1)
{
  const U64 x = ((U64) mgRevComp16 ((U32) f) << 32) | mgRevComp16 ((U32) (f >> 32));
  return x >> shift1;
}
/* inclusive prefix sum over the 64 lanes of a wave with DPP only (no LDS traffic): Hillis-Steele inside the
 * rows of 16 (row_shr 1,2,4,8), then lane 15 of rows 0 and 2 into rows 1 and 3 (row_bcast:15), then lane 31
 * into rows 2 and 3 (row_bcast:31).  Lanes without a source add 0. */
__device__ __forceinline__ U32 mgWaveInclusiveSum (U32 v)
{
  v += (U32) __builtin_amdgcn_update_dpp (0, (int) v, 0x111, 0xf, 0xf, false);
  v += (U32) __builtin_amdgcn_update_dpp (0, (int) v, 0x112, 0xf, 0xf, false);
  v += (U32) __builtin_amdgcn_update_dpp (0, (int) v, 0x114, 0xf, 0xf, false);
  v += (U32) __builtin_amdgcn_update_dpp (0, (int) v, 0x118, 0xf, 0xf, false);
  v += (U32) __builtin_amdgcn_update_dpp (0, (int) v, 0x142, 0xa, 0xf, false);
  v += (U32) __builtin_amdgcn_update_dpp (0, (int) v, 0x143, 0xc, 0xf, false);
  return v;
}
/* The modset table's hash of a k-mer: a BIJECTION of the 2k-bit k-mer onto 2k-bit values (odd multipliers and
 * xor-shifts, all invertible modulo 2^(2k)), murmur-style.  Bucket = its top log2NB bits, home slot = its low bits, and
 * the table stores the mixed value itself (+1) as the key: being a bijection it identifies the k-mer, and a k-mer's
 * bucket digits are a prefix of its key -- which lets the partition passes drop the digits a bin already implies and
 * carry the first-occurrence ordinal in the freed bits of one 8-byte element (see mg_table.hip). */
struct MgGeom { U32 R; int log2NB; int kbits; };      /* kbits = 2k.  NB = 2^log2NB is a power of two (the partition passes and the scan's digit
                                                          counts read the bucket id's top bits); R, the slots of a bucket, is ANY number (a multiple
                                                          of 64): the table is sized by its entries, not to the next power of two (round 6) */
__device__ __forceinline__ U64 mgMixBits (U64 x, int b)        /* murmur-style bijection of b-bit values */
{
  if (b <= 32)                                                  /* the same function in 32-bit arithmetic: modulo 2^b only the multipliers' low words count */
    { U32 y = (U32) x;
      const U32 m32 = b >= 32 ? ~0u : (((U32) 1 << b) - 1);
      const int h32 = (b + 1) >> 1;
      y ^= y >> h32; y = (y * 0xed558ccdu) & m32;
      y ^= y >> h32; y = (y * 0x1a85ec53u) & m32;
      y ^= y >> h32;
      return y;
    }
  const U64 mask = b >= 64 ? ~0ull : (((U64) 1 << b) - 1);
  const int h = (b + 1) >> 1;
  x ^= x >> h; x = (x * 0xff51afd7ed558ccdull) & mask;
  x ^= x >> h; x = (x * 0xc4ceb9fe1a85ec53ull) & mask;
  x ^= x >> h;
  return x;
}
/* From 2k = 24 bits up the mix is built so that its top MG_MIX_TOP bits -- which hold the coarse digit of the bucket id,
 * what the first partition pass sorts by -- cost a multiply and a shift: the scan kernel counts them per modimizer on
 * the way (a full mix there would be two 64-bit multiplies in an instruction-bound kernel).  With x = (A : L), A the top
 * MG_MIX_TOP bits:  mix = (A ^ g(L)) : mixBits (L),  g(L) = top bits of a 32-bit multiplicative hash of L's low word.
 * A bijection: L comes back from its own mix, then A from the top part. */
#define MG_MIX_TOP 10
#define MG_MIX_MUL 0x9E3779B1u
__device__ __forceinline__ U64 mgMixK (U64 x, int b)
{
  if (b < 24) return mgMixBits (x, b);
  const int lb = b - MG_MIX_TOP;
  const U64 L = x & (((U64) 1 << lb) - 1), A = x >> lb;
  const U32 low = lb >= 32 ? (U32) L : ((U32) L & (((U32) 1 << lb) - 1));
  const U32 g = (low * MG_MIX_MUL) >> (32 - MG_MIX_TOP);
  return ((A ^ (U64) g) << lb) | mgMixBits (L, lb);
}
/* the top hiB <= MG_MIX_TOP bits of mgMixK (x, b), b >= 24, from the k-mer itself */
__device__ __forceinline__ U32 mgMixTopOfKmer (U64 x, int b, int hiB)
{
  const int lb = b - MG_MIX_TOP;
  const U32 low = lb >= 32 ? (U32) x : ((U32) x & (((U32) 1 << lb) - 1));
  return (U32) (x >> (b - hiB)) ^ ((low * MG_MIX_MUL) >> (32 - hiB));
}
__device__ __forceinline__ U32 mgBucketOfM (U64 m, const MgGeom &g)
{
  if (!g.log2NB) return 0u;
  const int s = g.kbits - g.log2NB;
  return s >= 0 ? (U32) (m >> s) : ((U32) m << (-s));
}
/* home slot: the low bits, stirred with the top part (k-mers that share their low bits -- and so the mix of them --
   differ there) */
__device__ __forceinline__ U32 mgHomeOfM (U64 m, const MgGeom &g)
{
  /* R is no power of two: the home is the high part of hash x R (a multiply-high instead of a mask), so the hash's TOP bits count --
     one more multiply spreads the low word's bits there (a bucket's k-mers share the bits of the bucket id that fall into that word) */
  U32 x = (U32) m;
  if (g.kbits >= 24) x ^= (U32) (m >> (g.kbits - MG_MIX_TOP)) * 0x9E5u;
  return __umulhi (x * 0x85EBCA6Bu, g.R);
}
/* the next slot of a probe sequence: linear, wrapping inside the bucket */
__device__ __forceinline__ U32 mgNextSlot (U32 at, U32 R) { return at + 1 == R ? 0u : at + 1; }
#endif /* __HIPCC__ */

/* The bucket id (log2NB bits) is split into a coarse digit (high bits, first partition pass) and a fine one. */
static inline void mgPartSplit (int log2NB, int *hiB, int *loB)
{ if (log2NB <= 9) { *hiB = log2NB; *loB = 0; } else { *loB = log2NB / 2; *hiB = log2NB - *loB; } }

/* The first partition pass needs the number of modimizers per coarse digit.  The scan's compaction kernel
 * reads every k-mer anyway, so it can count them on the way (its ALUs are idle: it is a copy): a caller that
 * knows the table geometry asks for that with a request; log2NB says which geometry the counts are for. */
#define MG_HIST_STRIDE 16      /* words between the counts of consecutive digits: one count per 64 bytes, thousands of workgroups add to each */
struct MgHistReq { int log2NB; int kbits; U32 *binCount; int hiB; };   /* hiB: bits of the digit counted, 0 = the build's own split (mgPartSplit); */   /* binCount: device, 512 x MG_HIST_STRIDE words, zeroed by the launcher; kbits = 2k (the table hash is over 2k bits) */

/* The scan's output BEFORE compaction: worker w's modimizers are segKmer[w * segCap + i], i < segCount[w], and
 * segStart[w] (nSegs + 1 entries, the last one the total) is the ordinal of its first one.  The modset build of a large
 * batch reads them from here (first partition pass and index assignment) instead of from a dense copy. */
struct MgSegSrc { const U64 *segKmer; const U64 *segCount; const U64 *segStart; U64 segCap; U32 nSegs; };


/* launchers implemented in the .hip files */
MgStatus mgLaunchPack (const U8 *dBases, U64 nBases, U32 *dWords, hipStream_t st);
MgStatus mgLaunchUnpack (const U32 *dWords, U64 nBases, U8 *dBases, hipStream_t st);
MgStatus mgLaunchScan (const MgHashParams &p, const U32 *dPacked, U64 totalBases,
                       const U64 *dReadOffsets, U32 nReads,
                       U64 *dKmer, U32 *dPosF, U32 *dReadId, U64 capacity,
                       U64 *dCount, void *dWork, hipStream_t st, const MgHistReq *hist = 0, MgSegSrc *lazy = 0);

U64      mgScanTiles (U64 totalBases);
size_t   mgScanInfoBytes (U64 totalBases);
MgStatus mgScanPrepare (const U64 *dReadOffsets, U32 nReads, U64 totalBases, void *dInfo, hipStream_t st);
size_t   mgScanRangeWorkBytes (U64 nTilesRange, U64 capacity);
/* lazy != 0: the dense k-mer array is NOT written (only the first partition digit is counted, when hist asks for it); *lazy
   describes the segments, and mgLaunchSegCompact makes the dense arrays from them later if they are wanted after all */
MgStatus mgLaunchSegCompact (const MgSegSrc &src, U64 *dKmer, U64 capacity, const U64 *dCount, hipStream_t st);
MgStatus mgLaunchScanRange (const MgHashParams &p, const U32 *dPacked, U64 totalBases,
                            const U64 *dReadOffsets, U32 nReads, const void *dInfo, U64 tile0, U64 tile1,
                            U64 *dKmer, U32 *dPosF, U32 *dReadId, U64 capacity,
                            U64 *dCount, void *dWork, hipStream_t st, const MgHistReq *hist = 0, MgSegSrc *lazy = 0);

/* the per-read iterator facade: one read of at most mgIterMaxBases () in one launch, replay block into (pinned host) out[] */
U64      mgIterMaxBases (void);
U64      mgIterSegEntries (void);
MgStatus mgLaunchIterScan (const MgHashParams &p, const U32 *dPacked, U64 totalBases, const U64 *dReadOff,
                           U64 *dSegKmer, U32 *dSegPosF, U64 *out, U64 capEntries, U64 *flag, U64 seq, hipStream_t st);

/* minimizers (seqhash.c:83-152) of every read: per-read counts -> exclusive scan in dReadStart[nReads+1] -> write */
MgStatus mgLaunchMinimizers (const MgHashParams &p, U32 w, const U32 *dPacked, const U64 *dReadOffsets, U32 nReads,
                             U64 *dHash, U32 *dPosF, U64 *dReadStart, U64 capacity, U64 *totalOut, hipStream_t st);

/* device modset table: NB = 2^log2NB buckets of R slots; see mg_table.hip */
struct MgSlot { U64 key; U32 ord; U32 cnt; };      /* 16 bytes; key = kmer+1, 0 = empty */
struct MgTable {
  MgSlot *slots; U64 nSlots;      /* nSlots = NB x R: what the table USES of its allocation */
  U64 capSlots;        /* slots allocated (grow-only: the geometry changes inside it without a hipMalloc) */
  U32 capNB;           /* entries of occ[] allocated */
  U32 R; int log2NB;
  int kbits;           /* 2k: width of the k-mers this table holds (and of its hash) */
  U32 *occ;            /* [NB] non-zero when the bucket may hold entries */
  U64 *value;          /* [size] device copy of ms->value */
  U16 *baseDepth;      /* [size] host depth at last sync */
  bool baseZero;       /* baseDepth is known to be all zero (fresh or cleared set): the histogram skips its random loads */
  U64 *liveHist;       /* device U64[65536]: depth histogram kept by the merge kernel while liveHistValid */
  bool liveHistValid;  /* the set was empty before its one bucketed add: liveHist is its depth histogram */
  U32 size;            /* capacity in entries (ms->size) */
  U32 max;             /* entries known to the device table */
  U32 syncedMax;       /* entries whose value[] the host already has */
  U64 *counters;       /* device U64[8]: 0 = new entries of the last add, 1 = bucket overflow */
  bool pendingDepth;   /* an add with depth counting ran since the counts were last folded into baseDepth / the host's depth[] */
  bool dirty;          /* buckets with occ == 0 hold undefined bytes (never zeroed): see mgTableClean */
  U64 *find8;          /* the partitioned lookups' copy of the table, 8 bytes a slot: (key's bits below the bucket id + 1) << 31 | index, 0 = empty (mg_table.hip
                          mgTableFind8); made when a lookup batch finds it missing or stale, the table itself being 16 bytes a slot with the counts */
  U64 find8Cap;        /* slots allocated for it */
  U64 find8Version, version;      /* version: bumped by whatever changes a slot's key or index (add, load, rehash, clear); find8 is the copy of find8Version */
  bool empty;          /* nothing has been put into the table since it was made or forgotten (every occ[] is zero): its geometry is free to change */
  U64 *pin;            /* four page-locked host words for small read-backs in the middle of an add (mgDevBuild) */
  int newPct;          /* new entries per 100 modimizers in the last bucketed add (a hint for the next one: mg_table.hip, markDup) */
  int loadPct;         /* slots are provided for entries * 100 / loadPct (0: 60).  A set that is only being built and counted
                          (mgAddReadsDevice) takes 75: its size is set from the OCCURRENCES of a batch, an upper bound of
                          the new entries, and the bucket images are the largest stream of the build; before lookups the
                          table is brought back to 60 (a probe that misses walks to the next empty slot) */
  int  maxLog2Slots;   /* tableBits - 1: the size at which load <= 0.5 for the largest legal set */
  U32  wantR;          /* most slots per bucket (4096: 64 KiB of LDS); a table's R lies between half of it and it */
  int  tightPct;       /* a set built by ONE bucketed add into an empty table is brought to this load once the dedup kernel has counted its
                          entries (mgTableAdd; 0: MG_TIGHT_PCT_DEFAULT): the bucket images the merge kernel streams back hold entries, not air */
};
MgStatus mgTableAlloc (MgTable *t, U64 slotsWanted, hipStream_t st);     /* geometry for that many slots (or a few more), empty table; allocates only when the capacity is short */
MgStatus mgTableEnsure (MgTable *t, U64 nIncoming, hipStream_t st);      /* grow (rehash) so that max+nIncoming fits at load <= 0.6 */
MgStatus mgTableClean (MgTable *t, hipStream_t st);                      /* zero the never-written buckets; dirty = false */
void     mgTableForget (MgTable *t, hipStream_t st);                     /* all buckets empty again (no memset of the slots) */
size_t   mgTableAddScratchBytes (const MgTable *t, U64 n);
bool     mgTableUseBuckets (const MgTable *t, U64 n);
MgStatus mgTableAdd (MgTable *t, const U64 *dKmer, U64 n, int withDepth, void *scratch, hipStream_t st,
                     const MgHistReq *counted = 0,    /* counted: first-pass digit counts of exactly these n k-mers, if for this geometry */
                     const MgSegSrc *segSrc = 0);     /* != 0 (and dKmer 0): the k-mers still sit in the scan's segments; only if mgTableAddTakesSegments */
bool     mgTableAddTakesSegments (const MgTable *t, U64 n, const MgHistReq *counted);
MgStatus mgTableMarkOccupied (MgTable *t, const U64 *dKmer, U64 n, hipStream_t st);
MgStatus mgTableFind (MgTable *t, const U64 *dKmer, U64 n, U32 *dIndexOut, hipStream_t st);
size_t   mgTableFindPartScratchBytes (U64 n);
int      mgTableFindDigitBits (const MgTable *t);       /* bits of the partitioned lookup's digit: pieces of the table that fit an XCD's L2 */
bool     mgTableFindTakesPartition (const MgTable *t, U64 n, const MgHistReq *counted);
size_t   mgTableFindPart2ScratchBytes (U64 n);
MgStatus mgTableFindPartitioned (MgTable *t, const MgSegSrc &segSrc, U64 n, const MgHistReq *counted, U32 *dIndexOut, U64 *el, void *scratch, hipStream_t st,
                                 void *scratch2 = 0);   /* el: n words of scratch; scratch2 (mgTableFindPart2ScratchBytes): the two-level path */
MgStatus mgTableFindSegments (MgTable *t, const MgSegSrc &src, U64 n, U32 *dIndexOut, hipStream_t st);   /* the n k-mers of a lazy scan, in order */
MgStatus mgTableLoadHost (MgTable *t, const U64 *dValue, U32 first, U32 last, hipStream_t st);
MgStatus mgTableExportDepth (MgTable *t, U16 *dDelta, hipStream_t st);
MgStatus mgTableHistogram (MgTable *t, U64 *dHist, hipStream_t st);
MgStatus mgTableReplayIndex (MgTable *t, const MgHashParams &p, int tableBits, U32 *dIndex, hipStream_t st);
MgStatus mgTableMergeApply (const U32 *dIdx, const U16 *dDepth2, const U8 *dInfo2, U32 n2, U16 *dBaseDepth, U8 *dInfo1, hipStream_t st);
size_t   mgTablePruneScratchBytes (U32 n);
MgStatus mgTablePrune (MgTable *t, const U8 *dInfo, int lo, int hi, U64 *dNewValue, U16 *dNewDepth, U8 *dNewInfo,
                       void *scratch, hipStream_t st);
#define MG_COUNT_WORDS 4      /* dCount: {count, overflow flag, fullest block, capacity to retry with} */

MgStatus mgLaunchSynthGenome (U32 *dPacked, U64 nBases, U64 seed, hipStream_t st);
MgStatus mgLaunchSynthReads (const U32 *dGenome, U64 genomeBases, const U64 *dReadStart,
                             const U64 *dReadOffsets, const U8 *dStrand, U32 nReads, U64 totalBases,
                             double errRate, U64 seed, U32 *dOut, hipStream_t st);

#endif
