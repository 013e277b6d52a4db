/* include/modgpu.h — C ABI of libmodgpu.so: modimizer's seqhash + modset hot path on MI355X (gfx950).
 *
 * Two layers, both plain C (pointers and sizes only, no C++/torch types):
 *
 *  (1) the reference's own seqhash.h / modset.h API with identical signatures and struct layouts,
 *      so modmap/modutils-style callers link against this library unchanged
 *      (reference seqhash.h:36-60, modset.h:30-69; struct layouts seqhash.h:15-34, modset.h:17-28).
 *      A caller built inside the reference tree keeps including the reference's own modset.h (its
 *      unmodified modutils.c / modmap.c link against libmodgpu.so as they are: oracle/Makefile builds
 *      them that way and tests/test_dropin.py runs them); a caller without the reference tree gets
 *      the same declarations, header-inline helpers included, from modgpu_compat.h.
 *
 *  (2) this file: batch entry points that carry the GPU path: whole batches of reads are scanned, the
 *      modimizers compacted in (read,pos) order, and inserted into / looked up in a device-resident
 *      modset table.  These replace the per-read loops of the reference callers
 *      (modutils.c:19-31, modmap.c:106-118, modmap.c:197-206).
 *
 * Using (2) next to the reference's headers: include the reference's modset.h FIRST, then this file.
 * It notices (MS_MINOR is a macro of modset.h:49; or define MODGPU_WITH_REFERENCE_HEADERS) and takes
 * Seqhash / Modset / U8..U64 from there instead of declaring them again.
 *
 * Error behaviour follows the reference: invalid parameters and capacity overflow print
 * "FATAL ERROR: ..." and exit(-1) (utils.c:19-30) in layer (1); layer (2) functions return a
 * non-zero MgStatus and leave a message retrievable with mgLastError().  There is no CPU fallback:
 * every batch entry point fails with MG_ERR_NO_DEVICE when no HIP device is usable.
 */
#ifndef MODGPU_H
#define MODGPU_H

#include <stdio.h>
#include <stdint.h>
#include <stdbool.h>
#include <stddef.h>

#if defined (MODGPU_WITH_REFERENCE_HEADERS) || defined (MS_MINOR)
/* layer 1 comes from the reference's modset.h + seqhash.h + utils.h, already included */
#ifdef __cplusplus
extern "C" {
#endif
void mgSeqhashDestroy (Seqhash *sh) ;                       /* real symbols for the reference's header-inline destroys */
void mgSeqhashRCiteratorDestroy (SeqhashRCiterator *si) ;
#ifdef __cplusplus
}
#endif
#else
#include "modgpu_compat.h"
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------
 * Layer 2: batch / device entry points (the GPU hot path).
 * ------------------------------------------------------------------------------------------ */

typedef enum {
  MG_OK = 0,
  MG_ERR_NO_DEVICE = 1,     /* no usable HIP device / runtime: there is no CPU fallback */
  MG_ERR_HIP = 2,           /* a HIP call failed; see mgLastError() */
  MG_ERR_ARG = 3,           /* invalid argument */
  MG_ERR_CAPACITY = 4,      /* output or modset capacity exceeded */
  MG_ERR_NOMEM = 5
} MgStatus ;

const char *mgLastError (void) ;
int  mgDeviceCount (void) ;                 /* 0 when no device; never initialises a context */
MgStatus mgSetDevice (int device) ;         /* device used by subsequent calls of this thread */
const char *mgVersion (void) ;                 /* "modgpu <version> (gfx950) src=<hash>" */
const char *mgSourceHash (void) ;             /* 16 hex digits: hash of the sources this binary was compiled from (csrc/mg_version.c) */

/* The per-read facade's latency switch (seqhash.c:154-196 behind modRCiterator).  A synchronous per-read call cannot
 * hide a kernel launch (13-15 us launch + poll whatever the length), so modRCiterator scans reads shorter than
 * MODGPU_ITER_HOST_BELOW bases (default: the measured crossover, 12288, or 8192 where w < 16; 0 = every read through the kernel) with the library's own scalar loop and
 * longer ones with one kernel launch; both write the same replay block.  mgIterScanHost is that scalar loop by itself
 * (tests pin it to the golden vectors, the oracle and the compiled reference without a GPU): the malloc()ed block
 * {U64 n; U64 kmer[n]; U32 pos | isF << 31 [n]} of one read.  It is not a fallback: modRCiterator die()s without a HIP
 * device whatever the read's length, and no batch entry point ever takes it. */
U64 *mgIterScanHost (const Seqhash *sh, const char *s, int len) ;
void mgReloadKnobs (void) ;                 /* the library reads its MODGPU_* environment knobs once; this reads them again (tests that set one between calls; not while another thread is inside the library) */
int  mgIterHostBelow (int below) ;          /* sets the crossover in bases (below < 0: only asks; 1 << 30: back to the defaults by w); returns the one in force before */

/* Device memory helpers so a host language needs no other HIP binding. */
MgStatus mgDeviceAlloc (void **dptr, size_t bytes) ;
MgStatus mgDeviceFree (void *dptr) ;
MgStatus mgMemcpyH2D (void *dst, const void *src, size_t bytes, void *stream) ;
MgStatus mgMemcpyD2H (void *dst, const void *src, size_t bytes, void *stream) ;
MgStatus mgMemsetD (void *dst, int byte, size_t bytes, void *stream) ;
MgStatus mgStreamSynchronize (void *stream) ;
/* Whole arrays between the device and PAGEABLE host memory at the link's speed: a team of host threads (mgXferThreadCount) moves the
 * array in 4 MiB pieces through page-locked blocks, one copy stream each (what modsetSyncToHost and mgReferenceRead mirror their
 * results with; a plain copy into pageable memory goes through the runtime's staging at a few GB/s).  Both wait for the device to be
 * idle first and return when the bytes are in place. */
MgStatus mgCopyD2HBig (void *hostDst, const void *devSrc, size_t bytes) ;
MgStatus mgCopyH2DBig (void *devDst, const void *hostSrc, size_t bytes) ;

/* 2-bit packed read layout in HBM: base i of the concatenated batch lives in bits
 * [30-2*(i%16), 32-2*(i%16)) of 32-bit word i/16 (first base in the most significant bits, so a
 * k-mer read left to right is a contiguous big-endian bit field).  Buffers are
 * mgPackedWords(n) words long: ceil(n/16) plus MG_PACK_PAD zero words of read-ahead slack.
 * Input bytes are bases 0..3 (seqio.c:643-652 after the N->0 patch of modmap.c:97); only the low
 * two bits of each byte are used. */
#define MG_PACK_PAD 8
size_t   mgPackedWords (U64 nBases) ;
void     mgPackHost (const char *bases, U64 nBases, U32 *words) ;
MgStatus mgPackDevice (const U8 *dBases, U64 nBases, U32 *dWords, void *stream) ;
MgStatus mgUnpackDevice (const U32 *dWords, U64 nBases, U8 *dBases, void *stream) ;
/* host bytes -> packed words in HBM (dPacked: mgPackedWords(n) words): PCIe copy in pieces + K1 on the device */
MgStatus mgUploadPack (const char *bases, U64 nBases, U32 *dPacked, void *stream) ;

/* Scan (seqhash.c:154-196 over a whole batch).
 * dPacked: the batch, reads concatenated without padding; dReadOffsets[nReads+1]: start of each
 * read in bases (offsets[0] = 0, offsets[nReads] = totalBases).
 * Outputs, in (read, pos) order, one entry per modimizer:
 *   dKmer[i]  canonical k-mer (seqhash.c:183)
 *   dPosF[i]  pos within its read in bits 0..30 (seqhash.c:184), isForward in bit 31
 *   dReadId[i] read ordinal (may be NULL; so may dPosF)
 * dCount: device U64[4] -> {number of modimizers found, overflow flag, fullest workgroup segment,
 * capacity to retry with}.  Workgroups stage their modimizers in per-workgroup segments of dWork
 * sized from `capacity`; when the flag is set (total > capacity, or one segment too small) the
 * outputs are unspecified, count is still the true total, and a retry with capacity = dCount[3]
 * succeeds.
 * dWork: mgScanWorkBytes(totalBases, nReads, capacity) bytes of device scratch. */
#define MG_POS_MASK 0x7fffffffu
#define MG_FWD_BIT  0x80000000u
size_t   mgScanWorkBytes (U64 totalBases, U32 nReads, U64 capacity) ;
MgStatus seqhashScanBatchDevice (const Seqhash *sh, const U32 *dPacked, U64 totalBases,
                                 const U64 *dReadOffsets, U32 nReads,
                                 U64 *dKmer, U32 *dPosF, U32 *dReadId, U64 capacity,
                                 U64 *dCount, void *dWork, void *stream) ;

/* Host-buffer convenience form of the above: bases[] as the reference's iterator takes them
 * (one byte per base, values 0..3), readOffsets[nReads+1].  Outputs are malloc()ed arrays the
 * caller frees; *survStart[nReads+1] gives each read's slice of the outputs. Returns the number of
 * modimizers, or -1 on error. */
int64_t seqhashScanBatch (const Seqhash *sh, const char *bases, const int64_t *readOffsets, int nReads,
                          U64 **kmer, int **pos, bool **isF, int64_t **survStart) ;

/* minimizerRCiterator + minimizerRCnext (seqhash.c:83-152) run to exhaustion on every read of a batch:
 * the (hash, pos | isF<<31) pairs successive minimizerRCnext calls return, reads in order, read r's at
 * [dReadStart[r], dReadStart[r+1]) (dReadStart holds nReads+1 entries).  One wavefront per read walks the
 * reference's chain of windows (see mg_minimizer.hip).  *n receives the total (also when it exceeds
 * `capacity`: MG_ERR_CAPACITY, nothing written). */
MgStatus seqhashMinimizerBatchDevice (const Seqhash *sh, const U32 *dPacked, U64 totalBases,
                                      const U64 *dReadOffsets, U32 nReads,
                                      U64 *dHash, U32 *dPosF, U64 *dReadStart, U64 capacity, U64 *n, void *stream) ;
/* host buffers in, malloc()ed arrays out (as seqhashScanBatch); returns the number of minimizers, -1 on error */
int64_t seqhashMinimizerBatch (const Seqhash *sh, const char *bases, const int64_t *readOffsets, int nReads,
                               U64 **hash, int **pos, bool **isF, int64_t **start) ;

/* Device modset (modset.c:45-62 as a batch).  The device table is created on first use from the
 * host arrays of `ms` and lives until modsetDestroy / mgModsetDeviceRelease.
 *
 * modsetAddBatchDevice: for i in order, what `index = modsetIndexFind (ms, kmer[i], true) ;
 * ++depth[index]` (modutils.c:25-26) would do: new k-mers receive indices max+1, max+2, ... in
 * order of first occurrence, ms->max is updated before return, depth increments accumulate on the
 * device until modsetSyncToHost.  dIndexOut (optional) receives the index of every kmer[i].
 * withDepth = 0 gives the modmap.c:109 form (insert without touching depth).
 *
 * modsetFindBatchDevice: `modsetIndexFind (ms, kmer[i], false)` — 0 for absent k-mers.
 *
 * n must be < 2^31 per call. */
MgStatus modsetAddBatchDevice (Modset *ms, const U64 *dKmer, U64 n, U32 *dIndexOut, int withDepth, void *stream) ;
MgStatus modsetFindBatchDevice (Modset *ms, const U64 *dKmer, U64 n, U32 *dIndexOut, void *stream) ;

/* Bring the host arrays of ms up to date with the device: value[] for new entries, saturating
 * depth[] += device counts (modutils.c:26), and (when wantIndex) the open-addressed index[] table
 * rebuilt exactly as the reference's sequence of inserts would have left it (modset.c:51-57). */
MgStatus modsetSyncToHost (Modset *ms, int wantIndex) ;
int      mgXferThreadCount (void) ;         /* host threads that move whole arrays between the device and the Modset's own arrays (modsetSyncToHost, device rebuilds): 4 (measured best: tools/xfer_probe.py) or fewer if the process may use fewer CPUs; MODGPU_XFER_THREADS overrides, up to 16 */
/* Drop the device table (host arrays untouched; pending device depth counts are synced first). */
MgStatus mgModsetDeviceRelease (Modset *ms) ;
/* Tell the library the caller changed ms->value/max/depth on the host behind its back. */
void     mgModsetHostChanged (Modset *ms) ;
/* Forget every entry: the state modsetCreate (modset.c:15-31) returns, without reallocating.
 * Clears the device table (if any), ms->max, and the host index[]/depth[]/info[] of used entries. */
MgStatus mgModsetClear (Modset *ms, void *stream) ;

/* Slots of the device table behind ms (0 when there is none): what bench.py prices the bucket image with. */
U64 mgModsetDeviceSlots (Modset *ms) ;

/* modutils.c:53-63 on the device: dHist[65536] (U64) += histogram of depth[1..max], where depth is
 * the host depth at last sync plus pending device counts, saturated at 65535. */
MgStatus modsetDepthHistogramDevice (Modset *ms, U64 *dHist65536, void *stream) ;

/* One-call forms used by the drivers and the benchmark: scan a device-resident packed batch and
 * feed the modimizers straight into the modset.
 *   mgAddReadsDevice   = addSequence (modutils.c:19-31) over every read of the batch;
 *   mgQueryReadsDevice = the lookup loop of queryProcess (modmap.c:197-206): seeds (index,pos)
 *                        per read incl. misses.
 * nHash receives the number of modimizers.  Scratch is taken from an internal per-modset arena
 * that grows on demand. */
MgStatus mgAddReadsDevice (Modset *ms, const U32 *dPacked, U64 totalBases,
                           const U64 *dReadOffsets, U32 nReads, U64 *nHash, void *stream) ;
MgStatus mgQueryReadsDevice (Modset *ms, const U32 *dPacked, U64 totalBases,
                             const U64 *dReadOffsets, U32 nReads,
                             U32 *dSeedIndex, U32 *dSeedPosF, U32 *dSeedRead, U64 capacity,
                             U64 *nSeeds, void *stream) ;

/* mgQueryReadsDevice in two halves, for batches that follow one another (config 3: 90 Gbp of reads in batches of 10): Async starts the
 * batch's scan on a stream of the library's own, into the other of two scratch arenas, and returns at once; Wait runs the lookups on
 * `stream` and returns when the seeds are complete.  Called as   Async (0); for every i: { Async (i + 1); Wait (i); }   the scan of batch
 * i + 1 (bound by instruction issue) runs beside the lookups of batch i (bound by memory requests).  At most two batches in flight,
 * waited for in the order they were started, each with output arrays of its own; the batch on `stream` must be complete there when
 * Async is called (the scan waits for what that stream holds at that moment); until the last ticket has been waited for the modset takes
 * no other batch call (they fail with MG_ERR_ARG).  The results are those of mgQueryReadsDevice, bit for bit. */
MgStatus mgQueryReadsDeviceAsync (Modset *ms, const U32 *dPacked, U64 totalBases, const U64 *dReadOffsets, U32 nReads,
                                  U32 *dSeedIndex, U32 *dSeedPosF, U32 *dSeedRead, U64 capacity, void **ticket, void *stream) ;
MgStatus mgQueryReadsDeviceWait (void *ticket, U64 *nSeeds, void *stream) ;      /* frees the ticket, also on error */

/* mgInsertReadsDevice = the insert loop of referenceFastaRead (modmap.c:106-118, isAdd true): every
 * modimizer is inserted WITHOUT touching depth and its (index,pos,read) returned. */
MgStatus mgInsertReadsDevice (Modset *ms, const U32 *dPacked, U64 totalBases,
                              const U64 *dReadOffsets, U32 nReads,
                              U32 *dSeedIndex, U32 *dSeedPosF, U32 *dSeedRead, U64 capacity,
                              U64 *nSeeds, void *stream) ;

/* modsetMerge (modset.c:106-128) with the second set given as bare arrays, entries at [1..n2]: merging
 * per-GPU modsets in rank order reproduces the single-stream build exactly. */
bool mgModsetMergeArrays (Modset *ms1, U64 *value2, U16 *depth2, U8 *info2, U32 n2) ;
/* The same with the arrays in DEVICE memory, entry i at [i - 1], and ms1 on the device: nothing of the second set crosses the host
 * link (what mgModsetMergeRankOrder does with what its peers sent).  false = ms1 has no device table, nothing done. */
bool mgModsetMergeDeviceArrays (Modset *ms1, const U64 *dValue2, const U16 *dDepth2, const U8 *dInfo2, U32 n2) ;

/* Multi-GPU from C, straight on RCCL (BASELINE config 4).  Reads shard over the GPUs, every GPU builds its own modset, no
 * collective on the data path; these are the exchanges there are.  Communicators as RCCL has them: ONE PROCESS, N DEVICES, a host
 * thread per device (mgCommInitAll; each thread calls mgSetDevice (device of its comm) and then uses the library as on one GPU) -- or ONE
 * PROCESS PER DEVICE (rank 0 calls mgCommGetUniqueId and hands the 128 bytes to the others by its own means -- a file, a socket, MPI --,
 * every rank calls mgCommInitRank).  librccl is loaded by the first of these calls, not before. */
typedef struct MgComm MgComm ;
MgStatus mgCommInitAll (MgComm **comms /* [nDev] out */, int nDev, const int *devices /* 0: 0 .. nDev-1 */) ;
MgStatus mgCommGetUniqueId (void *id128) ;
MgStatus mgCommInitRank (MgComm **comm, int nRanks, int rank, const void *id128, int device) ;
int      mgCommRank (const MgComm *c) ;
int      mgCommSize (const MgComm *c) ;
void     mgCommDestroy (MgComm *c) ;
/* config 4's collective: hist65536[d] (host, U64) = over all ranks, the number of modset entries of depth d (modutils.c:53-63 per rank,
 * all-reduced with SUM: 512 KiB a rank over xGMI).  Every rank calls it with its own set and gets the sum. */
MgStatus mgHistogramAllReduce (Modset *ms, U64 *hist65536, MgComm *c) ;
/* reads counted against a FIXED set that every rank holds (the same entries everywhere; modasm.c:158-174: depth zeroed, ++depth per hit,
 * saturating): depth[i] = min (65535, sum over the ranks of their depth[i]) on every rank, in ms->depth and in the device table.  Exact: a
 * saturating add is associative.  4 bytes per entry per rank through one ncclAllReduce (sum, uint32). */
MgStatus mgDepthAllReduce (Modset *ms, MgComm *c) ;
/* the exact global set: on rank `root` ms becomes the merge of every rank's set in RANK order with modsetMerge semantics
 * (modset.c:106-128): with contiguous blocks of reads per rank and root = 0 that is, bit for bit, the set one stream over all the reads
 * builds.  Every rank calls it; the others' sets are sent (point to point, one rank at a time) and left as they are. */
MgStatus mgModsetMergeRankOrder (Modset *ms, MgComm *c, int root) ;

/* Host-side batch mirrors of the reference callers' loops. */
/* modutils.c:19-31 over nReads reads; returns total hashes, -1 on error. ms->max updated. */
int64_t mgAddSequenceBatch (Modset *ms, const char *bases, const int64_t *readOffsets, int nReads) ;
/* modutils.c:53-63 */
void    mgDepthHistogram (Modset *ms, FILE *f) ;
/* modutils.c:33-51 on reads already in memory: adds every read, prints the "added ..." line */
int     mgAddSequences (Modset *ms, const char *bases, const int64_t *readOffsets, int nReads, FILE *out) ;
/* modutils.c:194-198 ("-wt") */
void    mgModsetWriteText (Modset *ms, FILE *f) ;

/* modmap.c's Reference (modmap.c:35-47) and its three operations, on sequences already in memory
 * (bases 0..3 one byte each, offsets[n+1], names[n]).  The per-k-mer loops run on the GPU
 * (mgInsertReadsDevice / mgQueryReadsDevice), and so do the per-occurrence bookkeeping, the copy classes and
 * the CSR pack (modmap.c:106-134,74-91: ordered append, exclusive scan, stable sort by index) and the queries'
 * tallies and seed chaining (modmap.c:213-276); the arrays below are the caller's, filled from the device when
 * mgReferenceRead / mgReferenceFastaRead return; the host formats the Q / M lines. */
typedef struct {
  Modset *ms ;
  U32 size ;                   /* capacity of index/offset/id */
  U32 max ;                    /* occurrences stored */
  U32 *index, *offset, *id ;   /* per occurrence: modset index, position in its sequence, sequence id */
  U32 *depth ;                 /* occurrences per modset index */
  U32 *rev, *loc ;             /* CSR inverse: occurrences grouped by modset index */
  int nSeq ;
  char **names ;               /* sequence names (the reference keeps them in a DICT) */
  U32 *len ;
} MgReference ;
MgReference *mgReferenceCreate (Modset *ms, U32 size) ;                               /* modmap.c:49-64 */
void mgReferenceDestroy (MgReference *ref) ;                                          /* modmap.c:66-72 */
/* modmap.c:93-134: scan + insert/lookup every sequence, classify copy number, pack; prints the two
 * report lines to out.  Returns 0 on success. */
int  mgReferenceRead (MgReference *ref, const char *bases, const int64_t *offsets, int nSeq,
                      const char **names, bool isAdd, FILE *out) ;
/* modmap.c:136-182: <root>.mod + <root>.ref in the reference's on-disk format (gzip streams, as its
 * fzopen writes them; gzip or plain accepted on read), interchangeable with modmap -w / -r.
 * mgReferenceLoad also creates the Modset (ref->ms, with its own Seqhash) as referenceRead does; as in the
 * reference, mgReferenceDestroy leaves ref->ms alone (modmap.c:66-72): the caller destroys it, and its hasher. */
void mgReferenceWrite (MgReference *ref, const char *root) ;
/* What the library's writers (mgReferenceWrite, mgReadsetWrite) put between themselves and the disk, for a caller's own modsetWrite
 * (modset.c:79-88 takes any FILE *): the bytes written to the returned FILE * become a gzip file of independent members of 16 MiB,
 * deflated by a team of threads (MODGPU_GZIP_THREADS; default: the CPUs the process may use) and written in order.  gzread -- the
 * reference's fzopen "r" (utils.c:107-127) -- and gunzip read such a file as the one stream the reference's single gzwrite would have
 * made; fclose () finishes it.  0 if `name` cannot be created. */
FILE *mgGzipOpenWrite (const char *name) ;
/* ... and its counterpart for modsetRead (modset.c:90-104): every member the writer makes carries its compressed and uncompressed size
 * in a gzip extra field (RFC 1952 2.3.1.1, subfield 'M' 'G'; gzread and gunzip skip it), so the members are found without inflating
 * anything and inflated by the team -- whole members straight into the array a large fread hands over.  0 if `name` is not such a file
 * from its first byte to its last (any other gzip file, a plain file): the caller then takes the reference's fzopen / gzopen, as
 * mgReferenceLoad and mgReadsetLoad do by themselves. */
FILE *mgGzipOpenRead (const char *name) ;
/* the reference's fzopen (utils.c:107-127) on top of the two: "w" = mgGzipOpenWrite; "r" = mgGzipOpenRead, and for any other gzip file
 * or a plain file zlib's gzopen behind a FILE * (what modsetRead / modsetWrite, seqhashRead / seqhashWrite take). */
FILE *mgFzOpen (const char *name, const char *mode) ;
MgReference *mgReferenceLoad (const char *root) ;
/* modmap.c:188-281: "Q" line and "M" lines for every read. */
int  mgQueryProcess (MgReference *ref, const char *bases, const int64_t *offsets, int nReads,
                     const char **names, FILE *out) ;
/* modmap's -v toggle (modmap.c:23,348): with it on, mgQueryProcess / mgQueryFile also print the per-seed lines of
 * modmap.c:218-229 ("  <pos>\t<seq> <offset>[\t<seq2> <offset2>]", to stdout as the reference's printf does,
 * before the M line a seed closes); the seed lists then come back to the host and are chained there. */
void mgSetVerbose (int on) ;

/* modasm's long-read set (modasm.c:30-57,79-86) as readsetFileRead + invBuild leave it:
 * per read its length, hit / miss counts and copy-class tallies; the hits (modset index, bit 31 =
 * forward) with the 16-bit distance to the previous hit; per mod the reads that hit it.  Reads are
 * numbered from 1 as in the reference; the scan + lookup of every read is one GPU batch call. */
typedef struct {
  Modset *ms ;
  int nReads, capReads ;
  int *len, *nHit, *nMiss ;     /* [1..nReads] */
  int (*nCopy)[4] ;             /* [1..nReads]: hits on copy 0 / 1 / 2 / M mods */
  U64 *hitStart ;               /* hits of read i: hit/dx[hitStart[i] .. hitStart[i+1]) */
  U32 *hit ; U16 *dx ;
  U64 totHit, capHit ;
  U64 *invStart ;               /* [ms->max+2]: mod i is hit by the reads invSpace[invStart[i] .. invStart[i+1]) */
  U32 *invSpace ;               /* (one entry per hit, read order; none for mods whose depth saturated, modasm.c:266,278) */
} MgReadset ;
MgReadset *mgReadsetCreate (Modset *ms) ;                                                   /* modasm.c:90-98 */
void mgReadsetDestroy (MgReadset *rs) ;
/* modasm.c:151-191 + 258-287: depth[] is rebuilt from these reads (modasm.c:158) */
int  mgReadsetRead (MgReadset *rs, const char *bases, const int64_t *offsets, int nReads) ;
int  mgReadsetFileRead (MgReadset *rs, const char *filename) ;
void mgReadsetStats (MgReadset *rs, FILE *out) ;                                            /* modasm.c:193-253 */
void mgReadsetWrite (MgReadset *rs, const char *root) ;                                     /* modasm.c:108-126 */
MgReadset *mgReadsetLoad (const char *root) ;                                               /* modasm.c:128-149; creates rs->ms, which mgReadsetDestroy leaves to the caller (modasm.c:100-107) */

/* The file front end (seqio.c:30-346 for FASTA / FASTQ text, plain, gzip or blocked gzip, with the callers'
 * dna2indexConv + N->0 conversion): records are cut out of the text and converted by a pool of
 * threads, a batch at a time.  bases hold 0..3 (FASTQ keeps other bytes as (char)-2, as the
 * reference does); offsets[nSeq+1]; names = the record ids. */
typedef struct MgSeqReader MgSeqReader ;
typedef struct { char *bases ; int64_t *offsets ; char **names ; int nSeq ; int64_t total ; int isFastq ; int64_t basesCap ; } MgSeqBatch ;   /* release with mgSeqBatchFree only */
MgSeqReader *mgSeqOpen (const char *filename) ;                        /* 0 if unreadable, empty or not FASTA/FASTQ text */
int  mgSeqNextBatch (MgSeqReader *r, int64_t maxBases, MgSeqBatch *out) ; /* whole records, at least one; 0 at the end */
void mgSeqBatchFree (MgSeqBatch *b) ;
void mgSeqClose (MgSeqReader *r) ;
void mgSeqReleaseBuffers (void) ;	/* the readers keep their two largest buffers for the next file (unmapping and touching gigabytes again costs as much as parsing them); this gives them back -- a no-op while a reader is open; also run when the library is unloaded */
void mgReleaseBuffers (void) ;	/* everything the library caches between calls: the readers' buffers (mgSeqReleaseBuffers), the device buffers and pinned staging of the host-buffer entry points (mgAddSequenceBatch, mgUploadPack: they live on the device the last call ran on and are re-made by themselves when the caller moves to another one), the calling thread's iterator scratch (modRCiterator), the page-locked blocks and device arrays mgQueryFile leaves */
/* Plain FASTA / FASTQ text parsed ON THE DEVICE (the host only moves the bytes: parallel pread into pinned memory, the text as it
 * is across PCIe, record starts / headers / bases found by small kernels per window): mgAddSequenceFile, mgReferenceFastaRead and
 * mgQueryFile take this path by themselves for plain text and use the reader above for gzip, a file that does not end in a
 * newline, a FASTA file whose last line is a header, FASTQ that breaks a rule (from the first record not handed on).
 * This entry returns the parser's records as host arrays (bases 0..3 one per byte, offsets[nSeq + 1]; free () both): 0 = done,
 * -1 = error (mgLastError), -2 = not a file the device parser takes as a whole (nothing is returned). */
int  mgTextParseFileDevice (const char *filename, char **bases, int64_t **offsets, int64_t *nSeq) ;
/* the callers' per-file loops: parsing of the next batch overlaps the GPU work on the current one */
int  mgAddSequenceFile (Modset *ms, const char *filename, FILE *out) ;                        /* modutils.c:33-51 */
int  mgReferenceFastaRead (MgReference *ref, const char *filename, bool isAdd, FILE *out) ;   /* modmap.c:93-134 */
int  mgQueryFile (MgReference *ref, const char *filename, FILE *out) ;                        /* modmap.c:188-281 */
int  mgFormatF2 (char *buf64, double x) ;	/* test hook: the "%.2f" of the Q / M lines as the library's parallel formatter writes it (glibc's rounding of the double's exact value, in integer arithmetic; snprintf itself for nan / inf / negative); returns the length */

/* Deterministic synthetic reads generated directly in HBM (not from the reference):
 *   mgSynthGenome: nBases iid-uniform bases, base g = splitmix64(seed ^ g*0x9E3779B97F4A7C15) >> 62
 *   mgSynthReads : read r = genome[start[r], start[r]+len[r]) reverse-complemented when
 *                  strand[r] != 0, each base substituted with probability errRate (a counter-based
 *                  hash of (seed, global base ordinal) decides), written 2-bit packed. */
MgStatus mgSynthGenome (U32 *dPacked, U64 nBases, U64 seed, void *stream) ;
MgStatus mgSynthReads (const U32 *dGenomePacked, U64 genomeBases,
                       const U64 *dReadStart, const U64 *dReadOffsets, const U8 *dStrand, U32 nReads,
                       U64 totalBases, double errRate, U64 seed, U32 *dPackedOut, void *stream) ;

/* Per-kernel timing with HIP events on the launch stream (for bench.py's roofline object).
 * While enabled, every kernel launch of the library is bracketed by an event pair; mgProfileGet
 * returns, per kernel id in [0, mgProfileKernels()), its name, summed duration and launch count
 * since the last mgProfileReset (it synchronises the device to read the events). */
void     mgProfileEnable (int on) ;
void     mgProfileOnly (int kernelId) ;     /* >= 0: only that kernel is bracketed (an event pair costs a few microseconds of stream time per launch); -1: all */
void     mgProfileReset (void) ;
int      mgProfileKernels (void) ;
MgStatus mgProfileGet (int id, const char **name, double *totalMs, U64 *launches) ;

#ifdef __cplusplus
}
#endif
#endif /* MODGPU_H */
