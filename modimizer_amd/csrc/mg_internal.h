/* mg_internal.h — private glue between mg_host.c (reference-compatible scalar API, plain C)
 * and mg_api.hip (device state).  Not part of the public ABI. */
#ifndef MG_INTERNAL_H
#define MG_INTERNAL_H
#include "modgpu.h"
#include "mg_knobs.h"
#ifdef __cplusplus
extern "C" {
#endif
extern volatile int mgLiveDeviceModsets;          /* >0 when any Modset has a device table */
void mgHookDestroy (Modset *ms);                   /* modset is going away */
void mgHookHostRewrote (Modset *ms);               /* host arrays were rebuilt: drop the device table */
void mgHookNeedHost (Modset *ms, int wantIndex);   /* make value[] (and index[]) current; cheap when they are */
void mgHookNeedHostAll (Modset *ms, int wantIndex);/* also fold pending device depth counts into depth[] */
int  mgHookHasDevice (Modset *ms);
int  mgHookMergeDevice (Modset *ms1, Modset *ms2);   /* modsetMerge with ms1 on the device; 0 = done */
int  mgHookDeviceView (Modset *ms, const U64 **dValue1, const U16 **dDepth1, U32 *max);      /* entries 1 .. max on the device, counts folded; -1: no device table */
int  mgHookMergeDeviceArrays (Modset *ms1, const U64 *dValue2, const U16 *dDepth2, const U8 *dInfo2, U32 n2);   /* the second set as device arrays */
int  mgHookPruneDevice (Modset *ms, int lo, int hi);  /* modsetDepthPrune on the device; 0 = done */
/* one GPU scan of one read for the iterator facade: *blk = malloc()ed replay block {U64 n; U64 kmer[n]; U32 posF[n]} */
int  mgIterScan (Seqhash *sh, const char *s, int len, U64 **blk);
int  mgIterRequireDevice (void);                   /* 0 when a HIP device is usable (cached); else the error is set */
void mgIterReleaseBuffers (void);                  /* the calling thread's iterator scratch (pinned buffers, stream) */
/* the same for the minimizer iterator: *rec = malloc()ed {U64 hash[n]; U32 posF[n]} */
int  mgIterMinScan (Seqhash *sh, const char *s, int len, U64 **rec, U64 *nOut);
/* the file front end on the device (mg_textgpu.hip): plain FASTA / FASTQ text parsed by the GPU.  0 = done, -1 = error, -2 = not a
   file for this path (the host parser takes it), -3 = FASTQ text this parser leaves to the host parser from byte *resumeOff (a
   record start, line *resumeLine of the file; the counts are those of the records added so far) */
int  mgAddSequenceFileDevice (Modset *ms, const char *filename, U64 *nSeq, U64 *totLen, U64 *totHash, U64 *resumeOff, U64 *resumeLine);
int  mgTextParseFileDevice (const char *filename, char **basesOut, int64_t **offsetsOut, int64_t *nSeqOut);   /* test hook: the parser's records as host arrays (malloc) */
/* the same parser for the callers that print record ids: every batch of complete records (device resident: packed bases, read
   offsets) with its ids (id r = idBytes + idOff[r], 0-terminated: seqio.c:303-304) to fn; a non-zero return of fn ends the file */
typedef int (*MgTextBatchFn) (void *ctx, const U32 *dPacked, U64 totalBases, const U64 *dReadOffsets, U32 nReads, const char *idBytes, const U64 *idOff, void *stream);
int  mgTextForEachBatchDevice (const char *filename, MgTextBatchFn fn, void *ctx, U64 batchBases, U64 batchRecs, U64 *nSeq, U64 *totLen, U64 *resumeOff, U64 *resumeLine);   /* a batch is handed on (at a window's end) once it holds batchBases bases (0: the default, 1 Gbp) and batchRecs records, or the default's bases whatever the records */
/* the device halves of the modmap callers (mg_callers.c): a batch that is already on the device */
int  mgQueryProcessDevice (MgReference *ref, const U32 *dPacked, U64 totalBases, const U64 *dReadOffsets, int nReads, const char **names, FILE *out);
/* mgQueryFile's batches: Push runs the device half now, one writer thread formats and writes the lines, in order, behind it */
typedef struct MgQueryPipe MgQueryPipe;
MgQueryPipe *mgQueryPipeOpen (MgReference *ref, FILE *out);
int  mgQueryPipePush (MgQueryPipe *p, const U32 *dPacked, U64 totalBases, const U64 *dReadOffsets, int nReads, const char *idBytes, const U64 *idOff);
void mgQueryPipeClose (MgQueryPipe *p);                    /* returns when every line is written */
int  mgReferenceAddDevice (MgReference *ref, const U32 *dPacked, U64 totalBases, const U64 *dReadOffsets, int nSeq, const char **names, bool isAdd);
void mgReferenceFinish (MgReference *ref, U64 totLen, bool isAdd, FILE *out);
void mgTextReleaseBuffers (void);
/* shared by the caller mirrors (mg_callers.c, mg_readset.c): not exported */
#define MG_HIDDEN __attribute__ ((visibility ("hidden")))
typedef struct { void *dPacked, *dOff; U64 total; U32 nReads; } MgDevBatch;     /* host bytes -> 2-bit words in HBM */
MG_HIDDEN int mgSeqForEachBatchFrom (const char *filename, size_t startOff, U64 startLine, U64 startSeq, int (*fn) (MgSeqBatch *, void *), void *ctx);   /* the host parser's batches (the next one parsed while fn works), from a record start on */
MG_HIDDEN void mgBatchUpload (MgDevBatch *b, const char *bases, const int64_t *offsets, int nReads);
MG_HIDDEN void mgBatchFree (MgDevBatch *b);
MG_HIDDEN FILE *mgTagOpen (const char *root, const char *tag, const char *mode);
/* queryProcess on the device (mg_chain.hip): per read the tallies of its "Q" line and its "M" blocks */
typedef struct { U32 nSeeds, missed, copy1, copy2, copyM, nM; } MgChainQ;
typedef struct { U32 pos0, posN, id0, off0, offN; int n1, n2; U32 span; } MgChainM;
MG_HIDDEN int  mgChainQueryDevice (const MgReference *ref, const U32 *dPacked, U64 totalBases, const U64 *dReadOffsets, U32 nReads,
                                   MgChainQ *hQ, MgChainM **hMOut, U32 maxM, int hQPinned);
MG_HIDDEN void mgChainForget (const MgReference *ref);
/* the Reference built on the device (mg_refpack.hip): a batch's seeds appended (modmap.c:110-117), then copy classes + referencePack (modmap.c:125-129,74-91) into the caller's arrays */
MG_HIDDEN MgStatus mgRefBuildAppend (MgReference *ref, const U32 *dIx, const U32 *dPosF, const U32 *dRid, U64 n, U32 idBase, U32 *appended);
MG_HIDDEN int mgRefPackedTallies (MgReference *ref, U32 tallies[3]);      /* 1: packed on the device, nothing added since */
MG_HIDDEN MgStatus mgRefBuildFinish (MgReference *ref, U32 *hIndex, U32 *hOffset, U32 *hId, U32 *hDepth, U32 *hRev, U32 *hLoc, U8 *hInfo, U32 tallies[3]);
MG_HIDDEN void *mgPinnedAlloc (size_t bytes);          /* page-locked host memory (0: none to be had) */
MG_HIDDEN void mgPinnedFree (void *p);
MG_HIDDEN MgStatus mgCopyOutPinned (void *dstPinned, const void *srcDev, size_t bytes, void *stream);   /* device -> page-locked block by a kernel (not the copy engine), waited for */
MG_HIDDEN void mgChainReleaseBuffers (void);
MG_HIDDEN void mgQueryReleaseBuffers (void);           /* the query path's cached buffers: page-locked blocks (mg_callers.c), device arrays (mg_chain.hip) */
MG_HIDDEN void mgChainScratchKeep (int on);      /* 1: the query's device arrays stay allocated between batches; 0: ends that (and frees them) */
/* readsetFileRead's per-read loop on the device (mg_chain.hip): hit lists, distances, counts, hits per mod */
MG_HIDDEN int  mgReadsetSeedsDevice (Modset *ms, const U32 *dPacked, U64 totalBases, const U64 *dReadOffsets, U32 nReads,
                                     U64 *hHitStart, U32 *hNMiss, U32 **dHitOut, unsigned short **dDxOut, U32 *dDepthAccum);
/* the per-mod side of the read set on the device (mg_refpack.hip): hit counts kept across a file's batches; depth[], invStart[], invSpace[], nCopy[] at its end */
MG_HIDDEN MgStatus mgReadsetDevBegin (const void *rs, U32 msMax, U32 **dDepth);
MG_HIDDEN MgStatus mgReadsetFinishDevice (const void *rs, Modset *ms, U32 msMax, const U32 *hHit, U64 totHit, const U64 *hHitStart, U32 nReads, const U8 *hInfo,
                                          U16 *hDepth16, U64 *hInvStart, U32 **hInvSpace, int *hNCopy);
MG_HIDDEN void mgReadsetDevForget (const void *rs);
MG_HIDDEN void mgReadsetDevAppendHits (const void *rs, const U32 *dHit, U64 n);      /* a batch's hit list kept on the device for the end of the file */
MG_HIDDEN MgStatus mgModsetAdoptDepthDevice (Modset *ms, const U16 *dDepth16);      /* the device table's depth copy = dDepth16[0 .. max] (mg_api.hip) */
/* element count of the reference's Array after appending elements 0..n-1 (array.c:144-170,180-183) */
MG_HIDDEN int mgRefArrayDim (int first, int size, int n);
#ifdef __cplusplus
}
#endif
#endif
