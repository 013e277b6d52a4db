/* include/modgpu_compat.h — layer 1 of libmodgpu.so's C ABI for callers WITHOUT the reference tree:
 * the reference's seqhash.h + modset.h declarations restated (identical signatures and struct layouts are
 * the ABI: reference callers touch the fields), including the parts the reference keeps header-inline
 * (seqhash.h:37,54-60; modset.h:49-69), so that code written against those headers compiles against this
 * one unchanged.  Never include it next to the reference's own modset.h / seqhash.h (they have no include
 * guards): there, include the reference headers and then modgpu.h.
 */
#ifndef MODGPU_COMPAT_H
#define MODGPU_COMPAT_H

#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <stdbool.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef UTILS_DEFINED            /* the reference's utils.h:33-39 has the same typedefs */
typedef uint8_t  U8;
typedef uint16_t U16;
typedef uint32_t U32;
typedef uint64_t U64;
#endif

/* ------------------------------------------------------------------------------------------
 * Layer 1: reference-compatible types (layouts are ABI: reference callers touch the fields).
 * ------------------------------------------------------------------------------------------ */

/* seqhash.h:15-23 — 80 bytes, written raw into .mod files (seqhash.c:41-44) */
typedef struct {
  int seed ;
  int k ;
  int w ;
  U64 mask ;
  int shift1, shift2 ;
  U64 factor1, factor2 ;
  U64 patternRC[4] ;
} Seqhash ;

/* seqhash.h:25-34.  hashBuf and fBuf are free()-able heap blocks and the iterator itself is a
 * free()-able block, because the reference's destroy is header-inlined into callers
 * (seqhash.h:54-55).  This library keeps the precomputed modimizers of the read behind hashBuf. */
typedef struct {
  Seqhash *sh ;
  char *s, *sEnd ;
  U64 h, hRC ;
  U64 *hashBuf ;
  bool *fBuf ;
  int base ;
  int iStart, iMin ;
  bool isDone ;
} SeqhashRCiterator ;

/* modset.h:17-28 — transparent: callers read/write index/value/depth/info/max directly */
typedef struct {
  Seqhash *hasher ;
  int tableBits ;
  U32 size ;
  U64 tableSize ;
  U64 tableMask ;
  U32 *index ;
  U64 *value ;
  U16 *depth ;
  U8  *info ;
  U32 max ;
} Modset ;

/* seqhash.h:36-60 */
Seqhash *seqhashCreate (int k, int w, int seed) ;                         /* seqhash.c:20-37 */
void seqhashWrite (Seqhash *sh, FILE *f) ;                                /* seqhash.c:41-44 */
Seqhash *seqhashRead (FILE *f) ;                                          /* seqhash.c:46-53 */
void seqhashReport (Seqhash *sh, FILE *f) ;                               /* seqhash.c:55-56 */
SeqhashRCiterator *modRCiterator (Seqhash *sh, char *s, int len) ;        /* seqhash.c:154-177 */
bool modRCnext (SeqhashRCiterator *si, U64 *kmer, int *pos, bool *isF) ;  /* seqhash.c:179-196 */
SeqhashRCiterator *minimizerRCiterator (Seqhash *sh, char *s, int len) ;  /* seqhash.c:83-108: one GPU pass, replayed */
bool minimizerRCnext (SeqhashRCiterator *si, U64 *u, int *pos, bool *isF) ; /* seqhash.c:110-152 */
char *seqString (U64 kmer, int len) ;                                     /* seqhash.c:198-206 */
/* header-inline in the reference (seqhash.h:37,54-60); exported here as real symbols too */
void mgSeqhashDestroy (Seqhash *sh) ;
void mgSeqhashRCiteratorDestroy (SeqhashRCiterator *si) ;
static inline U64 seqhash (Seqhash *sh, U64 k) { return ((k * sh->factor1) >> sh->shift1) ; }       /* seqhash.h:58 */
static inline void seqhashDestroy (Seqhash *sh) { free (sh) ; }                                       /* seqhash.h:37 */
static inline void seqhashRCiteratorDestroy (SeqhashRCiterator *si)                                    /* seqhash.h:54-55 */
{ free (si->hashBuf) ; free (si->fBuf) ; free (si) ; }
static inline char *seqhashString (Seqhash *sh, U64 k) { return seqString (k, sh->k) ; }              /* seqhash.h:60 */

/* modset.h:30-42 */
Modset *modsetCreate (Seqhash *sh, int bits, U32 size) ;                  /* modset.c:15-31 */
void modsetDestroy (Modset *ms) ;                                         /* modset.c:33-34 */
void modsetWrite (Modset *ms, FILE *f) ;                                  /* modset.c:79-88 */
Modset *modsetRead (FILE *f) ;                                            /* modset.c:90-104 */
U32 modsetIndexFind (Modset *ms, U64 kmer, int isAdd) ;                   /* modset.c:45-62 */
void modsetSummary (Modset *ms, FILE *f) ;                                /* modset.c:130-153 */
bool modsetPack (Modset *ms) ;                                            /* modset.c:36-43 */
void modsetDepthPrune (Modset *ms, int min, int max) ;                    /* modset.c:64-77 */
bool modsetMerge (Modset *ms1, Modset *ms2) ;                             /* modset.c:106-128 */

/* modset.h:44-69: info fields.  Bits 1 and 2: copy number in {0,1,2,M} with 0 for errors; bit 3 minor variants;
 * bit 4 repeats within a read; bit 5 internal within a read; bit 6 rDNA. */
#define MS_MINOR 4
#define MS_REPEAT 8
#define MS_INTERNAL 0x10
#define MS_RDNA 0x20
static inline void msSetCopy0 (Modset *ms, U32 i) { ms->info[i] &= 0xfc ; }
static inline void msSetCopy1 (Modset *ms, U32 i) { ms->info[i] = (ms->info[i] & 0xfc) | 1 ; }
static inline void msSetCopy2 (Modset *ms, U32 i) { ms->info[i] = (ms->info[i] & 0xfc) | 2 ; }
static inline void msSetCopyM (Modset *ms, U32 i) { ms->info[i] |= 3 ; }
static inline void msSetMinor (Modset *ms, U32 i) { ms->info[i] |= MS_MINOR ; }
static inline void msSetRepeat (Modset *ms, U32 i) { ms->info[i] |= MS_REPEAT ; }
static inline void msSetInternal (Modset *ms, U32 i) { ms->info[i] |= MS_INTERNAL ; }
static inline void msSetRDNA (Modset *ms, U32 i) { ms->info[i] |= MS_RDNA ; }
static inline bool msIsCopy0 (Modset *ms, U32 i) { return ((ms->info[i] & 3) == 0) ; }
static inline bool msIsCopy1 (Modset *ms, U32 i) { return ((ms->info[i] & 3) == 1) ; }
static inline bool msIsCopy2 (Modset *ms, U32 i) { return ((ms->info[i] & 3) == 2) ; }
static inline bool msIsCopyM (Modset *ms, U32 i) { return ((ms->info[i] & 3) == 3) ; }
static inline int  msCopy (Modset *ms, U32 i) { return (ms->info[i] & 3) ; }
static inline bool msIsMinor (Modset *ms, U32 i) { return (ms->info[i] & MS_MINOR) ; }
static inline bool msIsRepeat (Modset *ms, U32 i) { return (ms->info[i] & MS_REPEAT) ; }
static inline bool msIsInternal (Modset *ms, U32 i) { return (ms->info[i] & MS_INTERNAL) ; }
static inline bool msIsRDNA (Modset *ms, U32 i) { return (ms->info[i] & MS_RDNA) ; }

#ifdef __cplusplus
}
#endif
#endif /* MODGPU_COMPAT_H */
