/* mg_readset.c — modasm's read ingest on top of the batch ABI:
 *   readsetFileRead (modasm.c:151-191)   scan + lookup of every read: one GPU batch call
 *   invBuild        (modasm.c:258-287)   per-mod lists of the reads that hit it
 *   readsetStats    (modasm.c:193-253)
 *   readsetWrite / readsetRead (modasm.c:108-149): <root>.mod + <root>.readset
 * The per-k-mer loop (modRCnext + modsetIndexFind) runs on the GPU (mgQueryReadsDevice); what is
 * left on the host is the reference's serial bookkeeping over the seeds — hit lists with the
 * orientation bit, 16-bit distances, saturating depth, copy-class tallies, the inverse lists —
 * restated with its quirks because the file and the printed statistics are the parity target.
 */
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "modgpu.h"
#include "mg_internal.h"

static double rsNowMs (void) { struct timespec t; clock_gettime (CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
#define RS_LAP(what) do { if (lapOn) { const double q_ = rsNowMs (); fprintf (stderr, "readset: %s %.2f ms\n", what, q_ - lap); lap = q_; } } while (0)      /* dev: MODGPU_SEED_TIMING=1 */

#define TOPBIT  0x80000000u        /* modasm.c:22: set for forward orientation */
#define TOPMASK 0x7fffffffu

static void fatal (const char *what)
{ fprintf (stderr, "FATAL ERROR: %s: %s\n", what, mgLastError ()); exit (-1); }

MgReadset *mgReadsetCreate (Modset *ms)                            /* modasm.c:90-98 */
{
  if (ms->max >= TOPBIT) { fprintf (stderr, "FATAL ERROR: too many entries in modset\n"); exit (-1); }   /* modasm.c:1562 */
  MgReadset *rs = (MgReadset *) calloc (1, sizeof (MgReadset));
  rs->ms = ms;
  rs->capReads = 1 << 16;
  rs->len = (int *) calloc ((size_t) rs->capReads, sizeof (int));
  rs->nHit = (int *) calloc ((size_t) rs->capReads, sizeof (int));
  rs->nMiss = (int *) calloc ((size_t) rs->capReads, sizeof (int));
  rs->nCopy = (int (*)[4]) calloc ((size_t) rs->capReads, sizeof (int[4]));
  rs->hitStart = (U64 *) calloc ((size_t) rs->capReads + 1, sizeof (U64));
  return rs;
}

void mgReadsetDestroy (MgReadset *rs)
{
  if (!rs) return;
  free (rs->len); free (rs->nHit); free (rs->nMiss); free (rs->nCopy); free (rs->hitStart);
  mgReadsetDevForget (rs);
  free (rs->hit); free (rs->dx); free (rs->invStart); free (rs->invSpace); free (rs);
}

static void reserveReads (MgReadset *rs, int more)
{
  int need = rs->nReads + more + 2;
  if (need <= rs->capReads) return;
  int cap = rs->capReads; while (cap < need) cap *= 2;
  rs->len = (int *) realloc (rs->len, (size_t) cap * sizeof (int));
  rs->nHit = (int *) realloc (rs->nHit, (size_t) cap * sizeof (int));
  rs->nMiss = (int *) realloc (rs->nMiss, (size_t) cap * sizeof (int));
  rs->nCopy = (int (*)[4]) realloc (rs->nCopy, (size_t) cap * sizeof (int[4]));
  rs->hitStart = (U64 *) realloc (rs->hitStart, ((size_t) cap + 1) * sizeof (U64));
  rs->capReads = cap;
}

/* modasm.c:158: depth is rebuilt from the reads that follow.  The hits per mod are counted on the device across the batches of the
   file (mg_refpack.hip); depth[] comes back, saturated, when the file is done */
static U32 *gDepthAccum (MgReadset *rs)
{
  U32 *d = 0;
  if (mgReadsetDevBegin (rs, rs->ms->max, &d)) fatal ("read set on the device");
  return d;
}
static U32 *readsetBegin (MgReadset *rs)
{
  if (modsetSyncToHost (rs->ms, 0)) fatal ("modsetSyncToHost");
  return gDepthAccum (rs);
}

/* modasm.c:161-188 for a batch of reads that is on the device (2-bit packed, offsets in bases) */
static void readsetAddBatchDevice (MgReadset *rs, U32 *dDepth, const void *dPacked, U64 total, const void *dOff, int nReads)
{
  Modset *ms = rs->ms;
  if (nReads <= 0) return;
  const int lapOn = mgKnobs ()->seedTiming == 1; double lap = lapOn ? rsNowMs () : 0;
  reserveReads (rs, nReads);
  int64_t *offsets = (int64_t *) malloc (((size_t) nReads + 1) * sizeof (int64_t));
  if (mgMemcpyD2H (offsets, dOff, ((size_t) nReads + 1) * 8, 0)) fatal ("D2H");
  /* scan, lookup, hit lists, distances and counts on the device (mg_chain.hip) */
  U64 *hStart = (U64 *) malloc (((size_t) nReads + 1) * sizeof (U64));
  U32 *hMiss = (U32 *) malloc (((size_t) nReads + 1) * sizeof (U32));
  U32 *dHit = 0; U16 *dDx = 0;
  if (mgReadsetSeedsDevice (ms, (const U32 *) dPacked, total, (const U64 *) dOff, (U32) nReads, hStart, hMiss, &dHit, &dDx, dDepth)) fatal ("read scan");
  const U64 n = hStart[nReads];
  RS_LAP ("batch: scan + lookups + hit lists");
  if (rs->totHit + n + 1 > rs->capHit)
    { rs->capHit = (rs->totHit + n + 1) * 2;
      rs->hit = (U32 *) realloc (rs->hit, rs->capHit * sizeof (U32));
      rs->dx = (U16 *) realloc (rs->dx, rs->capHit * sizeof (U16));
      if (!rs->hit || !rs->dx) { fprintf (stderr, "FATAL ERROR: out of memory\n"); exit (-1); }
      mgHugeHint (rs->hit, rs->capHit * sizeof (U32)); mgHugeHint (rs->dx, rs->capHit * sizeof (U16));      /* the lists arrive into fresh pages: one fault per 2 MiB, not per 4 KiB */
    }
  if (n && (mgCopyD2HBig (rs->hit + rs->totHit, dHit, (size_t) n * sizeof (U32)) || mgCopyD2HBig (rs->dx + rs->totHit, dDx, (size_t) n * sizeof (U16)))) fatal ("hit lists");
  RS_LAP ("batch: hit lists to the host");
  mgReadsetDevAppendHits (rs, dHit, n);              /* (the inverse lists are made from them on the device when the file is done) */
  mgDeviceFree (dHit); mgDeviceFree (dDx);
  const int first = rs->nReads + 1;                   /* reads are numbered from 1 (modasm.c:95) */
  for (int r = 0 ; r < nReads ; ++r)
    { const int id = first + r;
      rs->len[id] = (int) (offsets[r + 1] - offsets[r]);
      rs->nHit[id] = (int) (hStart[r + 1] - hStart[r]);
      rs->nMiss[id] = (int) hMiss[r];
      memset (rs->nCopy[id], 0, sizeof (int[4]));
      rs->hitStart[id] = rs->totHit + hStart[r];
    }
  rs->totHit += n;
  rs->nReads += nReads;
  rs->hitStart[rs->nReads + 1] = rs->totHit;
  free (hStart); free (hMiss); free (offsets);
  RS_LAP ("batch: per-read records");
}

/* ... from host bytes */
static void readsetAddBatch (MgReadset *rs, U32 *dDepth, const char *bases, const int64_t *offsets, int nReads)
{
  if (nReads <= 0) return;
  const int lapOn = mgKnobs ()->seedTiming == 1; double lap = lapOn ? rsNowMs () : 0;
  MgDevBatch b; mgBatchUpload (&b, bases, offsets, nReads);
  RS_LAP ("pack + upload");
  readsetAddBatchDevice (rs, dDepth, b.dPacked, b.total, b.dOff, nReads);
  mgBatchFree (&b);
  RS_LAP ("batch in all, + free");
}

/* invBuild (modasm.c:258-287) and the file's depth[] (modasm.c:174): on the device (mg_refpack.hip: counts saturated, the lists a stable sort
   of the hits' read numbers by mod, a read's copy-class tallies a lane per read); a set of 2^32 hits or more takes the loops below */
static void readsetFinishHost (MgReadset *rs);
/* hitsBefore: rs->totHit when this file began.  A further file into a read set that holds hits already (modasm.c:158 zeroes depth[] per file
   and invBuild then walks ALL hits with the last file's depths: undefined in the reference): the device counted this file's hits alone, so
   depth[] and the lists are made from all the hits by the host loops, which agree with one another */
static void readsetFinish (MgReadset *rs, U64 hitsBefore)
{
  Modset *ms = rs->ms;
  free (rs->invStart); free (rs->invSpace); rs->invSpace = 0;
  rs->invStart = (U64 *) calloc ((size_t) ms->max + 2, sizeof (U64));
  mgHugeHint (rs->invStart, ((size_t) ms->max + 2) * sizeof (U64));
  if (rs->totHit >= 0xfffffff0ull || hitsBefore) { mgReadsetDevForget (rs); readsetFinishHost (rs); return; }
  if (mgReadsetFinishDevice (rs, ms, ms->max, rs->hit, rs->totHit, rs->hitStart, (U32) rs->nReads, ms->info, ms->depth, rs->invStart, &rs->invSpace, (int *) rs->nCopy))
    fatal ("read set on the device");                  /* (the device table's depth copy was set there too) */
}

/* the same by the reference's loops, for a set too large for 32-bit places (depth[] from the hits themselves) */
static void readsetFinishHost (MgReadset *rs)
{
  Modset *ms = rs->ms;
  memset (ms->depth, 0, ((size_t) ms->max + 1) * sizeof (U16));
  for (U64 h = 0 ; h < rs->totHit ; ++h)
    { U16 *dp = &ms->depth[rs->hit[h] & TOPMASK]; if (*dp < 0xffff) ++*dp; }      /* modasm.c:174 */
  mgModsetHostChanged (ms);
  rs->invSpace = (U32 *) malloc ((rs->totHit ? rs->totHit : 1) * sizeof (U32));
  U64 off = 0;
  for (U32 i = 1 ; i <= ms->max ; ++i)
    { rs->invStart[i] = off;
      if (ms->depth[i] && ms->depth[i] < 0xffff) off += ms->depth[i];
    }
  rs->invStart[ms->max + 1] = off;
  U64 *fill = (U64 *) malloc (((size_t) ms->max + 2) * sizeof (U64));
  memcpy (fill, rs->invStart, ((size_t) ms->max + 2) * sizeof (U64));
  for (int r = 1 ; r <= rs->nReads ; ++r)
    { int *nc = rs->nCopy[r];
      nc[0] = nc[1] = nc[2] = nc[3] = 0;               /* rebuilt here in case the copy classes changed */
      for (U64 h = rs->hitStart[r] ; h < rs->hitStart[r + 1] ; ++h)
        { const U32 y = rs->hit[h] & TOPMASK;
          ++nc[ms->info[y] & 3];
          if (ms->depth[y] < 0xffff) rs->invSpace[fill[y]++] = (U32) r;
        }
    }
  free (fill);
}

int mgReadsetRead (MgReadset *rs, const char *bases, const int64_t *offsets, int nReads)
{
  const int lapOn = mgKnobs ()->seedTiming == 1; double lap = lapOn ? rsNowMs () : 0;
  const U64 hitsBefore = rs->totHit;
  U32 *dDepth = readsetBegin (rs);
  RS_LAP ("begin");
  readsetAddBatch (rs, dDepth, bases, offsets, nReads);
  lap = lapOn ? rsNowMs () : 0;
  readsetFinish (rs, hitsBefore);
  RS_LAP ("finish");
  return 0;
}

typedef struct { MgReadset *rs; U32 *dDepth; } RsFileCtx;
static int rsDeviceBatch (void *v, const U32 *dPacked, U64 total, const U64 *dOff, U32 nReads, const char *idBytes, const U64 *idOff, void *stream)
{ RsFileCtx *c = (RsFileCtx *) v; (void) idBytes; (void) idOff; (void) stream; readsetAddBatchDevice (c->rs, c->dDepth, dPacked, total, dOff, (int) nReads); return 0; }
static int rsHostBatch (MgSeqBatch *b, void *v)
{ RsFileCtx *c = (RsFileCtx *) v; readsetAddBatch (c->rs, c->dDepth, b->bases, b->offsets, b->nSeq); return 0; }

int mgReadsetFileRead (MgReadset *rs, const char *filename)       /* modasm.c:151-191 */
{
  { FILE *f = fopen (filename, "rb"); if (!f) return -1; fclose (f); }
  const U64 hitsBefore = rs->totHit;
  RsFileCtx c; c.rs = rs; c.dDepth = readsetBegin (rs);
  /* plain FASTA / FASTQ text: parsed on the device (mg_textgpu.hip), the batches never exist as host bytes; gzip, a last line without
     its newline, FASTQ that breaks a rule: the host parser, from the first record the device parser has not handed on */
  const MgKnobs *kn = mgKnobs ();
  U64 nSeq = 0, totLen = 0, resumeOff = 0, resumeLine = 1;
  U64 batch = (U64) (kn->fileBatchMbp != MG_KNOB_UNSET && kn->fileBatchMbp > 0 ? kn->fileBatchMbp : 512) * 1000000;
  int rc = mgTextForEachBatchDevice (filename, rsDeviceBatch, &c, batch, 0, &nSeq, &totLen, &resumeOff, &resumeLine);
  if (rc == -2) rc = mgSeqForEachBatchFrom (filename, 0, 1, 0, rsHostBatch, &c);
  else if (rc == -3) rc = mgSeqForEachBatchFrom (filename, (size_t) resumeOff, resumeLine, nSeq, rsHostBatch, &c);
  /* a parser error after batches were appended: the reads that came are in rs (nReads, hit[] moved on), so depth[] / the inverse lists /
     nCopy[] are still made for them and the file's device buffers released -- rs is never left half built; the error goes back */
  readsetFinish (rs, hitsBefore);
  return rc;
}

void mgReadsetStats (MgReadset *rs, FILE *out)                     /* modasm.c:193-253 */
{
  Modset *ms = rs->ms;
  const U32 n = (U32) rs->nReads;
  if (!n) { fprintf (stderr, "stats called on empty readset\n"); return; }
  modsetSummary (ms, out);
  int nUnique0 = 0, nUnique1 = 0;
  U64 totLen = 0, totMiss = 0, lenUnique0 = 0, lenUnique1 = 0, totCopy[4] = { 0, 0, 0, 0 };
  for (U32 i = 1 ; i <= n ; ++i)
    { totLen += (U64) rs->len[i]; totMiss += (U64) rs->nMiss[i];
      for (int j = 0 ; j < 4 ; ++j) totCopy[j] += (U64) rs->nCopy[i][j];
      if (rs->nCopy[i][1] == 0) { ++nUnique0; lenUnique0 += (U64) rs->len[i]; }
      else if (rs->nCopy[i][1] == 1) { ++nUnique1; lenUnique1 += (U64) rs->len[i]; }
    }
  fprintf (out, "RS %d sequences, total length %llu (av %.1f)\n", n, (unsigned long long) totLen, totLen / (double) n);
  fprintf (out, "RS %llu mod hits, %.1f bp/hit, frac hit %.2f, av hits/read %.1f\n",
           (unsigned long long) rs->totHit, totLen / (double) rs->totHit,
           rs->totHit / (double) (totMiss + rs->totHit), rs->totHit / (double) n);
  fprintf (out, "RS hit distribution %.2f copy0, %.2f copy1, %.2f copy2, %.2f copyM\n",
           totCopy[0] / (double) rs->totHit, totCopy[1] / (double) rs->totHit,
           totCopy[2] / (double) rs->totHit, totCopy[3] / (double) rs->totHit);
  const U32 nUniqueMulti = n - (U32) nUnique0 - (U32) nUnique1;
  fprintf (out, "RS num reads and av_len with 0 copy1 hits %d %.1f with 1 copy1 hits %d %.1f"
           " >1 copy1 hits %d %.1f av copy1 hits %.1f\n",
           nUnique0, lenUnique0 / (double) nUnique0, nUnique1, lenUnique1 / (double) nUnique1,
           nUniqueMulti, (totLen - lenUnique0 - lenUnique1) / (double) nUniqueMulti,
           (totCopy[1] - (U64) nUnique1) / (double) nUniqueMulti);
  /* the bad-read flags are set by modasm's later passes (-b), which are not part of the ingest */
  fprintf (out, "RS bad %u : %u repeat, %u order10, %u order1, ", 0u, 0u, 0u, 0u);
  fprintf (out, "%u no_match, %u low_hit, %u low_copy1\n", 0u, 0u, 0u);
  U32 nCopy[4] = { 0, 0, 0, 0 }, hitCopy[4] = { 0, 0, 0, 0 }, hit2Copy[4] = { 0, 0, 0, 0 };
  U64 depthCopy[4] = { 0, 0, 0, 0 };
  for (U32 i = 1 ; i <= ms->max ; ++i)
    { const int j = ms->info[i] & 3;
      ++nCopy[j];
      if (ms->depth[i] > 0) ++hitCopy[j];
      if (ms->depth[i] > 1) { ++hit2Copy[j]; depthCopy[j] += ms->depth[i]; }
    }
  static const char *label[4] = { "copy0", "copy1", "copy2", "copyM" };
  fprintf (out, "RS mod frac hit hit>1 av:");
  for (int j = 0 ; j < 4 ; ++j)
    fprintf (out, " %s %.3f %.3f %.1f", label[j], hitCopy[j] / (double) nCopy[j], hit2Copy[j] / (double) nCopy[j],
             depthCopy[j] / (double) hit2Copy[j]);
  fprintf (out, "\n");
}

/* ---- <root>.mod + <root>.readset (modasm.c:108-149) ---- */

/* one element of the reference's Array of Read (modasm.c:30-57), 72 bytes; hit and dx are the
 * addresses of the read's lists in the writing process, meaningless in a file */
typedef struct {
  int len, nHit; U64 hitPtr, dxPtr; U8 bad, otherFlags; U16 pad1; int nMiss, contained, nCopy[4]; U32 pad2[4]; U32 tail;
} FileRead;
typedef struct { int magic, pad0; U64 base; int dim, size, max, pad1; } FileArray;   /* array.h:41-50 */
#define FILE_ARRAY_MAGIC 8918274

static void die1 (const char *fmt, const char *arg)
{ fprintf (stderr, "FATAL ERROR: "); fprintf (stderr, fmt, arg); fprintf (stderr, "\n"); exit (-1); }
#define WR(ptr, sz, cnt, what) do { if (fwrite ((ptr), (sz), (cnt), f) != (size_t) (cnt)) die1 ("failed to write %s", what); } while (0)
#define RD(ptr, sz, cnt, what) do { if (fread ((ptr), (sz), (cnt), f) != (size_t) (cnt)) die1 ("failed to read %s", what); } while (0)

void mgReadsetWrite (MgReadset *rs, const char *root)              /* modasm.c:108-126 */
{
  FILE *f;
  if (!(f = mgTagOpen (root, "mod", "w"))) die1 ("can't open file %s.mod", root);
  modsetWrite (rs->ms, f); fclose (f);
  if (!(f = mgTagOpen (root, "readset", "w"))) die1 ("can't open file %s.readset", root);
  WR ("RSMSHv2", 8, 1, "readset header");
  WR (&rs->totHit, sizeof (U64), 1, "totHit");
  FileArray ah; memset (&ah, 0, sizeof (ah));
  ah.magic = FILE_ARRAY_MAGIC; ah.size = (int) sizeof (FileRead); ah.max = rs->nReads + 1;
  ah.dim = mgRefArrayDim (1 << 16, (int) sizeof (FileRead), rs->nReads + 1);           /* readsetCreate (ms, 1<<16), modasm.c:1568 */
  WR (&ah, sizeof (ah), 1, "reads");
  FileRead *recs = (FileRead *) calloc ((size_t) ah.dim, sizeof (FileRead));
  for (int i = 1 ; i <= rs->nReads ; ++i)
    { recs[i].len = rs->len[i]; recs[i].nHit = rs->nHit[i]; recs[i].nMiss = rs->nMiss[i];
      memcpy (recs[i].nCopy, rs->nCopy[i], sizeof (int[4]));
    }
  WR (recs, sizeof (FileRead), ah.dim, "reads");
  free (recs);
  for (int i = 1 ; i <= rs->nReads ; ++i)
    { const U64 a = rs->hitStart[i], n = rs->hitStart[i + 1] - a;
      if (!n) continue;
      WR (rs->hit + a, sizeof (U32), n, "hits");
      WR (rs->dx + a, sizeof (U16), n, "dx");
    }
  if (fclose (f)) die1 ("failed to close %s.readset", root);
}

MgReadset *mgReadsetLoad (const char *root)                        /* modasm.c:128-149 */
{
  FILE *f;
  if (!(f = mgTagOpen (root, "mod", "r"))) die1 ("can't open file %s.mod", root);
  Modset *ms = modsetRead (f); fclose (f);
  if (!(f = mgTagOpen (root, "readset", "r"))) die1 ("can't open file %s.readset", root);
  char tag[8]; RD (tag, 8, 1, "readset header");
  if (memcmp (tag, "RSMSHv2", 8)) die1 ("bad readset header %s != RSMSHv2", tag);
  MgReadset *rs = mgReadsetCreate (ms);
  RD (&rs->totHit, sizeof (U64), 1, "totHit");
  FileArray ah; RD (&ah, sizeof (ah), 1, "reads");
  if (ah.magic != FILE_ARRAY_MAGIC || ah.size != (int) sizeof (FileRead) || ah.max < 1 || ah.dim < ah.max) die1 ("bad reads array in %s.readset", root);
  FileRead *recs = (FileRead *) malloc ((size_t) ah.dim * sizeof (FileRead));
  RD (recs, sizeof (FileRead), ah.dim, "reads");
  reserveReads (rs, ah.max);
  rs->nReads = ah.max - 1;
  rs->capHit = rs->totHit + 1;
  rs->hit = (U32 *) malloc (rs->capHit * sizeof (U32));
  rs->dx = (U16 *) malloc (rs->capHit * sizeof (U16));
  U64 at = 0;
  for (int i = 1 ; i <= rs->nReads ; ++i)
    { rs->len[i] = recs[i].len; rs->nHit[i] = recs[i].nHit; rs->nMiss[i] = recs[i].nMiss;
      memcpy (rs->nCopy[i], recs[i].nCopy, sizeof (int[4]));
      rs->hitStart[i] = at;
      const U64 n = (U64) recs[i].nHit;
      if (recs[i].nHit < 0 || at + n > rs->totHit) die1 ("bad hit counts in %s.readset", root);
      if (n) { RD (rs->hit + at, sizeof (U32), n, "hits"); RD (rs->dx + at, sizeof (U16), n, "dx"); }
      at += n;
    }
  rs->hitStart[rs->nReads + 1] = at;
  free (recs);
  fclose (f);
  /* invBuild as readsetRead does (modasm.c:147), from the depth the .mod file carries */
  free (rs->invStart); rs->invStart = 0;
  {
    Modset *m = rs->ms;
    rs->invStart = (U64 *) calloc ((size_t) m->max + 2, sizeof (U64));
    rs->invSpace = (U32 *) malloc ((rs->totHit ? rs->totHit : 1) * sizeof (U32));
    U64 off = 0;
    for (U32 i = 1 ; i <= m->max ; ++i) { rs->invStart[i] = off; if (m->depth[i] && m->depth[i] < 0xffff) off += m->depth[i]; }
    rs->invStart[m->max + 1] = off;
    U64 *fill = (U64 *) malloc (((size_t) m->max + 2) * sizeof (U64));
    memcpy (fill, rs->invStart, ((size_t) m->max + 2) * sizeof (U64));
    for (int r = 1 ; r <= rs->nReads ; ++r)
      { int *nc = rs->nCopy[r]; nc[0] = nc[1] = nc[2] = nc[3] = 0;
        for (U64 h = rs->hitStart[r] ; h < rs->hitStart[r + 1] ; ++h)
          { const U32 y = rs->hit[h] & TOPMASK;
            if (y > m->max) die1 ("hit beyond the modset in %s.readset", root);
            ++nc[m->info[y] & 3];
            if (m->depth[y] < 0xffff && fill[y] < rs->totHit) rs->invSpace[fill[y]++] = (U32) r;
          }
      }
    free (fill);
  }
  return rs;
}
