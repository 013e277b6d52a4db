/* mg_callers.c — C counterparts of the reference's hot-path callers, on top of the batch ABI:
 *   modutils: addSequenceFile's loop + report (modutils.c:33-51), "-wt" dump (modutils.c:194-198)
 *   modmap  : Reference build (modmap.c:49-134) and queryProcess (modmap.c:188-281)
 * Inputs are sequences already in memory (the FASTA front end, seqio.c, is out of scope).  Every
 * per-k-mer loop of the reference is one GPU batch call here; what remains on the host is the
 * reference's own serial bookkeeping, restated with its quirks because its printed output is the
 * parity target.
 */
#define _GNU_SOURCE             /* fopencookie */
#include <stdlib.h>
#include <string.h>
#include <sched.h>
#include <unistd.h>
#include <zlib.h>
#include <time.h>
#include "modgpu.h"
#include "mg_internal.h"

static void fatal (const char *what)
{ fprintf (stderr, "FATAL ERROR: %s: %s\n", what, mgLastError ()); exit (-1); }

/* ---- packed batch on the device (host bytes -> 2-bit words -> HBM) ---- */

void mgBatchUpload (MgDevBatch *b, const char *bases, const int64_t *offsets, int nReads)
{
  b->nReads = (U32) nReads;
  b->total = nReads ? (U64) offsets[nReads] : 0;
  size_t nw = mgPackedWords (b->total);
  if (mgDeviceAlloc (&b->dPacked, nw * 4) || mgDeviceAlloc (&b->dOff, ((size_t) nReads + 1) * 8)) fatal ("device alloc");
  if (mgUploadPack (bases, b->total, (U32 *) b->dPacked, 0) || mgMemcpyH2D (b->dOff, offsets, ((size_t) nReads + 1) * 8, 0)
      || mgStreamSynchronize (0)) fatal ("H2D");
}

void mgBatchFree (MgDevBatch *b) { mgDeviceFree (b->dPacked); mgDeviceFree (b->dOff); }

/* ---------------------------------- modutils ---------------------------------- */

int mgAddSequences (Modset *ms, const char *bases, const int64_t *readOffsets, int nReads, FILE *out)
{
  int64_t nHash = mgAddSequenceBatch (ms, bases, readOffsets, nReads);
  if (nHash < 0) return -1;
  U64 totLen = nReads ? (U64) readOffsets[nReads] : 0;
  fprintf (out, "added %llu sequences total length %llu total hashes %llu, new max %u\n",
           (unsigned long long) nReads, (unsigned long long) totLen, (unsigned long long) nHash, ms->max);
  return 0;
}

void mgModsetWriteText (Modset *ms, FILE *f)
{
  if (modsetSyncToHost (ms, 0)) fatal ("modsetSyncToHost");
  Seqhash *sh = ms->hasher;
  fprintf (f, "modset bits %d size %d k %d w %d seed %d\n", ms->tableBits, ms->max + 1, sh->k, sh->w, sh->seed);
  for (U32 i = 1 ; i <= ms->max ; ++i)
    fprintf (f, "%d\t%s\t%d\t%d\n", (int) i, seqString (ms->value[i], sh->k), ms->depth[i], ms->info[i]);
}

/* ---------------------------------- modmap ---------------------------------- */

MgReference *mgReferenceCreate (Modset *ms, U32 size)
{
  if (!ms || !ms->size) { fprintf (stderr, "FATAL ERROR: modset must be initialised before reference\n"); exit (-1); }
  if (!size) { fprintf (stderr, "FATAL ERROR: refCreate must have size > 0\n"); exit (-1); }
  MgReference *ref = (MgReference *) calloc (1, sizeof (MgReference));
  ref->ms = ms;
  ref->size = size;
  ref->depth = (U32 *) calloc (ms->size, sizeof (U32));
  ref->index = (U32 *) malloc ((size_t) size * sizeof (U32));
  ref->offset = (U32 *) malloc ((size_t) size * sizeof (U32));
  ref->id = (U32 *) malloc ((size_t) size * sizeof (U32));
  return ref;
}

void mgReferenceDestroy (MgReference *ref)
{
  if (!ref) return;
  mgChainForget (ref);
  free (ref->depth); free (ref->loc); free (ref->rev);
  free (ref->index); free (ref->offset); free (ref->id);
  for (int i = 0 ; i < ref->nSeq ; ++i) free (ref->names[i]);
  free (ref->names); free (ref->len); free (ref);
}

/* the names and lengths of nSeq more sequences; the reference die()s on a duplicate name (modmap.c:102) */
static void refRegister (MgReference *ref, const char **names, const int64_t *offsets, int nSeq)
{
  ref->names = (char **) realloc (ref->names, (size_t) (ref->nSeq + nSeq) * sizeof (char *));
  ref->len = (U32 *) realloc (ref->len, (size_t) (ref->nSeq + nSeq) * sizeof (U32));
  /* an open-addressed set of the names seen so far (the reference keeps them in a DICT): a fragmented assembly
     has 1e5 - 1e6 contigs, so no pairwise comparison */
  const size_t all = (size_t) ref->nSeq + (size_t) nSeq;
  size_t cap = 16; while (cap < 4 * all) cap *= 2;
  int *slot = (int *) calloc (cap, sizeof (int));                       /* 1 + position in ref->names; 0 = empty */
  for (size_t i = 0 ; i < all ; ++i)
    { const char *nm = i < (size_t) ref->nSeq ? ref->names[i] : names[i - ref->nSeq];
      U64 h = 0xcbf29ce484222325ull;
      for (const unsigned char *c = (const unsigned char *) nm ; *c ; ++c) h = (h ^ *c) * 0x100000001b3ull;
      size_t at = (size_t) (h ^ (h >> 29)) & (cap - 1);
      while (slot[at])
        { if (!strcmp (ref->names[slot[at] - 1], nm))
            { fprintf (stderr, "FATAL ERROR: duplicate ref sequence name %s\n", nm); exit (-1); }
          at = (at + 1) & (cap - 1);
        }
      slot[at] = (int) i + 1;
      if (i >= (size_t) ref->nSeq)
        { ref->names[i] = strdup (nm);
          ref->len[i] = (U32) (offsets[i - ref->nSeq + 1] - offsets[i - ref->nSeq]);
        }
    }
  free (slot);
}

/* modmap.c:106-118 for a batch of sequences that is on the device already (2-bit packed, offsets in bases): scan + insert
 * (or lookup) there, the (index, offset, id) of every occurrence appended here */
int mgReferenceAddDevice (MgReference *ref, const U32 *dPacked, U64 totalBases, const U64 *dReadOffsets, int nSeq, const char **names, bool isAdd)
{
  Modset *ms = ref->ms;
  int64_t *offsets = (int64_t *) malloc ((size_t) (nSeq + 1) * 8);
  if (mgMemcpyD2H (offsets, dReadOffsets, (size_t) (nSeq + 1) * 8, 0)) fatal ("D2H");
  refRegister (ref, names, offsets, nSeq);
  free (offsets);
  U64 n = 0;
  void *dIx = 0, *dPos = 0, *dRid = 0;
  U64 guess = totalBases / (U64) ms->hasher->w; guess += guess / 2 + 65536; if (guess > totalBases) guess = totalBases;
  if (guess < 1) guess = 1;
  const int timing = mgKnobs ()->seedTiming == 1;          /* dev */
  struct timespec a0, a1, a2, a3; clock_gettime (CLOCK_MONOTONIC, &a0);
  for (int attempt = 0 ; attempt < 2 ; ++attempt)
    { if (mgDeviceAlloc (&dIx, guess * 4) || mgDeviceAlloc (&dPos, guess * 4) || mgDeviceAlloc (&dRid, guess * 4)) fatal ("device alloc");
      MgStatus s = isAdd ? mgInsertReadsDevice (ms, dPacked, totalBases, dReadOffsets, (U32) nSeq, (U32 *) dIx, (U32 *) dPos, (U32 *) dRid, guess, &n, 0)
                         : mgQueryReadsDevice (ms, dPacked, totalBases, dReadOffsets, (U32) nSeq, (U32 *) dIx, (U32 *) dPos, (U32 *) dRid, guess, &n, 0);
      if (s == MG_OK) break;
      if (s == MG_ERR_CAPACITY && n > guess && attempt == 0)
        { mgDeviceFree (dIx); mgDeviceFree (dPos); mgDeviceFree (dRid); guess = n; continue; }
      if (s == MG_ERR_CAPACITY) { fprintf (stderr, "FATAL ERROR: %s\n", mgLastError ()); exit (-1); }   /* modset.c:58 */
      fatal ("reference scan");
    }
  clock_gettime (CLOCK_MONOTONIC, &a1);
  /* modmap.c:110-117: the occurrences stay on the device (mg_refpack.hip), appended in order behind those of earlier batches */
  U32 added = 0;
  MgStatus as = mgRefBuildAppend (ref, (const U32 *) dIx, (const U32 *) dPos, (const U32 *) dRid, n, (U32) ref->nSeq, &added);
  if (as == MG_ERR_CAPACITY) { fprintf (stderr, "FATAL ERROR: reference size overflow\n"); exit (-1); }      /* modmap.c:111 */
  if (as) fatal ("reference append");
  ref->max += added;
  clock_gettime (CLOCK_MONOTONIC, &a2);
  mgDeviceFree (dIx); mgDeviceFree (dPos); mgDeviceFree (dRid);
  clock_gettime (CLOCK_MONOTONIC, &a3);
  if (timing) fprintf (stderr, "mgReferenceAddDevice: %.3f Gbp, %d sequences: allocations + scan + insert %.1f ms, append %.1f ms, frees %.1f ms\n", totalBases / 1e9, nSeq,
                       (a1.tv_sec - a0.tv_sec) * 1e3 + (a1.tv_nsec - a0.tv_nsec) * 1e-6, (a2.tv_sec - a1.tv_sec) * 1e3 + (a2.tv_nsec - a1.tv_nsec) * 1e-6, (a3.tv_sec - a2.tv_sec) * 1e3 + (a3.tv_nsec - a2.tv_nsec) * 1e-6);
  ref->nSeq += nSeq;
  return 0;
}

/* modmap.c:120-133: the report of a file, the copy classes, the packed arrays.  Classification, loc (exclusive sums) and rev (stable
   sort of the occurrences by index) are made on the device from the occurrences it holds (mg_refpack.hip); the host's arrays -- sized
   as referencePack sizes them (modmap.c:76-81) -- receive them in one piece each */
void mgReferenceFinish (MgReference *ref, U64 totLen, bool isAdd, FILE *out)
{
  Modset *ms = ref->ms;
  fprintf (out, "  %d hashes from %d reference sequences, total length %lld\n", ref->max, ref->nSeq, (long long) totLen);
  const int timing = mgKnobs ()->seedTiming == 1;          /* dev */
  struct timespec f0, f1, f2, f3; clock_gettime (CLOCK_MONOTONIC, &f0);
  { U32 again[3];
    if (mgRefPackedTallies (ref, again))                  /* a further file that added nothing to a packed Reference (one that adds died in the append, modmap.c:111): */
      { fprintf (out, "  %d copy 1, %d copy 2, %d multiple\n", again[0], again[1], again[2]);      /* the classes and the packed arrays are what they were */
        if (isAdd) modsetPack (ms);
        return;
      }
  }
  if (modsetSyncToHost (ms, 0)) fatal ("modsetSyncToHost");       /* value[] of the new entries */
  clock_gettime (CLOCK_MONOTONIC, &f1);
  const U32 n = ref->max ? ref->max : 1, m = ms->max + 1;
  free (ref->index); free (ref->offset); free (ref->id); free (ref->depth); free (ref->rev); free (ref->loc);   /* (nothing was written into them: the occurrences are on the device) */
  ref->index = (U32 *) mgAllocBig ((size_t) n * sizeof (U32));
  ref->offset = (U32 *) mgAllocBig ((size_t) n * sizeof (U32));
  ref->id = (U32 *) mgAllocBig ((size_t) n * sizeof (U32));
  ref->rev = (U32 *) mgAllocBig ((size_t) n * sizeof (U32));
  ref->depth = (U32 *) mgAllocBig ((size_t) (m > ms->size ? m : ms->size) * sizeof (U32));
  ref->loc = (U32 *) mgAllocBig ((size_t) m * sizeof (U32));
  if (!ref->index || !ref->offset || !ref->id || !ref->rev || !ref->depth || !ref->loc) { fprintf (stderr, "FATAL ERROR: out of memory\n"); exit (-1); }
  ref->size = ref->max;
  U32 tal[3];
  clock_gettime (CLOCK_MONOTONIC, &f2);
  if (mgRefBuildFinish (ref, ref->index, ref->offset, ref->id, ref->depth, ref->rev, ref->loc, ms->info, tal)) fatal ("reference pack");
  clock_gettime (CLOCK_MONOTONIC, &f3);
  if (timing) fprintf (stderr, "mgReferenceFinish: modset mirror %.1f ms, host arrays %.1f ms, classes + loc + rev + mirror %.1f ms\n",
                       (f1.tv_sec - f0.tv_sec) * 1e3 + (f1.tv_nsec - f0.tv_nsec) * 1e-6, (f2.tv_sec - f1.tv_sec) * 1e3 + (f2.tv_nsec - f1.tv_nsec) * 1e-6,
                       (f3.tv_sec - f2.tv_sec) * 1e3 + (f3.tv_nsec - f2.tv_nsec) * 1e-6);
  if ((size_t) ms->size > (size_t) m) memset (ref->depth + m, 0, ((size_t) ms->size - m) * sizeof (U32));      /* (the reference's resize keeps ms->size zeroed entries when the set is not packed) */
  fprintf (out, "  %d copy 1, %d copy 2, %d multiple\n", tal[0], tal[1], tal[2]);
  if (isAdd) modsetPack (ms);
}

int mgReferenceRead (MgReference *ref, const char *bases, const int64_t *offsets, int nSeq,
                     const char **names, bool isAdd, FILE *out)
{
  const int timing = mgKnobs ()->seedTiming == 1;          /* dev */
  struct timespec c0, c1, c2, c3; clock_gettime (CLOCK_MONOTONIC, &c0);
  MgDevBatch b; mgBatchUpload (&b, bases, offsets, nSeq);
  clock_gettime (CLOCK_MONOTONIC, &c1);
  mgReferenceAddDevice (ref, (const U32 *) b.dPacked, b.total, (const U64 *) b.dOff, nSeq, names, isAdd);
  mgBatchFree (&b);
  clock_gettime (CLOCK_MONOTONIC, &c2);
  mgReferenceFinish (ref, nSeq ? (U64) offsets[nSeq] : 0, isAdd, out);
  clock_gettime (CLOCK_MONOTONIC, &c3);
#define MSD_(a, b) (((b).tv_sec - (a).tv_sec) * 1e3 + ((b).tv_nsec - (a).tv_nsec) * 1e-6)
  if (timing) fprintf (stderr, "mgReferenceRead: pack + upload %.1f ms, scan + insert + append %.1f ms, finish (mirror, classes, pack) %.1f ms\n", MSD_ (c0, c1), MSD_ (c1, c2), MSD_ (c2, c3));
  return 0;
}

/* ---- the reference's files: <root>.mod + <root>.ref (modmap.c:136-182) ----
 * Both go through the reference's fzopen (utils.c:107-127): gzopen wrapped as a FILE*, so files it
 * writes are gzip streams and it reads gzip or plain alike.  Same here, with glibc's fopencookie in
 * the place of BSD funopen. */
static ssize_t gzCookieRead (void *c, char *buf, size_t n)        /* (gzread takes an unsigned count: a 4 GiB fread -- index[] at table bits 30 -- comes in pieces) */
{ if (n > ((size_t) 1 << 30)) n = (size_t) 1 << 30; int r = gzread ((gzFile) c, buf, (unsigned) n); return r < 0 ? -1 : r; }
static int gzCookieClose (void *c) { return gzclose ((gzFile) c) == Z_OK ? 0 : -1; }

/* the reference's fzopen (utils.c:107-127): "w" a gzip file, "r" a gzip file or a plain one */
FILE *mgFzOpen (const char *name, const char *mode)
{
  FILE *f = 0;
  if (mode[0] == 'w') return mgGzipOpenWrite (name);                /* a gzip file of independent members, deflated by a team of threads (mg_pgzip.c) */
  if ((f = mgGzipOpenRead (name))) return f;                        /* a file of this library's members: they are found by their size fields and inflated in parallel */
  gzFile z = gzopen (name, mode);
  if (z)
    { (void) gzbuffer (z, 1 << 20);
      cookie_io_functions_t io = { gzCookieRead, 0, 0, gzCookieClose };
      f = fopencookie (z, mode, io);
      if (!f) gzclose (z);
    }
  return f;
}

FILE *mgTagOpen (const char *root, const char *tag, const char *mode)      /* utils.c:129-139 */
{
  char *name = (char *) malloc (strlen (root) + strlen (tag) + 2);
  sprintf (name, "%s.%s", root, tag);
  FILE *f = mgFzOpen (name, mode);
  free (name);
  return f;
}

static void die1 (const char *fmt, const char *arg)
{ fprintf (stderr, "FATAL ERROR: "); fprintf (stderr, fmt, arg); fprintf (stderr, "\n"); exit (-1); }

/* The reference keeps sequence lengths in an Array and names in a DICT and dumps both raw
 * (array.c:213-218, dict.c:90-103), so the file holds their in-memory shape: the Array header
 * (with a stale pointer) plus `dim` allocated elements, and the DICT's open-addressed table.
 * These two restate how that shape comes about, so the reference can read what is written here. */
typedef struct { int magic; char *base; int dim, size, max; } RefArrayHeader;     /* array.h:41-50, 32 bytes */
#define REF_ARRAY_MAGIC 8918274                                                    /* array.h:56 */

int mgRefArrayDim (int first, int size, int n)      /* allocated elements after appending 0..n-1 to arrayCreate (first, size): array.c:144-170,180-183 */
{
  int dim = first < 1 ? 1 : first;
  for (int i = 0 ; i < n ; ++i)
    if (i >= dim)
      { if ((long) dim * size < (1 << 23)) dim *= 2; else dim += 1024 + ((1 << 23) / size);
        if (i >= dim) dim = i + 1;
      }
  return dim;
}

static unsigned refNameHash (const char *s, int bits, int forStep)               /* dict.c:44-62 (its fold loop never runs: i starts at bits >= 10 > sizeof(int)) */
{
  const int rot = forStep ? 21 : 13;
  unsigned x = 0;
  for ( ; *s ; ++s) x = (unsigned) (int) *s ^ ((x >> (32 - rot)) | (x << rot));
  x &= (1u << bits) - 1;
  return forStep ? (x | 1) : x;
}

static void refDictPlace (int *table, int bits, char **names, int i)             /* names[i-1] takes entry i */
{
  const unsigned mask = (1u << bits) - 1;
  unsigned x = refNameHash (names[i - 1], bits, 0);
  if (table[x])
    { unsigned d = refNameHash (names[i - 1], bits, 1);
      do x = (x + d) & mask; while (table[x]);
    }
  table[x] = i;
}

/* the DICT table after dictAdd of names[0..n) in order (dict.c:66-74,155-189): 1024 slots, doubled
 * and rebuilt whenever the count exceeds 0.3 of the size */
static int *refDictTable (char **names, int n, int *bitsOut)
{
  int bits = 10, size = 1024;
  int *table = (int *) calloc ((size_t) size, sizeof (int));
  for (int i = 1 ; i <= n ; ++i)
    { refDictPlace (table, bits, names, i);
      if (i > 0.3 * size)
        { ++bits; size *= 2;
          free (table); table = (int *) calloc ((size_t) size, sizeof (int));
          for (int j = 1 ; j <= i ; ++j) refDictPlace (table, bits, names, j);
        }
    }
  *bitsOut = bits;
  return table;
}

#define WR(ptr, sz, cnt, what) do { if (fwrite ((ptr), (sz), (cnt), f) != (size_t) (cnt)) die1 ("failed to write %s", what); } while (0)
#define RD(ptr, sz, cnt, what) do { if (fread ((ptr), (sz), (cnt), f) != (size_t) (cnt)) die1 ("failed to read %s", what); } while (0)

void mgReferenceWrite (MgReference *ref, const char *root)                       /* modmap.c:136-156 */
{
  FILE *f;
  if (!(f = mgTagOpen (root, "mod", "w"))) die1 ("failed to open %s.mod to write", root);
  modsetWrite (ref->ms, f);
  fclose (f);
  if (!(f = mgTagOpen (root, "ref", "w"))) die1 ("failed to open %s.ref to write", root);
  const U32 size = ref->max, m = ref->ms->max + 1;
  WR ("RFMSHv1", 8, 1, "reference header");
  WR (&size, sizeof (U32), 1, "size");
  WR (&ref->max, sizeof (U32), 1, "max");
  WR (ref->index, sizeof (U32), size, "ref index");
  WR (ref->offset, sizeof (U32), size, "ref offset");
  WR (ref->id, sizeof (U32), size, "ref id");
  WR (ref->depth, sizeof (U32), m, "depth");
  WR (ref->rev, sizeof (U32), size, "rev");
  WR (ref->loc, sizeof (U32), m, "loc");
  /* len: Array header + dim elements */
  RefArrayHeader ah; memset (&ah, 0, sizeof (ah));
  ah.magic = REF_ARRAY_MAGIC; ah.dim = mgRefArrayDim (1024, (int) sizeof (U32), ref->nSeq); ah.size = (int) sizeof (U32); ah.max = ref->nSeq;
  U32 *lenBuf = (U32 *) calloc ((size_t) ah.dim, sizeof (U32));
  memcpy (lenBuf, ref->len, (size_t) ref->nSeq * sizeof (U32));
  WR (&ah, sizeof (ah), 1, "ref len");
  WR (lenBuf, sizeof (U32), ah.dim, "ref len");
  free (lenBuf);
  /* dict: dim, max, table, the (max+1) name pointers (meaningless in a file; zero here), then the names */
  int bits; int *table = refDictTable (ref->names, ref->nSeq, &bits);
  char **noPtr = (char **) calloc ((size_t) ref->nSeq + 1, sizeof (char *));
  WR (&bits, sizeof (int), 1, "ref dict");
  WR (&ref->nSeq, sizeof (int), 1, "ref dict");
  WR (table, sizeof (int), (size_t) 1 << bits, "ref dict");
  WR (noPtr, sizeof (char *), (size_t) ref->nSeq + 1, "ref dict");
  for (int i = 0 ; i < ref->nSeq ; ++i)
    { int len = (int) strlen (ref->names[i]);
      WR (&len, sizeof (int), 1, "ref dict");
      if (len) WR (ref->names[i], 1, len, "ref dict");
    }
  free (table); free (noPtr);
  if (fclose (f)) die1 ("failed to close %s.ref", root);
}

MgReference *mgReferenceLoad (const char *root)                                  /* modmap.c:158-182 */
{
  FILE *f;
  if (!(f = mgTagOpen (root, "mod", "r"))) die1 ("failed to open %s.mod to read", root);
  Modset *ms = modsetRead (f);
  fclose (f);
  if (!(f = mgTagOpen (root, "ref", "r"))) die1 ("failed to open %s.ref to read", root);
  char tag[8];
  RD (tag, 8, 1, "reference header");
  if (memcmp (tag, "RFMSHv1", 8)) die1 ("bad reference header%s", "");
  U32 size; RD (&size, sizeof (U32), 1, "size");
  MgReference *ref = mgReferenceCreate (ms, size ? size : 1);
  RD (&ref->max, sizeof (U32), 1, "max");
  const U32 m = ms->max + 1;
  if ((size_t) m > (size_t) ms->size) ref->depth = (U32 *) realloc (ref->depth, (size_t) m * sizeof (U32));   /* the reference under-allocates here (modmap.c:54,171) */
  ref->rev = (U32 *) malloc ((size_t) (size ? size : 1) * sizeof (U32));
  ref->loc = (U32 *) malloc ((size_t) m * sizeof (U32));
  RD (ref->index, sizeof (U32), size, "ref index");
  RD (ref->offset, sizeof (U32), size, "ref offset");
  RD (ref->id, sizeof (U32), size, "ref id");
  RD (ref->depth, sizeof (U32), m, "depth");
  RD (ref->rev, sizeof (U32), size, "rev");
  RD (ref->loc, sizeof (U32), m, "loc");
  RefArrayHeader ah; RD (&ah, sizeof (ah), 1, "ref len");
  if (ah.magic != REF_ARRAY_MAGIC || ah.size != (int) sizeof (U32) || ah.max < 0 || ah.dim < ah.max) die1 ("failed read ref len%s", "");
  U32 *lenBuf = (U32 *) malloc ((size_t) (ah.dim ? ah.dim : 1) * sizeof (U32));
  RD (lenBuf, sizeof (U32), ah.dim, "ref len");
  int bits, nNames; RD (&bits, sizeof (int), 1, "ref dict"); RD (&nNames, sizeof (int), 1, "ref dict");
  if (bits < 10 || bits > 30 || nNames < 0 || nNames > ah.max) die1 ("failed read ref dict%s", "");
  { size_t skip = ((size_t) 1 << bits) * sizeof (int) + ((size_t) nNames + 1) * sizeof (char *);   /* table + stale pointers */
    char *junk = (char *) malloc (skip); RD (junk, 1, skip, "ref dict"); free (junk);
  }
  ref->nSeq = nNames;
  ref->names = (char **) calloc ((size_t) nNames + 1, sizeof (char *));
  ref->len = (U32 *) malloc (((size_t) nNames + 1) * sizeof (U32));
  memcpy (ref->len, lenBuf, (size_t) nNames * sizeof (U32));
  free (lenBuf);
  for (int i = 0 ; i < nNames ; ++i)
    { int len; RD (&len, sizeof (int), 1, "ref dict");
      if (len < 0 || len > (1 << 20)) die1 ("failed read ref dict%s", "");
      ref->names[i] = (char *) calloc ((size_t) len + 1, 1);
      if (len) RD (ref->names[i], 1, len, "ref dict");
    }
  fclose (f);
  return ref;
}

/* one end-of-block test, modmap.c:232-241 (repeated at :245-254 without the "no block" clause) */
static bool blockEnds (const MgReference *ref, U32 loc, U32 loc0, U32 locN, U32 i0, U32 iN, bool withUnset)
{
  if (withUnset && !loc0) return true;
  if (ref->id[loc] != ref->id[loc0]) return true;
  bool end = false;
  if (loc0 < locN)
    { if (loc < locN) end = true;
      int dd = (int) (locN - loc0 - iN + i0); if (dd > 50 || dd < -50) end = true;
    }
  else if (loc0 > locN)
    { if (loc > locN) end = true;
      int dd = (int) (loc0 - locN - iN + i0); if (dd > 50 || dd < -50) end = true;
    }
  return end;
}

static void printM (const MgReference *ref, FILE *out, const char *name, const U32 *seedPos,
                    U32 i0, U32 iN, U32 loc0, U32 locN, int n1, int n2, int copy1)
{
  fprintf (out, "M\t%s\t%d\t%d\t%d\t%s\t%d\t%d\t%d %d\t%.2f\t%.2f\n",
           name, (int) seedPos[i0], (int) seedPos[iN], (int) (seedPos[iN] - seedPos[i0]),
           ref->names[ref->id[loc0]], (int) ref->offset[loc0], (int) ref->offset[locN],
           n1, n2, (n1 + n2) / (double) ((locN > loc0) ? (locN - loc0) : (loc0 - locN)),
           n1 / (double) copy1);
}

/* the long way: seed lists back on the host, tallies and chaining here (used when a read has more blocks
 * than the device kernel keeps) */
static int gVerbose = 0;                          /* modmap's global isVerbose (modmap.c:23), toggled by -v (modmap.c:348) */
void mgSetVerbose (int on) { gVerbose = on != 0; }

static int queryProcessHostChain (MgReference *ref, MgDevBatch *b, const int64_t *offsets, int nReads,
                                  const char **names, FILE *out)
{
  Modset *ms = ref->ms;
  U64 n = 0, guess = b->total / (U64) ms->hasher->w; guess += guess / 2 + 65536; if (guess > b->total) guess = b->total;
  if (guess < 1) guess = 1;
  void *dIx = 0, *dPos = 0, *dRid = 0;
  for (int attempt = 0 ; attempt < 2 ; ++attempt)
    { if (mgDeviceAlloc (&dIx, guess * 4) || mgDeviceAlloc (&dPos, guess * 4) || mgDeviceAlloc (&dRid, guess * 4)) fatal ("device alloc");
      MgStatus s = mgQueryReadsDevice (ms, (U32 *) b->dPacked, b->total, (U64 *) b->dOff, b->nReads, (U32 *) dIx, (U32 *) dPos, (U32 *) dRid, guess, &n, 0);
      if (s == MG_OK) break;
      if (s == MG_ERR_CAPACITY && attempt == 0)
        { mgDeviceFree (dIx); mgDeviceFree (dPos); mgDeviceFree (dRid); guess = n; continue; }
      fatal ("query scan");
    }
  U32 *six = (U32 *) malloc ((size_t) (n + 1) * 4), *spos = (U32 *) malloc ((size_t) (n + 1) * 4), *srid = (U32 *) malloc ((size_t) (n + 1) * 4);
  if (n && (mgMemcpyD2H (six, dIx, n * 4, 0) || mgMemcpyD2H (spos, dPos, n * 4, 0) || mgMemcpyD2H (srid, dRid, n * 4, 0))) fatal ("D2H");
  mgDeviceFree (dIx); mgDeviceFree (dPos); mgDeviceFree (dRid);
  for (U64 i = 0 ; i < n ; ++i) spos[i] &= MG_POS_MASK;

  U64 at = 0;
  for (int r = 0 ; r < nReads ; ++r)
    { U64 first = at;
      while (at < n && srid[at] == (U32) r) ++at;
      U32 ns = (U32) (at - first);
      const U32 *ix = six + first, *ps = spos + first;
      int missed = 0, copy[4] = { 0, 0, 0, 0 };
      for (U32 i = 0 ; i < ns ; ++i)
        if (ix[i]) ++copy[ms->info[ix[i]] & 3]; else ++missed;
      fprintf (out, "Q\t%s\t%llu\t%d miss, %d copy1, %d copy2, %d multi, %.2f hit\n",
               names[r], (unsigned long long) (offsets[r + 1] - offsets[r]), missed, copy[1], copy[2], copy[3],
               ((int) ns - missed) / (double) ((int) ns));
      /* chaining, modmap.c:213-276: occurrence number 0 doubles as "no block open"; a block is
         printed when it ends only with more than two copy-1 seeds, and the block left open at the
         end of the read only with more than two copy-2 seeds */
      U32 loc0 = 0, locN = 0, i0 = 0, iN = 0;
      int n1 = 0, n2 = 0;
      for (U32 i = 0 ; i < ns ; ++i)
        { U32 x = ix[i];
          if (!x || (ms->info[x] & 3) == 3) continue;
          U32 loc = ref->rev[ref->loc[x]];
          bool is1 = (ms->info[x] & 3) == 1;
          if (gVerbose)                                         /* modmap.c:218-229: printf, i.e. stdout whatever outFile is */
            { if (is1) printf ("  %6d\t%s %d\n", (int) ps[i], ref->names[ref->id[loc]], (int) ref->offset[loc]);
              else
                { U32 loc2 = ref->rev[ref->loc[x] + 1];
                  printf ("  %6d\t%s %d\t%s %d\n", (int) ps[i], ref->names[ref->id[loc]], (int) ref->offset[loc],
                          ref->names[ref->id[loc2]], (int) ref->offset[loc2]);
                }
            }
          bool end = blockEnds (ref, loc, loc0, locN, i0, iN, true);
          if (end && loc0 && !is1)
            { loc = ref->rev[ref->loc[x] + 1];
              end = blockEnds (ref, loc, loc0, locN, i0, iN, false);
            }
          if (end)
            { if (n1 > 2) printM (ref, out, names[r], ps, i0, iN, loc0, locN, n1, n2, copy[1]);
              n1 = n2 = 0; loc0 = loc; i0 = i;
            }
          if (is1) ++n1; else ++n2;
          locN = loc; iN = i;
        }
      if (n2 > 2) printM (ref, out, names[r], ps, i0, iN, loc0, locN, n1, n2, copy[1]);
    }
  free (six); free (spos); free (srid);
  return 0;
}

#define MG_QUERY_MAXM 16
/* ---- the "Q" / "M" lines, formatted by a team of threads ----
 * A short-read query file is tens of millions of lines a second at the rate the kernels deliver the tallies, and fprintf does
 * three to five million: the lines of a batch are formatted into per-thread buffers by hand-written conversions and written out
 * in order.  "%d", "%llu", "%s" are what they are; "%.2f" is glibc's: the double's EXACT binary value rounded to two decimals,
 * ties to even -- done here in integer arithmetic on the mantissa (x = m * 2^e, so 100 x = 100 m * 2^e exactly; the quotient
 * and remainder of the shift decide), with snprintf itself for what never occurs at speed (nan from 0 / 0, inf, negative). */
#include <math.h>
#include <pthread.h>
static char *fmtU (char *p, unsigned long long v)
{ char t[24]; int n = 0; do { t[n++] = (char) ('0' + v % 10); v /= 10; } while (v); while (n) *p++ = t[--n]; return p; }
static char *fmtI (char *p, long long v) { if (v < 0) { *p++ = '-'; return fmtU (p, (unsigned long long) (-(v + 1)) + 1); } return fmtU (p, (unsigned long long) v); }
static char *fmtS (char *p, const char *s) { size_t n = strlen (s); memcpy (p, s, n); return p + n; }
static char *fmtF2 (char *p, double x)
{
  if (!(x >= 0) || x >= 1e15 || signbit (x)) return p + snprintf (p, 48, "%.2f", x);
  int e; const double fr = frexp (x, &e);                              /* x = fr * 2^e, 0.5 <= fr < 1 (or x == 0) */
  const unsigned long long mant = (unsigned long long) ldexp (fr, 53);   /* exact: 53 bits */
  const int s = 53 - e;                                                /* x = mant * 2^-s; x < 1e15 < 2^50, so s >= 3 */
  const unsigned long long v = mant * 100;                             /* < 2^60 */
  unsigned long long q = 0;
  if (s < 64)
    { q = v >> s;
      const unsigned long long rem = v & ((1ull << s) - 1), half = 1ull << (s - 1);
      if (rem > half || (rem == half && (q & 1))) ++q;
    }
  const unsigned long long whole = (unsigned long long) (q / 100); const unsigned frac = (unsigned) (q % 100);
  p = fmtU (p, whole); *p++ = '.'; *p++ = (char) ('0' + frac / 10); *p++ = (char) ('0' + frac % 10);
  return p;
}

int mgFormatF2 (char *buf, double x) { char *e = fmtF2 (buf, x); *e = 0; return (int) (e - buf); }      /* test hook: buf of 64 bytes */

/* the buffers of the output stages -- the teams' pieces, the ids that travel with a batch -- come from a small pool and go back to
   it: a fresh 2 MB block from malloc () is 500 page faults when it is written, by sixteen threads at once, and as many pages given
   back when it is freed; 4e6 lines a batch-sequence are 57 000 of each.  The pool keeps up to MG_POOL_SLOTS blocks between calls
   (mgReleaseBuffers () frees them); a block remembers its capacity in the 16 bytes in front of it. */
#define MG_POOL_SLOTS 128
static struct { pthread_mutex_t mu; void *blk[MG_POOL_SLOTS]; int n; } gPool = { PTHREAD_MUTEX_INITIALIZER, { 0 }, 0 };
static size_t poolCap (const void *p) { return *(const size_t *) ((const char *) p - 16); }
static void *poolGet (size_t bytes)
{
  void *hit = 0;
  pthread_mutex_lock (&gPool.mu);
  int pick = -1;
  for (int i = 0 ; i < gPool.n ; ++i)
    { const size_t c = poolCap (gPool.blk[i]);
      if (c >= bytes && (pick < 0 || c < poolCap (gPool.blk[pick]))) pick = i;      /* the tightest fit */
    }
  if (pick >= 0) { hit = gPool.blk[pick]; gPool.blk[pick] = gPool.blk[--gPool.n]; }
  pthread_mutex_unlock (&gPool.mu);
  if (hit) return hit;
  const size_t cap = bytes + bytes / 8 + 64;
  char *raw = (char *) malloc (cap + 16);
  if (!raw) fatal ("out of memory");
  *(size_t *) raw = cap;
  return raw + 16;
}
static void poolPut (void *p)
{
  if (!p) return;
  pthread_mutex_lock (&gPool.mu);
  if (gPool.n < MG_POOL_SLOTS) { gPool.blk[gPool.n++] = p; p = 0; }
  pthread_mutex_unlock (&gPool.mu);
  if (p) free ((char *) p - 16);
}
static void *poolGrow (void *p, size_t keep, size_t bytes)  /* a block of at least `bytes` holding the first `keep` bytes of p */
{
  void *q = poolGet (bytes);
  if (keep) memcpy (q, p, keep);
  poolPut (p);
  return q;
}
static void poolFreeAll (void)
{
  pthread_mutex_lock (&gPool.mu);
  for (int i = 0 ; i < gPool.n ; ++i) free ((char *) gPool.blk[i] - 16);
  gPool.n = 0;
  pthread_mutex_unlock (&gPool.mu);
}

typedef struct { char *buf; size_t len, cap; } FmtBuf;      /* buf: a pool block */
static char *fmtRoom (FmtBuf *b, size_t need)
{ if (b->len + need > b->cap) { b->buf = (char *) poolGrow (b->buf, b->len, 2 * (b->len + need) + 4096); b->cap = poolCap (b->buf); } return b->buf + b->len; }
typedef struct { const MgReference *ref; const MgChainQ *q; const MgChainM *m; const U64 *mStart; const int64_t *offsets; const char **names; int r0, r1; FmtBuf out; double cpuMs; } FmtJob;
static void *fmtQM (void *v)
{
  FmtJob *j = (FmtJob *) v;
  struct timespec u0, u1; clock_gettime (CLOCK_THREAD_CPUTIME_ID, &u0);
  j->out.buf = (char *) poolGet ((size_t) (j->r1 - j->r0) * 80 + 4096); j->out.cap = poolCap (j->out.buf);      /* a Q line of a short read is 60 bytes */
  for (int r = j->r0 ; r < j->r1 ; ++r)
    { const MgChainQ *qq = &j->q[r];
      const char *nm = j->names[r]; const size_t nl = strlen (nm);
      char *p0 = fmtRoom (&j->out, 256 + nl), *p = p0;
      *p++ = 'Q'; *p++ = '\t'; memcpy (p, nm, nl); p += nl; *p++ = '\t';
      p = fmtU (p, (unsigned long long) (j->offsets[r + 1] - j->offsets[r])); *p++ = '\t';
      p = fmtI (p, (int) qq->missed); p = fmtS (p, " miss, "); p = fmtI (p, (int) qq->copy1); p = fmtS (p, " copy1, ");
      p = fmtI (p, (int) qq->copy2); p = fmtS (p, " copy2, "); p = fmtI (p, (int) qq->copyM); p = fmtS (p, " multi, ");
      p = fmtF2 (p, ((int) qq->nSeeds - (int) qq->missed) / (double) ((int) qq->nSeeds)); p = fmtS (p, " hit\n");
      j->out.len += (size_t) (p - p0);
      for (U32 k = 0 ; k < qq->nM ; ++k)
        { const MgChainM *e = &j->m[j->mStart[r] + k];
          const char *rn = j->ref->names[e->id0];
          p0 = fmtRoom (&j->out, 320 + nl + strlen (rn)); p = p0;
          *p++ = 'M'; *p++ = '\t'; memcpy (p, nm, nl); p += nl; *p++ = '\t';
          p = fmtI (p, (int) e->pos0); *p++ = '\t'; p = fmtI (p, (int) e->posN); *p++ = '\t'; p = fmtI (p, (int) (e->posN - e->pos0)); *p++ = '\t';
          p = fmtS (p, rn); *p++ = '\t'; p = fmtI (p, (int) e->off0); *p++ = '\t'; p = fmtI (p, (int) e->offN); *p++ = '\t';
          p = fmtI (p, e->n1); *p++ = ' '; p = fmtI (p, e->n2); *p++ = '\t';
          p = fmtF2 (p, (e->n1 + e->n2) / (double) e->span); *p++ = '\t';
          p = fmtF2 (p, e->n1 / (double) (int) qq->copy1); *p++ = '\n';
          j->out.len += (size_t) (p - p0);
        }
    }
  clock_gettime (CLOCK_THREAD_CPUTIME_ID, &u1);
  j->cpuMs = (u1.tv_sec - u0.tv_sec) * 1e3 + (u1.tv_nsec - u0.tv_nsec) * 1e-6;
  return 0;
}

static int fmtThreads (int nReads)
{
  if (nReads < 20000) return 1;
  long v = mgCpuBudget ();
  const long pk = mgKnobs ()->parseThreads; if (pk != MG_KNOB_UNSET && pk > 0) v = pk;
  if (v > 16) v = 16;
  if (v < 1) v = 1;
  return (int) v;
}

#define MS_(a, b) (((b).tv_sec - (a).tv_sec) * 1e3 + ((b).tv_nsec - (a).tv_nsec) * 1e-6)
/* the lines of a batch whose tallies and blocks are on the host: formatted by a team of threads, each its range of the reads
   into its own buffer (piece[0 .. *nPieces)), and written in order.  (Positioned writes of the pieces by the team itself were tried:
   writes to one file take turns on its inode lock, 234 MB take the 40 ms one fwrite takes.) */
static void queryFormatLines (const MgReference *ref, const MgChainQ *q, const MgChainM *m, const int64_t *offsets, int nReads, const char **names,
                              FmtBuf piece[16], int *nPieces, double *ms, double *cpuMs)
{
  struct timespec c1, c2; clock_gettime (CLOCK_MONOTONIC, &c1);
  U64 *mStart = (U64 *) poolGet (((size_t) nReads + 1) * 8);
  { U64 tot = 0; for (int r = 0 ; r < nReads ; ++r) { mStart[r] = tot; tot += q[r].nM; } mStart[nReads] = tot; }
  const int T = fmtThreads (nReads);
  FmtJob job[16]; pthread_t th[16]; int started[16];
  for (int t = 0 ; t < T ; ++t)
    { memset (&job[t], 0, sizeof (FmtJob));
      job[t].ref = ref; job[t].q = q; job[t].m = m; job[t].mStart = mStart; job[t].offsets = offsets; job[t].names = names;
      job[t].r0 = (int) ((int64_t) nReads * t / T); job[t].r1 = (int) ((int64_t) nReads * (t + 1) / T);
      started[t] = t && pthread_create (&th[t], 0, fmtQM, &job[t]) == 0;
    }
  for (int t = 0 ; t < T ; ++t) if (!started[t]) fmtQM (&job[t]);
  for (int t = 0 ; t < T ; ++t) if (started[t]) pthread_join (th[t], 0);
  for (int t = 0 ; t < T ; ++t) { piece[t] = job[t].out; if (cpuMs) *cpuMs += job[t].cpuMs; }
  *nPieces = T;
  poolPut (mStart);
  clock_gettime (CLOCK_MONOTONIC, &c2);
  if (ms) *ms += MS_ (c1, c2);
}
static void queryWritePieces (FmtBuf piece[16], int nPieces, FILE *out, double *ms)
{
  struct timespec c1, c2; clock_gettime (CLOCK_MONOTONIC, &c1);
  for (int t = 0 ; t < nPieces ; ++t)
    { if (piece[t].len && fwrite (piece[t].buf, 1, piece[t].len, out) != piece[t].len) fatal ("write");
      poolPut (piece[t].buf); piece[t].buf = 0;
    }
  clock_gettime (CLOCK_MONOTONIC, &c2);
  if (ms) *ms += MS_ (c1, c2);
}

/* modmap.c:188-281.  Scan, lookup, the tallies of the "Q" line and the chaining into "M" blocks all run on the
 * device (mg_chain.hip: one lane per read); the host formats the lines from a few integers per read and block.
 * The batch is on the device already (2-bit packed, offsets in bases): mgQueryProcess uploads one, the file entry point
 * (mgQueryFile) gets its batches from the device text parser. */
int mgQueryProcessDevice (MgReference *ref, const U32 *dPacked, U64 totalBases, const U64 *dReadOffsets, int nReads, const char **names, FILE *out)
{
  if (nReads <= 0) return 0;
  const int timing = mgKnobs ()->seedTiming == 1;          /* dev */
  struct timespec c0, c1; if (timing) clock_gettime (CLOCK_MONOTONIC, &c0);
  if (modsetSyncToHost (ref->ms, 0)) fatal ("modsetSyncToHost");
  int64_t *offsets = (int64_t *) malloc ((size_t) (nReads + 1) * 8);
  if (mgMemcpyD2H (offsets, dReadOffsets, (size_t) (nReads + 1) * 8, 0)) fatal ("D2H");
  MgDevBatch b; b.dPacked = (void *) dPacked; b.dOff = (void *) dReadOffsets; b.total = totalBases; b.nReads = (U32) nReads;
  MgChainQ *q = (MgChainQ *) malloc ((size_t) nReads * sizeof (MgChainQ));
  MgChainM *m = 0;                                         /* all reads' blocks, densely, in read order */
  const int hostChain = mgKnobs ()->queryHostChain == 1;   /* test knob */
  int rc = (hostChain || gVerbose) ? 1 : mgChainQueryDevice (ref, dPacked, totalBases, dReadOffsets, (U32) nReads, q, &m, MG_QUERY_MAXM, 0);
  if (rc < 0) fatal ("query");
  if (timing) clock_gettime (CLOCK_MONOTONIC, &c1);
  if (rc == 1) rc = queryProcessHostChain (ref, &b, offsets, nReads, names, out);
  else
    { double msF = 0, msW = 0;
      FmtBuf piece[16]; int nPieces = 0;
      queryFormatLines (ref, q, m, offsets, nReads, names, piece, &nPieces, &msF, 0);
      queryWritePieces (piece, nPieces, out, &msW);
      if (timing)
        fprintf (stderr, "mgQueryProcessDevice: %d reads: device (scan, lookups, chain, copies) %.1f ms, format %.1f ms (%d threads), write %.1f ms\n",
                 nReads, MS_ (c0, c1), msF, nPieces, msW);
    }
  free (q); free (m); free (offsets);
  return rc;
}

/* ---- the file entry point's batches: three stages, three threads ----
 * mgQueryFile hands over a batch per window of the file.  Push runs the device half (scan, lookups, tallies, chaining, the copies
 * to the host) on the caller's thread -- which goes on to read, parse and query the next window -- and queues the rest: a
 * FORMATTER thread (with its team) turns a batch's tallies and blocks into lines, a WRITER thread puts them out; batches pass
 * through both in order, at most two waiting in front of each.  A batch that needs the long way (-v, a read with more blocks than
 * the chain kernel keeps) waits for both to be idle and is done on the spot, so the order of the lines is the file's. */
typedef struct MgQJob
{ struct MgQJob *next; MgChainQ *q; MgChainM *m; int64_t *offsets; char *idBytes; U64 *idOff; int nReads;
  int pinQ, pinOff;                                        /* q / offsets are blocks of the pipe's page-locked pool (slot + 1), not malloc ()ed */
  FmtBuf piece[16]; int nPieces;
} MgQJob;
#define MG_QPIPE_PINS 10
typedef struct { MgQJob *head, *tail; int n; } MgQQueue;
struct MgQueryPipe
{ MgReference *ref; FILE *out; pthread_t thFormat, thWrite; int started;
  pthread_mutex_t mu; pthread_cond_t cv;
  MgQQueue toFormat, toWrite; int pending;                 /* batches anywhere between Push and the last fwrite */
  int closing;
  double msDevice, msFormat, msFormatCpu, msWrite, msWait; int nBatches;      /* (dev timing) */
};

/* page-locked blocks the device halves copy into: kept between calls like the parser's windows (page-locking 10 MB takes 2 ms),
   given back by mgReleaseBuffers ().  A block of at least `bytes` for a job (*slot = its number + 1), or malloc () memory
   (*slot = 0) when the pool has none */
static struct { pthread_mutex_t mu; void *pin[MG_QPIPE_PINS]; size_t cap[MG_QPIPE_PINS]; int busy[MG_QPIPE_PINS]; } gPin = { PTHREAD_MUTEX_INITIALIZER, { 0 }, { 0 }, { 0 } };
static void *pipePinGet (size_t bytes, int *slot)
{
  void *r = 0; *slot = 0;
  pthread_mutex_lock (&gPin.mu);
  int pick = -1;
  for (int i = 0 ; i < MG_QPIPE_PINS ; ++i) if (!gPin.busy[i] && gPin.cap[i] >= bytes && (pick < 0 || gPin.cap[i] < gPin.cap[pick])) pick = i;
  if (pick < 0) for (int i = 0 ; i < MG_QPIPE_PINS ; ++i) if (!gPin.busy[i] && (pick < 0 || gPin.cap[i] < gPin.cap[pick])) pick = i;      /* the smallest free one grows */
  if (pick >= 0) gPin.busy[pick] = 1;
  pthread_mutex_unlock (&gPin.mu);
  if (pick >= 0 && gPin.cap[pick] < bytes)                 /* (busy: nobody else looks at it) */
    { mgPinnedFree (gPin.pin[pick]); gPin.cap[pick] = 0;
      gPin.pin[pick] = mgPinnedAlloc (bytes + bytes / 4);
      if (gPin.pin[pick]) gPin.cap[pick] = bytes + bytes / 4;
      else { pthread_mutex_lock (&gPin.mu); gPin.busy[pick] = 0; pthread_mutex_unlock (&gPin.mu); pick = -1; }
    }
  if (pick >= 0) { r = gPin.pin[pick]; *slot = pick + 1; }
  else { r = malloc (bytes); if (!r) fatal ("out of memory"); }
  return r;
}
static void pipePinPut (void *ptr, int slot)
{
  if (!slot) { free (ptr); return; }
  pthread_mutex_lock (&gPin.mu); gPin.busy[slot - 1] = 0; pthread_mutex_unlock (&gPin.mu);
}
void mgQueryReleaseBuffers (void)
{
  pthread_mutex_lock (&gPin.mu);
  for (int i = 0 ; i < MG_QPIPE_PINS ; ++i) if (!gPin.busy[i] && gPin.pin[i]) { mgPinnedFree (gPin.pin[i]); gPin.pin[i] = 0; gPin.cap[i] = 0; }
  pthread_mutex_unlock (&gPin.mu);
  poolFreeAll ();
  mgChainReleaseBuffers ();
}

static void queryJobFree (MgQueryPipe *p, MgQJob *j)
{ pipePinPut (j->q, j->pinQ); pipePinPut (j->offsets, j->pinOff); free (j->m); poolPut (j->idBytes); poolPut (j->idOff);
  for (int t = 0 ; t < j->nPieces ; ++t) poolPut (j->piece[t].buf);
  free (j);
}

/* (all three with p->mu held) */
static void qPut (MgQQueue *q, MgQJob *j) { j->next = 0; if (q->tail) q->tail->next = j; else q->head = j; q->tail = j; ++q->n; }
static MgQJob *qTake (MgQQueue *q) { MgQJob *j = q->head; if (j) { q->head = j->next; if (!q->head) q->tail = 0; --q->n; } return j; }

static void *queryFormatter (void *v)
{
  MgQueryPipe *p = (MgQueryPipe *) v;
  for (;;)
    { pthread_mutex_lock (&p->mu);
      while (!p->toFormat.head && !p->closing) pthread_cond_wait (&p->cv, &p->mu);
      MgQJob *j = qTake (&p->toFormat);
      pthread_mutex_unlock (&p->mu);
      if (!j) break;                                       /* closing, nothing left */
      const char **names = (const char **) poolGet (((size_t) j->nReads + 1) * sizeof (char *));
      for (int r = 0 ; r < j->nReads ; ++r) names[r] = j->idBytes + j->idOff[r];
      queryFormatLines (p->ref, j->q, j->m, j->offsets, j->nReads, names, j->piece, &j->nPieces, &p->msFormat, &p->msFormatCpu);
      poolPut (names);
      pipePinPut (j->q, j->pinQ); pipePinPut (j->offsets, j->pinOff); free (j->m); poolPut (j->idBytes); poolPut (j->idOff);
      j->q = 0; j->m = 0; j->offsets = 0; j->idBytes = 0; j->idOff = 0; j->pinQ = j->pinOff = 0;
      pthread_mutex_lock (&p->mu);
      while (p->toWrite.n >= 2) pthread_cond_wait (&p->cv, &p->mu);
      qPut (&p->toWrite, j);
      pthread_cond_broadcast (&p->cv);
      pthread_mutex_unlock (&p->mu);
    }
  pthread_mutex_lock (&p->mu); p->closing = 2; pthread_cond_broadcast (&p->cv); pthread_mutex_unlock (&p->mu);      /* the writer's turn to finish */
  return 0;
}

static void *queryWriter (void *v)
{
  MgQueryPipe *p = (MgQueryPipe *) v;
  for (;;)
    { pthread_mutex_lock (&p->mu);
      while (!p->toWrite.head && p->closing != 2) pthread_cond_wait (&p->cv, &p->mu);
      MgQJob *j = qTake (&p->toWrite);
      if (j) pthread_cond_broadcast (&p->cv);             /* room in front of the writer */
      pthread_mutex_unlock (&p->mu);
      if (!j) return 0;
      queryWritePieces (j->piece, j->nPieces, p->out, &p->msWrite);
      queryJobFree (p, j);
      pthread_mutex_lock (&p->mu);
      --p->pending;
      pthread_cond_broadcast (&p->cv);
      pthread_mutex_unlock (&p->mu);
    }
}

MgQueryPipe *mgQueryPipeOpen (MgReference *ref, FILE *out)
{
  MgQueryPipe *p = (MgQueryPipe *) calloc (1, sizeof (MgQueryPipe));
  if (!p) fatal ("out of memory");
  p->ref = ref; p->out = out;
  pthread_mutex_init (&p->mu, 0); pthread_cond_init (&p->cv, 0);
  if (pthread_create (&p->thFormat, 0, queryFormatter, p) == 0)
    { if (pthread_create (&p->thWrite, 0, queryWriter, p) == 0) p->started = 1;
      else
        { pthread_mutex_lock (&p->mu); p->closing = 1; pthread_cond_broadcast (&p->cv); pthread_mutex_unlock (&p->mu);
          pthread_join (p->thFormat, 0); p->closing = 0;
        }
    }                                                      /* (no threads: Push does everything on the spot) */
  mgChainScratchKeep (1);
  return p;
}

static void queryPipeDrain (MgQueryPipe *p)
{
  pthread_mutex_lock (&p->mu);
  while (p->pending) pthread_cond_wait (&p->cv, &p->mu);
  pthread_mutex_unlock (&p->mu);
}

int mgQueryPipePush (MgQueryPipe *p, const U32 *dPacked, U64 totalBases, const U64 *dReadOffsets, int nReads, const char *idBytes, const U64 *idOff)
{
  if (nReads <= 0) return 0;
  struct timespec c0, c1, c2; clock_gettime (CLOCK_MONOTONIC, &c0);
  const int hostChain = mgKnobs ()->queryHostChain == 1;
  int rc = 1;
  MgQJob *j = 0;
  if (!hostChain && !gVerbose && p->started)
    { if (modsetSyncToHost (p->ref->ms, 0)) fatal ("modsetSyncToHost");
      j = (MgQJob *) calloc (1, sizeof (MgQJob));
      if (!j) fatal ("out of memory");
      j->nReads = nReads;
      j->offsets = (int64_t *) pipePinGet ((size_t) (nReads + 1) * 8, &j->pinOff);
      j->q = (MgChainQ *) pipePinGet ((size_t) nReads * sizeof (MgChainQ), &j->pinQ);
      if (j->pinOff ? mgCopyOutPinned (j->offsets, dReadOffsets, (size_t) (nReads + 1) * 8, 0) : mgMemcpyD2H (j->offsets, dReadOffsets, (size_t) (nReads + 1) * 8, 0)) fatal ("D2H");
      rc = mgChainQueryDevice (p->ref, dPacked, totalBases, dReadOffsets, (U32) nReads, j->q, &j->m, MG_QUERY_MAXM, j->pinQ != 0);
      if (rc < 0) fatal ("query");
    }
  clock_gettime (CLOCK_MONOTONIC, &c1);
  p->msDevice += MS_ (c0, c1); ++p->nBatches;
  if (rc == 1)                                             /* the long way, on the spot, once the lines before it are out */
    { if (j) queryJobFree (p, j);
      queryPipeDrain (p);
      const char **names = (const char **) malloc (((size_t) nReads + 1) * sizeof (char *));
      if (!names) fatal ("out of memory");
      for (int r = 0 ; r < nReads ; ++r) names[r] = idBytes + idOff[r];
      rc = mgQueryProcessDevice (p->ref, dPacked, totalBases, dReadOffsets, nReads, names, p->out);
      free (names);
      return rc;
    }
  /* the ids belong to the parser, which moves on: a copy goes with the job */
  const size_t idLen = (size_t) idOff[nReads - 1] + strlen (idBytes + idOff[nReads - 1]) + 1;
  j->idBytes = (char *) poolGet (idLen); j->idOff = (U64 *) poolGet ((size_t) nReads * 8);
  memcpy (j->idBytes, idBytes, idLen); memcpy (j->idOff, idOff, (size_t) nReads * 8);
  pthread_mutex_lock (&p->mu);
  while (p->toFormat.n >= 2) pthread_cond_wait (&p->cv, &p->mu);
  qPut (&p->toFormat, j); ++p->pending;
  pthread_cond_broadcast (&p->cv);
  pthread_mutex_unlock (&p->mu);
  clock_gettime (CLOCK_MONOTONIC, &c2);
  p->msWait += MS_ (c1, c2);
  return 0;
}

void mgQueryPipeClose (MgQueryPipe *p)
{
  if (!p) return;
  struct timespec c0, c1; clock_gettime (CLOCK_MONOTONIC, &c0);
  if (p->started)
    { pthread_mutex_lock (&p->mu); p->closing = 1; pthread_cond_broadcast (&p->cv); pthread_mutex_unlock (&p->mu);
      pthread_join (p->thFormat, 0); pthread_join (p->thWrite, 0);
    }
  clock_gettime (CLOCK_MONOTONIC, &c1);
  mgChainScratchKeep (0);
  if (mgKnobs ()->seedTiming == 1)
    fprintf (stderr, "mgQueryFile: %d batches: device halves %.1f ms, waiting for room in the queue %.1f + %.1f ms at the end; formatter %.1f ms (its threads' CPU time %.1f ms), writer %.1f ms\n",
             p->nBatches, p->msDevice, p->msWait, MS_ (c0, c1), p->msFormat, p->msFormatCpu, p->msWrite);
  pthread_mutex_destroy (&p->mu); pthread_cond_destroy (&p->cv);
  free (p);
}
#undef MS_

int mgQueryProcess (MgReference *ref, const char *bases, const int64_t *offsets, int nReads,
                    const char **names, FILE *out)
{
  if (nReads <= 0) return 0;
  MgDevBatch b; mgBatchUpload (&b, bases, offsets, nReads);
  int rc = mgQueryProcessDevice (ref, (const U32 *) b.dPacked, b.total, (const U64 *) b.dOff, nReads, names, out);
  mgBatchFree (&b);
  return rc;
}
