/* mg_seqio.c — the front end of the hot path: FASTA / FASTQ text (plain, gzip, blocked gzip) -> bases 0..3.
 *
 * Restates what the reference's callers get from seqIOopenRead + seqIOread (seqio.c:30-110,187-346)
 * with dna2indexConv after their N->0 patch (seqio.c:643-652, modutils.c:39, modmap.c:97):
 *   - the type is decided by the first byte: '>' FASTA, '@' FASTQ (seqio.c:47-73); the binary,
 *     ONEcode and BAM inputs of seqio.c are not handled here;
 *   - id = the characters after the marker up to the first white space (seqio.c:302-304);
 *   - FASTA: the sequence is everything up to the next line that starts with '>'; A/a C/c G/g T/t
 *     N/n become 0 1 2 3 0, every other byte (line ends included) is dropped, so the sequence
 *     shortens (seqio.c:316-323);
 *   - FASTQ: four lines; the sequence line is converted in place, bytes outside ACGTN stay in the
 *     sequence as (char)-2 (seqio.c:325-331); the '+' line and the quality length are checked
 *     (seqio.c:333-339);
 *   - a last record that is not closed by a newline is reported ("incomplete sequence record line
 *     N") and not returned (seqio.c:213-217).
 * Unlike the reference (one record at a time, one thread) a batch of records is cut out of the
 * text and converted by a pool of threads (only the last record of a window, which may be unfinished,
 * is cut serially); batches stream, so memory is bounded by the batch size, not by the file.
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <time.h>
#include "modgpu.h"
#include "mg_internal.h"
static double nowS (void) { struct timespec t; clock_gettime (CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
static double gPhase[8]; static const char *gPhaseName[8];
#define TIMING(slot, tag) do { double t_ = nowS (); gPhase[slot] += t_ - tLast; gPhaseName[slot] = tag; tLast = t_; } while (0)

#define UNIT_BYTES ((size_t) 1 << 20)        /* raw text per conversion work unit */

struct MgSeqReader {
  gzFile gz;                                 /* gzip stream, or 0: plain text (or BGZF blocks) read straight from fd */
  int fd;
  int bgzf; size_t bgzfOff;                  /* blocked gzip: file offset of the next block */
  char *cbuf; size_t ccap;                   /* blocked gzip: compressed bytes being taken apart */
  size_t fileSize, consumed;                 /* plain regular file: its size, and bytes read so far */
  char *buf; size_t cap, len, pos;           /* raw text window: [pos, len) not yet consumed */
  int eof, isFastq, finished;
  U64 line;                                  /* 1-based line number at buf[pos] */
  U64 nSeq;
  int nThreads;
} ;

/* Large buffers are touched once, front to back: first-touch page faults cost more than the parsing,
 * so they come straight from mmap and the last two given back are kept for the next batch -- and for the next
 * file: unmapping them when a reader closes cost 0.10 s of the 0.29 s a 4 Gbp FASTA file takes end to end, and mapping
 * and touching them again as much at the next open.  mgSeqReleaseBuffers () gives them back (also run at unload).
 * (Requesting transparent huge pages was tried: direct compaction made first touch 20x slower.) */
static pthread_mutex_t bigMu = PTHREAD_MUTEX_INITIALIZER;
static int bigReaders;                           /* open readers */
static struct { void *p; size_t n; unsigned long age; } bigSpare[2];   /* the last buffers given back: batches alternate between two sizes */
static unsigned long bigClock;

static void *bigAlloc (size_t n)
{
  if (!n) n = 1;
  pthread_mutex_lock (&bigMu);
  for (int i = 0 ; i < 2 ; ++i)
    if (bigSpare[i].p && bigSpare[i].n >= n && bigSpare[i].n / 2 <= n)
      { void *p = bigSpare[i].p; bigSpare[i].p = 0;
        pthread_mutex_unlock (&bigMu);
        return p;                              /* its mapped length stays bigSpare[i].n: see bigSize */
      }
  pthread_mutex_unlock (&bigMu);
  void *p = mmap (0, n + 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (p == MAP_FAILED) { fprintf (stderr, "FATAL ERROR: sequence buffer of %zu bytes\n", n); exit (-1); }
#ifdef MADV_HUGEPAGE
  if (n >= ((size_t) 8 << 20)) (void) madvise (p, n + 4096, MADV_HUGEPAGE);      /* first touch by 2 MiB pages where the kernel allows: 512x fewer faults */
#endif
  *(size_t *) p = n + 4096;                     /* mapped length, kept in a leading page */
  return (char *) p + 4096;
}
static void bigRelease (void *p) { char *base = (char *) p - 4096; munmap (base, *(size_t *) base); }
static void bigFree (void *p, size_t unused)
{
  (void) unused;
  if (!p) return;
  size_t n = *(size_t *) ((char *) p - 4096) - 4096;
  void *drop = p;
  pthread_mutex_lock (&bigMu);
  for (int i = 0 ; i < 2 ; ++i) if (!bigSpare[i].p) { bigSpare[i].p = p; bigSpare[i].n = n; bigSpare[i].age = ++bigClock; drop = 0; break; }
  if (drop)                                                      /* both slots taken: the one given back longer ago goes (sizes change from file to file) */
    { const int i = bigSpare[0].age < bigSpare[1].age ? 0 : 1;
      drop = bigSpare[i].p; bigSpare[i].p = p; bigSpare[i].n = n; bigSpare[i].age = ++bigClock;
    }
  pthread_mutex_unlock (&bigMu);
  if (drop) bigRelease (drop);
}
static void bigFlush (void)                     /* mgSeqClose: one reader less (its buffers stay in the spare slots) */
{
  pthread_mutex_lock (&bigMu);
  --bigReaders;
  pthread_mutex_unlock (&bigMu);
}
/* give the (at most two) spare buffers back to the system; with a reader open they would only be mapped again */
__attribute__ ((destructor)) void mgSeqReleaseBuffers (void)
{
  pthread_mutex_lock (&bigMu);
  if (bigReaders <= 0)
    for (int i = 0 ; i < 2 ; ++i) if (bigSpare[i].p) { bigRelease (bigSpare[i].p); bigSpare[i].p = 0; bigSpare[i].n = 0; }
  pthread_mutex_unlock (&bigMu);
}

static signed char convTable[256];
static pthread_once_t convOnce = PTHREAD_ONCE_INIT;
static void convInit (void)
{
  memset (convTable, -2, sizeof (convTable));
  convTable['A'] = convTable['a'] = 0; convTable['C'] = convTable['c'] = 1;
  convTable['G'] = convTable['g'] = 2; convTable['T'] = convTable['t'] = 3;
  convTable['N'] = convTable['n'] = 0;       /* the callers' patch */
}

static void dieLine (const char *fmt, U64 line)
{ fprintf (stderr, "FATAL ERROR: "); fprintf (stderr, fmt, (unsigned long long) line); fprintf (stderr, "\n"); exit (-1); }

static int threadCount (void)
{
  const long pk = mgKnobs ()->parseThreads;
  const int e = pk != MG_KNOB_UNSET;
  const long budget = mgCpuBudget ();                 /* the CPUs this process may really use: affinity mask and cgroup quota, not what is online */
  long n = e ? pk : budget;
  if (n < 1) n = 1;
  if (n > 32) n = 32;
  return (int) n;
}

static void runPool (int n, void *(*fn) (void *), void *arg);
static int threadCount (void);

/* ---- blocked gzip (BGZF: bgzip, htslib) ----
 * A gzip file made of members of at most 64 KiB, each carrying its compressed length in a 'BC' extra
 * field and, like every gzip member, its CRC and uncompressed length at its end.  zlib reads such a file
 * as any other (which is what the reference does, seqio.c:33-40, one thread); here the members of a
 * stretch of the file are inflated by the pool, each straight to its place in the text window.
 * A member that is not a BGZF block hands the rest of the file to zlib's stream reader. */
typedef struct { size_t cOff, cLen, uOff, uLen; } BgBlock;
typedef struct { const unsigned char *src; char *dst; const BgBlock *blocks; size_t n, next; int bad; } BgJob;

static size_t bgzfBlockSize (const unsigned char *p, size_t avail, size_t *payloadOff)     /* 0: not a block header */
{
  if (avail < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || p[3] != 4) return 0;
  const size_t xlen = (size_t) p[10] | ((size_t) p[11] << 8);
  if (avail < 12 + xlen) return 0;
  for (size_t at = 12 ; at + 4 <= 12 + xlen ; )
    { const size_t len = (size_t) p[at + 2] | ((size_t) p[at + 3] << 8);
      if (p[at] == 'B' && p[at + 1] == 'C' && len == 2 && at + 6 <= 12 + xlen)
        { *payloadOff = 12 + xlen; return ((size_t) p[at + 4] | ((size_t) p[at + 5] << 8)) + 1; }
      at += 4 + len;
    }
  return 0;
}

static void *bgzfWorker (void *arg)
{
  BgJob *j = (BgJob *) arg;
  z_stream zs; memset (&zs, 0, sizeof (zs));
  if (inflateInit2 (&zs, -15) != Z_OK) { j->bad = 1; return 0; }
  for (;;)
    { size_t b0 = __atomic_fetch_add (&j->next, 16, __ATOMIC_RELAXED);
      if (b0 >= j->n) break;
      size_t b1 = b0 + 16 < j->n ? b0 + 16 : j->n;
      for (size_t b = b0 ; b < b1 ; ++b)
        { const BgBlock *k = &j->blocks[b];
          const unsigned char *blk = j->src + k->cOff;
          size_t pay = 0; (void) bgzfBlockSize (blk, k->cLen, &pay);
          inflateReset (&zs);
          zs.next_in = (Bytef *) (blk + pay); zs.avail_in = (uInt) (k->cLen - pay - 8);
          zs.next_out = (Bytef *) (j->dst + k->uOff); zs.avail_out = (uInt) k->uLen;
          const int rc = inflate (&zs, Z_FINISH);
          const unsigned char *t = blk + k->cLen - 8;
          const U32 crc = (U32) t[0] | ((U32) t[1] << 8) | ((U32) t[2] << 16) | ((U32) t[3] << 24);
          if (rc != Z_STREAM_END || zs.total_out != k->uLen || zs.avail_in != 0
              || (U32) crc32 (crc32 (0L, Z_NULL, 0), (const Bytef *) (j->dst + k->uOff), (uInt) k->uLen) != crc)
            __atomic_store_n (&j->bad, 1, __ATOMIC_RELAXED);
        }
    }
  inflateEnd (&zs);
  return 0;
}

static void bgzfToStream (MgSeqReader *r)          /* the rest of the file is ordinary gzip: zlib's reader takes over */
{
  r->bgzf = 0;
  (void) lseek (r->fd, (off_t) r->bgzfOff, SEEK_SET);
  r->gz = gzdopen (r->fd, "r");
  if (!r->gz) { fprintf (stderr, "FATAL ERROR: gzip stream at offset %zu\n", r->bgzfOff); exit (-1); }
  gzbuffer (r->gz, 1 << 20);
}

/* whole blocks, as many as fit in `room` (at least 64 KiB), inflated into dst; 0 at the end of the file */
static size_t bgzfRead (MgSeqReader *r, char *dst, size_t room)
{
  for (;;)
    { size_t want = room / 3 + ((size_t) 128 << 10);                /* compressed bytes to look at */
      if (want > ((size_t) 512 << 20)) want = (size_t) 512 << 20;
      if (want > r->ccap) { free (r->cbuf); r->cbuf = (char *) malloc (want); r->ccap = want; }
      size_t have = 0;
      while (have < want)
        { ssize_t g = pread (r->fd, r->cbuf + have, want - have, (off_t) (r->bgzfOff + have));
          if (g <= 0) break;
          have += (size_t) g;
        }
      if (!have) return 0;
      const unsigned char *c = (const unsigned char *) r->cbuf;
      size_t cap = 1024, n = 0, cAt = 0, uAt = 0; int foreign = 0;
      BgBlock *blocks = (BgBlock *) malloc (cap * sizeof (BgBlock));
      while (cAt < have)
        { size_t pay = 0; const size_t len = bgzfBlockSize (c + cAt, have - cAt, &pay);
          if (!len)
            { if (have - cAt >= 18 + 65536 || have < want) foreign = 1;     /* a whole header is in view and it is not a block's */
              break;
            }
          if (cAt + len > have) { if (have < want) foreign = 1; break; }    /* the file ends inside the block: zlib reports it */
          if (len < pay + 8 + 2) { foreign = 1; break; }
          const unsigned char *t = c + cAt + len - 4;
          const size_t uLen = (size_t) t[0] | ((size_t) t[1] << 8) | ((size_t) t[2] << 16) | ((size_t) t[3] << 24);
          if (uLen > 65536) { foreign = 1; break; }
          if (uAt + uLen > room) break;
          if (n == cap) { cap *= 2; blocks = (BgBlock *) realloc (blocks, cap * sizeof (BgBlock)); }
          blocks[n].cOff = cAt; blocks[n].cLen = len; blocks[n].uOff = uAt; blocks[n].uLen = uLen; ++n;
          cAt += len; uAt += uLen;
        }
      if (n)
        { BgJob j; j.src = c; j.dst = dst; j.blocks = blocks; j.n = n; j.next = 0; j.bad = 0;
          int nt = r->nThreads > 0 ? r->nThreads : threadCount ();
          if ((size_t) nt > n / 16 + 1) nt = (int) (n / 16 + 1);
          runPool (nt, bgzfWorker, &j);
          if (j.bad) { fprintf (stderr, "FATAL ERROR: corrupt BGZF block near file offset %zu\n", r->bgzfOff); exit (-1); }
        }
      free (blocks);
      r->bgzfOff += cAt;
      if (foreign && !uAt) { bgzfToStream (r); int got = gzread (r->gz, dst, (unsigned) (room > ((size_t) 1 << 30) ? (size_t) 1 << 30 : room)); return got > 0 ? (size_t) got : 0; }
      if (uAt) return uAt;
      if (!cAt) { fprintf (stderr, "FATAL ERROR: BGZF block at file offset %zu does not fit the window\n", r->bgzfOff); exit (-1); }   /* (the caller keeps 64 KiB free) */
      /* only empty blocks (bgzip's end marker, possibly in mid-file after cat): look further */
    }
}

static size_t readSome (MgSeqReader *r, char *dst, size_t room)
{
  if (r->bgzf) return bgzfRead (r, dst, room);
  if (room > ((size_t) 1 << 30)) room = (size_t) 1 << 30;
  if (r->gz)
    { int got = gzread (r->gz, dst, (unsigned) room);
      if (got < 0) { int e = 0; fprintf (stderr, "gzip read error: %s -- input ends here\n", gzerror (r->gz, &e)); }   /* the reference ends silently */
      return got > 0 ? (size_t) got : 0;
    }
  ssize_t got = read (r->fd, dst, room);
  if (got > 0) r->consumed += (size_t) got;
  return got > 0 ? (size_t) got : 0;
}
/* plain regular file: the pool reads disjoint slices with pread (the copy out of the page cache and
 * the first touch of the window are the cost, and both spread over the threads) */
typedef struct { int fd; char *dst; size_t fileOff, n, slice, next; size_t got; } ReadJob;
static void *readWorker (void *arg)
{
  ReadJob *j = (ReadJob *) arg;
  for (;;)
    { size_t at = __atomic_fetch_add (&j->next, j->slice, __ATOMIC_RELAXED);
      if (at >= j->n) break;
      size_t len = at + j->slice < j->n ? j->slice : j->n - at, done = 0;
      while (done < len)
        { ssize_t g = pread (j->fd, j->dst + at + done, len - done, (off_t) (j->fileOff + at + done));
          if (g <= 0) break;
          done += (size_t) g;
        }
      __atomic_fetch_add (&j->got, done, __ATOMIC_RELAXED);
    }
  return 0;
}
static size_t readParallel (MgSeqReader *r, char *dst, size_t n)
{
  ReadJob j; j.fd = r->fd; j.dst = dst; j.fileOff = r->consumed; j.n = n; j.slice = (size_t) 8 << 20; j.next = 0; j.got = 0;
  pthread_t th[32]; int started[32];
  int nt = r->nThreads; if ((size_t) nt > n / j.slice + 1) nt = (int) (n / j.slice + 1);
  for (int i = 1 ; i < nt ; ++i) started[i] = pthread_create (&th[i], 0, readWorker, &j) == 0;
  readWorker (&j);
  for (int i = 1 ; i < nt ; ++i) if (started[i]) pthread_join (th[i], 0);
  r->consumed += j.got;
  (void) lseek (r->fd, (off_t) r->consumed, SEEK_SET);
  return j.got;
}

static void closeInput (MgSeqReader *r) { if (r->gz) gzclose (r->gz); else close (r->fd); free (r->cbuf); r->cbuf = 0; }

/* startOff / startLine / startSeq: a reader that takes a plain regular file up from a record that starts at byte startOff, which
   is line startLine of the file and record startSeq + 1 (the device text parser hands a file over like this when it meets text
   it leaves to this parser: mg_textgpu.hip); 0 / 1 / 0 for a whole file */
static MgSeqReader *seqOpenAt (const char *filename, size_t startOff, U64 startLine, U64 startSeq)
{
  pthread_once (&convOnce, convInit);
  /* the reference reads everything through gzread (seqio.c:33-40), which passes plain text through;
     plain text is read directly here (zlib's pass-through copies at a fraction of memcpy speed) */
  int fd = strcmp (filename, "-") ? open (filename, O_RDONLY) : dup (0);
  if (fd < 0) return 0;
  unsigned char magic[2] = { 0, 0 };
  gzFile gz = 0;
  unsigned char head[64]; size_t pay = 0; int bgzf = 0;
  { ssize_t g = pread (fd, head, sizeof (head), 0); if (g >= 18 && bgzfBlockSize (head, (size_t) g, &pay)) bgzf = 1; }
  if (bgzf) ;                                                                    /* blocked gzip: inflated by the pool */
  else if (pread (fd, magic, 2, 0) != 2 || (magic[0] == 0x1f && magic[1] == 0x8b))   /* gzip, or not seekable: let zlib decide */
    { gz = gzdopen (fd, "r");
      if (!gz) { close (fd); return 0; }
      gzbuffer (gz, 1 << 20);
    }
  MgSeqReader *r = (MgSeqReader *) calloc (1, sizeof (MgSeqReader));
  pthread_mutex_lock (&bigMu); ++bigReaders; pthread_mutex_unlock (&bigMu);
  r->gz = gz; r->fd = fd; r->bgzf = bgzf;
  r->nThreads = threadCount ();
  r->cap = (size_t) 1 << 24;                                     /* seqio.c:36 */
  { struct stat st;                                              /* plain file: the window it will need is known */
    if (!gz && !bgzf && fstat (fd, &st) == 0 && S_ISREG (st.st_mode)) r->fileSize = (size_t) st.st_size;
  }
  if (startOff)
    { if (!r->fileSize || startOff >= r->fileSize || lseek (fd, (off_t) startOff, SEEK_SET) < 0) { closeInput (r); free (r); bigFlush (); return 0; }
      r->consumed = startOff;
    }
  r->buf = (char *) bigAlloc (r->cap);
  r->len = readSome (r, r->buf, r->cap);
  if (!r->len)
    { fprintf (stderr, "sequence file %s unreadable or empty\n", filename);   /* seqio.c:41-45 */
      closeInput (r); bigFree (r->buf, r->cap); free (r); bigFlush ();
      return 0;
    }
  if (r->buf[0] == '>') r->isFastq = 0;
  else if (r->buf[0] == '@') r->isFastq = 1;
  else
    { fprintf (stderr, "sequence file %s is neither FASTA nor FASTQ text\n", filename);
      closeInput (r); bigFree (r->buf, r->cap); free (r); bigFlush ();
      return 0;
    }
  r->line = startLine; r->nSeq = startSeq;
  r->nThreads = threadCount ();
  return r;
}

MgSeqReader *mgSeqOpen (const char *filename) { return seqOpenAt (filename, 0, 1, 0); }

void mgSeqClose (MgSeqReader *r)
{
  if (!r) return;
  closeInput (r); bigFree (r->buf, r->cap); free (r);
  bigFlush ();
  if (mgKnobs ()->parseTiming == 1)          /* dev: where the parser's time went, summed over the batches */
    { for (int i = 0 ; i < 8 ; ++i) if (gPhaseName[i]) { fprintf (stderr, "  [parse] %-8s %.3f s\n", gPhaseName[i], gPhase[i]); gPhase[i] = 0; } }
}

/* read until the window holds at least `want` unread bytes or the file ends (the buffer doubles as
 * it fills, seqio.c:197-205) */
static void refill (MgSeqReader *r, size_t want)
{
  if (r->eof) return;
  if (r->pos)
    { memmove (r->buf, r->buf + r->pos, r->len - r->pos);
      r->len -= r->pos; r->pos = 0;
    }
  if (r->fileSize && want > r->len + (r->fileSize - r->consumed) + 1)       /* never more than what is left of the file */
    want = r->len + (r->fileSize - r->consumed) + 1;
  if (r->fileSize && want > r->cap)                                         /* one allocation instead of doublings */
    { char *grown = (char *) bigAlloc (want);
      memcpy (grown, r->buf, r->len);
      bigFree (r->buf, r->cap);
      r->buf = grown; r->cap = want;
    }
  if (r->fileSize && want > r->len && want - r->len >= ((size_t) 32 << 20))
    { size_t left = r->fileSize - r->consumed, n = want - r->len < left ? want - r->len : left;
      size_t got = readParallel (r, r->buf + r->len, n);
      r->len += got;
      if (got < n || r->consumed >= r->fileSize) { r->eof = (r->consumed >= r->fileSize || got < n); }
    }
  while (r->len < want && !r->eof)
    { if (r->len == r->cap || (r->bgzf && r->cap - r->len < 65536))      /* a BGZF block inflates to up to 64 KiB */
        { size_t cap2 = r->cap * 2, lim = (size_t) 1 << 30;      /* straight to what is wanted (pages are only touched as they fill), 1 GiB at a time */
          if (want > cap2) cap2 = want < lim ? want : (cap2 > lim ? cap2 : lim);
          char *grown = (char *) bigAlloc (cap2);
          memcpy (grown, r->buf, r->len);
          bigFree (r->buf, r->cap);
          r->buf = grown; r->cap = cap2;
        }
      size_t room = r->cap - r->len;
      if (room > want - r->len && want - r->len >= ((size_t) 1 << 20)) room = want - r->len;
      size_t got = readSome (r, r->buf + r->len, room);
      if (!got) { r->eof = 1; break; }
      r->len += got;
    }
}

__attribute__ ((target_clones ("avx2", "default")))
static U64 countLines (const char *s, const char *e)
{ U64 n = 0; for ( ; s < e ; ++s) n += (*s == '\n'); return n; }

/* ---- records cut out of the window ---- */
typedef struct { size_t id, idLen, seq, seqEnd, end; } RawRec;      /* offsets into buf; [seq, seqEnd) is the text that holds the sequence */

/* FASTA record starts ('>' as the first byte of a line) of the unread window, found by the pool: each
 * thread looks through slices of the text with memchr; the slices' finds are then strung together in order. */
typedef struct { const char *buf; size_t lo, hi, slice, nSlices, next; size_t **found; size_t *nFound; } StartJob;
static void *startWorker (void *arg)
{
  StartJob *j = (StartJob *) arg;
  for (;;)
    { size_t k = __atomic_fetch_add (&j->next, 1, __ATOMIC_RELAXED);
      if (k >= j->nSlices) break;
      size_t a = j->lo + k * j->slice, b = a + j->slice < j->hi ? a + j->slice : j->hi;
      size_t cap = 64, n = 0; size_t *v = (size_t *) malloc (cap * sizeof (size_t));
      const char *p = j->buf + a, *e = j->buf + b;
      while (p < e)
        { const char *g = (const char *) memchr (p, '>', (size_t) (e - p));
          if (!g) break;
          if (g == j->buf + j->lo || g[-1] == '\n')
            { if (n == cap) { cap *= 2; v = (size_t *) realloc (v, cap * sizeof (size_t)); }
              v[n++] = (size_t) (g - j->buf);
            }
          p = g + 1;
        }
      j->found[k] = v; j->nFound[k] = n;
    }
  return 0;
}
static size_t findFastaStarts (MgSeqReader *r, size_t **startsOut)
{
  StartJob j; j.buf = r->buf; j.lo = r->pos; j.hi = r->len; j.slice = (size_t) 4 << 20; j.next = 0;
  j.nSlices = (j.hi - j.lo + j.slice - 1) / j.slice;
  j.found = (size_t **) calloc (j.nSlices + 1, sizeof (size_t *)); j.nFound = (size_t *) calloc (j.nSlices + 1, sizeof (size_t));
  pthread_t th[32]; int started[32];
  int nt = r->nThreads < (int) j.nSlices ? r->nThreads : (int) j.nSlices; if (nt < 1) nt = 1;
  for (int i = 1 ; i < nt ; ++i) started[i] = pthread_create (&th[i], 0, startWorker, &j) == 0;
  startWorker (&j);
  for (int i = 1 ; i < nt ; ++i) if (started[i]) pthread_join (th[i], 0);
  size_t n = 0;
  for (size_t k = 0 ; k < j.nSlices ; ++k) n += j.nFound[k];
  size_t *starts = (size_t *) malloc ((n + 1) * sizeof (size_t)), at = 0;
  for (size_t k = 0 ; k < j.nSlices ; ++k)
    { memcpy (starts + at, j.found[k], j.nFound[k] * sizeof (size_t)); at += j.nFound[k]; free (j.found[k]); }
  free (j.found); free (j.nFound);
  *startsOut = starts;
  return n;
}

/* the record that starts at `at`, the next record starting at `next` (0: no later start in the window).
 * Returns 1 and fills *rec when the record is whole; 0 when the window ends inside it. */
static int cutFasta (MgSeqReader *r, size_t at, size_t next, RawRec *rec, U64 line)
{
  const char *b = r->buf, *end = r->buf + (next ? next : r->len);
  if (b[at] != '>') dieLine ("no initial > for FASTA record line %llu", line);
  const char *nl = (const char *) memchr (b + at, '\n', (size_t) (end - (b + at)));
  if (!nl) return 0;
  const char *p = b + at + 1;
  while (p < nl && !isspace ((unsigned char) *p)) ++p;
  rec->id = at + 1; rec->idLen = (size_t) (p - (b + at + 1));
  rec->seq = (size_t) (nl + 1 - b);
  if (next) { rec->seqEnd = rec->end = next; return 1; }
  if (!r->eof) return 0;
  if (end[-1] != '\n') return 0;             /* unterminated last line: the caller reports it */
  if (nl + 1 == end) return 0;               /* the file ends with this header line: the reference reads on for the sequence, finds the end and
                                                reports "incomplete sequence record" without returning the record (seqio.c:213-217,314) */
  rec->seqEnd = rec->end = r->len;
  return 1;
}

static int cutFastq (MgSeqReader *r, size_t at, RawRec *rec, U64 line)
{
  const char *b = r->buf, *end = r->buf + r->len;
  if (b[at] != '@') dieLine ("no initial @ for FASTQ record line %llu", line);
  const char *l1 = (const char *) memchr (b + at, '\n', (size_t) (end - (b + at)));
  if (!l1) return 0;
  const char *l2 = (const char *) memchr (l1 + 1, '\n', (size_t) (end - (l1 + 1)));
  if (!l2) return 0;
  if (l2 + 1 >= end) return 0;
  if (l2[1] != '+') dieLine ("missing + FASTQ line %llu", line + 2);
  const char *l3 = (const char *) memchr (l2 + 1, '\n', (size_t) (end - (l2 + 1)));
  if (!l3) return 0;
  const char *l4 = (const char *) memchr (l3 + 1, '\n', (size_t) (end - (l3 + 1)));
  if (!l4) return 0;
  if ((l4 - (l3 + 1)) != (l2 - (l1 + 1))) dieLine ("qual not same length as seq line %llu", line + 3);
  const char *p = b + at + 1;
  while (p < l1 && !isspace ((unsigned char) *p)) ++p;
  rec->id = at + 1; rec->idLen = (size_t) (p - (b + at + 1));
  rec->seq = (size_t) (l1 + 1 - b); rec->seqEnd = (size_t) (l2 - b);
  rec->end = (size_t) (l4 + 1 - b);
  return 1;
}


/* ---- records cut by the pool ----
 * FASTQ: the pool lists the line ends of the window (slices of text, memchr), a record is then four
 * consecutive lines, so every record can be delimited and checked on its own.  FASTA: the record
 * starts are known (findFastaStarts), a record runs up to the next start.  What the pool cannot
 * decide (the last, possibly unfinished, record of the window) is left to cutFasta / cutFastq. */
static void runPool (int n, void *(*fn) (void *), void *arg)
{
  pthread_t th[32]; int started[32];
  if (n > 32) n = 32;
  for (int i = 1 ; i < n ; ++i) started[i] = pthread_create (&th[i], 0, fn, arg) == 0;
  fn (arg);
  for (int i = 1 ; i < n ; ++i) if (started[i]) pthread_join (th[i], 0);
}

typedef struct { const char *buf; size_t lo, hi, slice, nSlices, next; U32 **found; size_t *nFound, *prefix; } LineJob;
static void *lineWorker (void *arg)
{
  LineJob *j = (LineJob *) arg;
  for (;;)
    { size_t k = __atomic_fetch_add (&j->next, 1, __ATOMIC_RELAXED);
      if (k >= j->nSlices) break;
      size_t a = j->lo + k * j->slice, b = a + j->slice < j->hi ? a + j->slice : j->hi;
      size_t cap = (b - a) / 64 + 16, n = 0; U32 *v = (U32 *) malloc (cap * sizeof (U32));
      const char *p = j->buf + a, *e = j->buf + b;
      while (p < e)
        { const char *g = (const char *) memchr (p, '\n', (size_t) (e - p));
          if (!g) break;
          if (n == cap) { cap *= 2; v = (U32 *) realloc (v, cap * sizeof (U32)); }
          v[n++] = (U32) ((size_t) (g - j->buf) - j->lo);
          p = g + 1;
        }
      j->found[k] = v; j->nFound[k] = n;
    }
  return 0;
}
/* a walk over the listed line ends, in order */
typedef struct { const LineJob *j; size_t s, k; } LineCursor;
static void lineSeek (LineCursor *c, const LineJob *j, size_t line)
{
  size_t lo = 0, hi = j->nSlices;                   /* the slice with prefix[s] <= line < prefix[s+1] */
  while (hi - lo > 1) { size_t mid = (lo + hi) / 2; if (j->prefix[mid] <= line) lo = mid; else hi = mid; }
  c->j = j; c->s = lo; c->k = line - j->prefix[lo];
}
static inline size_t lineNext (LineCursor *c)
{
  while (c->k >= c->j->nFound[c->s]) { ++c->s; c->k = 0; }
  return c->j->lo + c->j->found[c->s][c->k++];
}

enum { CUT_OK = 0, CUT_NO_AT, CUT_NO_PLUS, CUT_QUAL };
typedef struct {
  const char *buf; const LineJob *lines; const size_t *starts;     /* FASTQ: lines; FASTA: starts */
  size_t at0; RawRec *recs; size_t nCand, grain, next;
  pthread_mutex_t mu; size_t errRec; int errKind;
} CutJob;
static void cutFail (CutJob *j, size_t i, int kind)
{
  pthread_mutex_lock (&j->mu);
  if (i < j->errRec) { j->errRec = i; j->errKind = kind; }
  pthread_mutex_unlock (&j->mu);
}
static void *cutWorker (void *arg)
{
  CutJob *j = (CutJob *) arg;
  const char *b = j->buf;
  for (;;)
    { size_t i0 = __atomic_fetch_add (&j->next, j->grain, __ATOMIC_RELAXED);
      if (i0 >= j->nCand) break;
      size_t i1 = i0 + j->grain < j->nCand ? i0 + j->grain : j->nCand;
      if (j->lines)
        { LineCursor c; size_t at = j->at0;
          if (i0) { lineSeek (&c, j->lines, 4 * i0 - 1); at = lineNext (&c) + 1; } else lineSeek (&c, j->lines, 0);
          for (size_t i = i0 ; i < i1 ; ++i)
            { const size_t l1 = lineNext (&c), l2 = lineNext (&c), l3 = lineNext (&c), l4 = lineNext (&c);
              RawRec *rec = &j->recs[i];
              if (b[at] != '@') cutFail (j, i, CUT_NO_AT);
              else if (b[l2 + 1] != '+') cutFail (j, i, CUT_NO_PLUS);
              else if (l4 - (l3 + 1) != l2 - (l1 + 1)) cutFail (j, i, CUT_QUAL);
              const char *p = b + at + 1;
              while (p < b + l1 && !isspace ((unsigned char) *p)) ++p;
              rec->id = at + 1; rec->idLen = (size_t) (p - (b + at + 1));
              rec->seq = l1 + 1; rec->seqEnd = l2; rec->end = l4 + 1;
              at = l4 + 1;
            }
        }
      else
        for (size_t i = i0 ; i < i1 ; ++i)
          { const size_t at = j->starts[i], next = j->starts[i + 1];
            const char *nl = (const char *) memchr (b + at, '\n', next - at);   /* there is one: the next start follows a line end */
            const char *p = b + at + 1;
            while (p < nl && !isspace ((unsigned char) *p)) ++p;
            RawRec *rec = &j->recs[i];
            rec->id = at + 1; rec->idLen = (size_t) (p - (b + at + 1));
            rec->seq = (size_t) (nl + 1 - b); rec->seqEnd = rec->end = next;
          }
    }
  return 0;
}

/* Whole records from position *at of the window on, cut by the pool and appended to *recs, until
 * *rawSeq reaches maxBases or the window holds no further whole record that the pool can delimit.
 * Returns how many were appended; *at, *rawSeq, *line move as the serial loop would move them.
 * FASTQ looks at a bounded stretch of text per call (about what maxBases needs), so call it again
 * while it makes progress. */
static size_t cutByPool (MgSeqReader *r, int64_t maxBases, const size_t *starts, size_t nStarts,
                         RawRec **recs, size_t *nRec, size_t *capRec, size_t *at, size_t *rawSeq, U64 *line)
{
  CutJob cj; memset (&cj, 0, sizeof (cj));
  LineJob lj; memset (&lj, 0, sizeof (lj));
  cj.buf = r->buf; cj.at0 = *at; cj.errRec = (size_t) -1;
  size_t si = 0;
  if (*at >= r->len || (int64_t) *rawSeq >= maxBases) return 0;
  if (r->isFastq)
    { const size_t need = (size_t) (maxBases - (int64_t) *rawSeq);
      size_t span = need < ((size_t) 1 << 29) ? 4 * need + 65536 : (size_t) 0xfff00000u;   /* line ends are kept as 32-bit offsets */
      lj.buf = r->buf; lj.lo = *at; lj.hi = r->len - *at > span ? *at + span : r->len; lj.slice = (size_t) 4 << 20;
      lj.nSlices = (lj.hi - lj.lo + lj.slice - 1) / lj.slice;
      lj.found = (U32 **) calloc (lj.nSlices, sizeof (U32 *)); lj.nFound = (size_t *) calloc (lj.nSlices, sizeof (size_t));
      lj.prefix = (size_t *) calloc (lj.nSlices + 1, sizeof (size_t));
      runPool (r->nThreads < (int) lj.nSlices ? r->nThreads : (int) lj.nSlices, lineWorker, &lj);
      for (size_t k = 0 ; k < lj.nSlices ; ++k) lj.prefix[k + 1] = lj.prefix[k] + lj.nFound[k];
      cj.lines = &lj; cj.nCand = lj.prefix[lj.nSlices] / 4;
    }
  else
    { while (si < nStarts && starts[si] < *at) ++si;
      if (si + 1 >= nStarts || starts[si] != *at) return 0;
      cj.starts = starts + si; cj.nCand = nStarts - 1 - si;
    }
  size_t taken = 0;
  if (cj.nCand)
    { RawRec *cand = (RawRec *) malloc (cj.nCand * sizeof (RawRec));
      cj.recs = cand; cj.grain = cj.nCand / ((size_t) r->nThreads * 16) + 1;
      pthread_mutex_init (&cj.mu, 0);
      runPool (r->nThreads, cutWorker, &cj);
      pthread_mutex_destroy (&cj.mu);
      while (taken < cj.nCand && (int64_t) *rawSeq < maxBases)
        { if (taken == cj.errRec)
            { const U64 l = *line + 4 * (U64) taken;
              if (cj.errKind == CUT_NO_AT) dieLine ("no initial @ for FASTQ record line %llu", l);
              if (cj.errKind == CUT_NO_PLUS) dieLine ("missing + FASTQ line %llu", l + 2);
              dieLine ("qual not same length as seq line %llu", l + 3);
            }
          *rawSeq += cand[taken].seqEnd - cand[taken].seq;
          ++taken;
        }
      if (taken)
        { *at = cand[taken - 1].end;
          if (r->isFastq) *line += 4 * (U64) taken;
          if (!*recs) { *recs = cand; *capRec = cj.nCand; *nRec = taken; cand = 0; }       /* the usual case: one call per batch */
          else
            { if (*nRec + taken > *capRec) { *capRec = (*nRec + taken) * 2; *recs = (RawRec *) realloc (*recs, *capRec * sizeof (RawRec)); }
              memcpy (*recs + *nRec, cand, taken * sizeof (RawRec));
              *nRec += taken;
            }
        }
      free (cand);
    }
  if (lj.found) { for (size_t k = 0 ; k < lj.nSlices ; ++k) free (lj.found[k]); free (lj.found); free (lj.nFound); free (lj.prefix); }
  return taken;
}

/* ---- parallel conversion ----
 * Two passes over the raw text, both by the pool: count what each unit keeps (FASTA only; a FASTQ
 * line keeps every byte), then, the destinations being known, convert straight into the batch. */
typedef struct { size_t from, to; int rec; size_t outLen, lines; } Unit;  /* raw range of one record */
typedef struct {
  const char *raw; Unit *units; size_t nUnits, grain; int isFastq;
  size_t next;
  char *dst; const size_t *unitDst;                               /* second pass */
  const RawRec *recs; char **names;                               /* second pass: the unit that opens a record copies its id */
  int phase;
} Job;

/* ---- the byte-crunching inner loops, with AVX2 where the CPU has it ----
 * (the plain C loops below do not get vectorised by the compiler: 0.9 GB/s per core; the explicit ones run at 7) */
#include <immintrin.h>
static int haveAvx2 (void)                       /* MODGPU_NO_AVX2=1 (tests): the portable loops */
{
  return mgKnobs ()->noAvx2 == 1 ? 0 : (__builtin_cpu_supports ("avx2") ? 1 : 0);
}
static size_t countKept (const unsigned char *s, const unsigned char *e)      /* bytes convTable keeps */
{
  size_t n = 0;
  for ( ; s < e ; ++s)
    { unsigned char c = (unsigned char) (*s | 0x20);
      n += (c == 'a') | (c == 'c') | (c == 'g') | (c == 't') | (c == 'n');
    }
  return n;
}
static void countScalar (const unsigned char *s, const unsigned char *e, size_t *kept, size_t *lines)
{
  size_t n = 0, l = 0;
  for ( ; s < e ; ++s)
    { unsigned char c = (unsigned char) (*s | 0x20);
      n += (c == 'a') | (c == 'c') | (c == 'g') | (c == 't') | (c == 'n');
      l += (*s == '\n');
    }
  *kept = n; *lines = l;
}
/* 0xff in the bytes that are A C G T N in either case */
#define KEPT_MASK(x) \
  ({ const __m256i c_ = _mm256_or_si256 ((x), _mm256_set1_epi8 (0x20)); \
     _mm256_or_si256 (_mm256_or_si256 (_mm256_cmpeq_epi8 (c_, _mm256_set1_epi8 ('a')), _mm256_cmpeq_epi8 (c_, _mm256_set1_epi8 ('c'))), \
                      _mm256_or_si256 (_mm256_or_si256 (_mm256_cmpeq_epi8 (c_, _mm256_set1_epi8 ('g')), _mm256_cmpeq_epi8 (c_, _mm256_set1_epi8 ('t'))), \
                                       _mm256_cmpeq_epi8 (c_, _mm256_set1_epi8 ('n')))); })
__attribute__ ((target ("avx2")))
static void countAvx2 (const unsigned char *s, const unsigned char *e, size_t *kept, size_t *lines)
{
  size_t n = 0, l = 0;
  const __m256i zero = _mm256_setzero_si256 (), NL = _mm256_set1_epi8 ('\n');
  while (s + 32 <= e)
    { __m256i kacc = zero, lacc = zero;                        /* byte-wide partial sums: at most 255 steps before they are folded */
      size_t steps = (size_t) (e - s) / 32; if (steps > 255) steps = 255;
      for (size_t i = 0 ; i < steps ; ++i, s += 32)
        { const __m256i x = _mm256_loadu_si256 ((const __m256i *) s);
          kacc = _mm256_sub_epi8 (kacc, KEPT_MASK (x));         /* a match is -1 */
          lacc = _mm256_sub_epi8 (lacc, _mm256_cmpeq_epi8 (x, NL));
        }
      const __m256i ks = _mm256_sad_epu8 (kacc, zero), ls = _mm256_sad_epu8 (lacc, zero);
      n += (size_t) _mm256_extract_epi64 (ks, 0) + (size_t) _mm256_extract_epi64 (ks, 1) + (size_t) _mm256_extract_epi64 (ks, 2) + (size_t) _mm256_extract_epi64 (ks, 3);
      l += (size_t) _mm256_extract_epi64 (ls, 0) + (size_t) _mm256_extract_epi64 (ls, 1) + (size_t) _mm256_extract_epi64 (ls, 2) + (size_t) _mm256_extract_epi64 (ls, 3);
    }
  size_t n2, l2; countScalar (s, e, &n2, &l2);
  *kept = n + n2; *lines = l + l2;
}
/* what a unit keeps and how many lines it holds, in ONE pass over its text */
static void countKeptAndLines (const unsigned char *s, const unsigned char *e, size_t *kept, size_t *lines)
{
  if (haveAvx2 ()) countAvx2 (s, e, kept, lines); else countScalar (s, e, kept, lines);
}

/* A stretch of text without a line end -> bases, optimistically: ((c >> 1) ^ (c >> 2)) & 3 sends A C G T to 0 1 2 3 and
 * N to 0.  Returns 1 when every byte was A C G T N (either case) and t[0..len) is right; 0 as soon as another byte is
 * met (the caller redoes the stretch through the table: t may hold rubbish).  keepOthers (FASTQ): other bytes become
 * (char) -2 as the table has them and the stretch always succeeds. */
__attribute__ ((target ("avx2")))
static int convAvx2 (const unsigned char *s, char *t, size_t len, int keepOthers)
{
  const __m256i three = _mm256_set1_epi8 (3), other = _mm256_set1_epi8 ((char) -2);
  size_t i = 0;
  for ( ; i + 32 <= len ; i += 32)
    { const __m256i x = _mm256_loadu_si256 ((const __m256i *) (s + i)), ok = KEPT_MASK (x);
      __m256i code = _mm256_and_si256 (_mm256_xor_si256 (_mm256_srli_epi16 (x, 1), _mm256_srli_epi16 (x, 2)), three);   /* bits 1..3 of a byte stay in it */
      if (_mm256_movemask_epi8 (ok) != -1)
        { if (!keepOthers) return 0;
          code = _mm256_blendv_epi8 (other, code, ok);
        }
      _mm256_storeu_si256 ((__m256i *) (t + i), code);
    }
  for ( ; i < len ; ++i)
    { const unsigned char c = (unsigned char) (s[i] | 0x20);
      if ((c == 'a') | (c == 'c') | (c == 'g') | (c == 't') | (c == 'n')) t[i] = (char) (((s[i] >> 1) ^ (s[i] >> 2)) & 3);
      else if (keepOthers) t[i] = (char) -2;
      else return 0;
    }
  return 1;
}
static int convScalar (const unsigned char *s, char *t, size_t len, int keepOthers)
{
  if (!keepOthers && countKept (s, s + len) != len) return 0;
  for (size_t i = 0 ; i < len ; ++i)
    { const unsigned char c = (unsigned char) (s[i] | 0x20);
      t[i] = ((c == 'a') | (c == 'c') | (c == 'g') | (c == 't') | (c == 'n')) ? (char) (((s[i] >> 1) ^ (s[i] >> 2)) & 3) : (char) -2;
    }
  return 1;
}
static int convStretch (const unsigned char *s, char *t, size_t len, int keepOthers)
{
  return haveAvx2 () ? convAvx2 (s, t, len, keepOthers) : convScalar (s, t, len, keepOthers);
}

static void *worker (void *arg)
{
  Job *j = (Job *) arg;
  for (;;)
    { size_t u0 = __atomic_fetch_add (&j->next, j->grain, __ATOMIC_RELAXED);
      if (u0 >= j->nUnits) break;
      size_t u1 = u0 + j->grain < j->nUnits ? u0 + j->grain : j->nUnits;
      for (size_t u = u0 ; u < u1 ; ++u)
        { Unit *un = &j->units[u];
          const unsigned char *s = (const unsigned char *) j->raw + un->from, *e = (const unsigned char *) j->raw + un->to;
          if (j->phase == 0)
            { if (j->isFastq) { un->outLen = (size_t) (e - s); un->lines = 0; }
              else countKeptAndLines (s, e, &un->outLen, &un->lines);
            }
          else
            { char *t = j->dst + j->unitDst[u];
              const RawRec *rc = &j->recs[un->rec];
              if (un->from == rc->seq) { memcpy (j->names[un->rec], j->raw + rc->id, rc->idLen); j->names[un->rec][rc->idLen] = 0; }
              /* line by line: a line made of A C G T N only (either case) converts with arithmetic on whole vectors
                 -- ((c >> 1) ^ (c >> 2)) & 3 sends A C G T to 0 1 2 3 and N to 0 -- anything else goes through the table */
              while (s < e)
                { const unsigned char *nl = (const unsigned char *) memchr (s, '\n', (size_t) (e - s));
                  const unsigned char *le = nl ? nl : e;
                  const size_t len = (size_t) (le - s);
                  if (convStretch (s, t, len, j->isFastq)) t += len;
                  else for ( ; s < le ; ++s) { signed char c = convTable[*s]; if (c >= 0) *t++ = (char) c; }   /* FASTA with other bytes in the line: no store for a dropped byte */
                  s = le;
                  if (nl) { if (j->isFastq) *t++ = (char) convTable['\n']; ++s; }                         /* (a FASTQ unit never holds a newline) */
                }
            }
        }
    }
  return 0;
}

static void runJob (Job *j, int nThreads)
{
  j->next = 0;
  j->grain = j->nUnits / ((size_t) nThreads * 16) + 1;
  if (nThreads > 1 && j->nUnits > 1)
    { pthread_t th[32]; int started[32];
      int n = nThreads < (int) j->nUnits ? nThreads : (int) j->nUnits;
      for (int i = 1 ; i < n ; ++i) started[i] = pthread_create (&th[i], 0, worker, j) == 0;
      worker (j);
      for (int i = 1 ; i < n ; ++i) if (started[i]) pthread_join (th[i], 0);
    }
  else worker (j);
}

/* ---- per-record bookkeeping by the pool ---- */
typedef struct { size_t units, nameBytes, bases, lines; } BookRun;       /* sums of a run of records, then their prefixes */
typedef struct {
  const RawRec *recs; size_t nRec, nRuns, next; int isFastq, phase;
  BookRun *run; Unit *units; char **names; char *nameArena; int64_t *offsets; size_t *unitDst;
} Book;
/* units a record of `len` raw sequence bytes is cut into: ceil (len / UNIT_BYTES), and one (empty) unit for an empty
 * record -- exactly what the do/while of phase 1 creates (an exact multiple of UNIT_BYTES must not count one more) */
static inline size_t unitsOf (size_t len) { return len ? (len + UNIT_BYTES - 1) / UNIT_BYTES : 1; }
static void *bookWorker (void *arg)
{
  Book *b = (Book *) arg;
  for (;;)
    { const size_t c = __atomic_fetch_add (&b->next, 1, __ATOMIC_RELAXED);
      if (c >= b->nRuns) break;
      const size_t i0 = b->nRec * c / b->nRuns, i1 = b->nRec * (c + 1) / b->nRuns;
      BookRun *rn = &b->run[c];
      if (b->phase == 0)                                          /* units and id bytes of the run */
        { size_t nu = 0, nb = 0;
          for (size_t i = i0 ; i < i1 ; ++i) { nu += unitsOf (b->recs[i].seqEnd - b->recs[i].seq); nb += b->recs[i].idLen + 1; }
          rn->units = nu; rn->nameBytes = nb;
        }
      else if (b->phase == 1)                                     /* the run's units and id slots */
        { size_t u = rn->units; char *nm = b->nameArena + rn->nameBytes;
          for (size_t i = i0 ; i < i1 ; ++i)
            { size_t s = b->recs[i].seq;
              do
                { size_t e = s + UNIT_BYTES < b->recs[i].seqEnd ? s + UNIT_BYTES : b->recs[i].seqEnd;
                  Unit *un = &b->units[u++];
                  un->from = s; un->to = e; un->rec = (int) i; un->outLen = b->isFastq ? e - s : 0; un->lines = 0;
                  s = e;
                }
              while (s < b->recs[i].seqEnd);
              b->names[i] = nm; nm += b->recs[i].idLen + 1;
            }
        }
      else if (b->phase == 2)                                     /* what the run's units keep */
        { size_t nb = 0, nl = 0;
          for (size_t u = rn->units ; u < rn[1].units ; ++u) { nb += b->units[u].outLen; nl += b->units[u].lines; }
          rn->bases = nb; rn->lines = nl;
        }
      else                                                        /* where the run's records and units go */
        { size_t u = rn->units, total = rn->bases;
          for (size_t i = i0 ; i < i1 ; ++i)
            { b->offsets[i] = (int64_t) total;
              while (u < rn[1].units && b->units[u].rec == (int) i) { b->unitDst[u] = total; total += b->units[u].outLen; ++u; }
            }
        }
    }
  return 0;
}

/* Next batch of whole records: at least one, and no more once `maxBases` sequence characters have
 * been taken.  Returns the number of records (0 at the end of the file). */
int mgSeqNextBatch (MgSeqReader *r, int64_t maxBases, MgSeqBatch *out)
{
  memset (out, 0, sizeof (*out));
  out->isFastq = r->isFastq;
  if (r->finished) return 0;
  if (maxBases < 1) maxBases = 1;
  if (maxBases > ((int64_t) 1 << 60)) maxBases = (int64_t) 1 << 60;
  size_t want = (size_t) maxBases + (size_t) maxBases / 4 + ((size_t) 1 << 24);
  double tLast = nowS ();
  if (r->len - r->pos < want) refill (r, want);
  TIMING (0, "read");

  RawRec *recs = 0; size_t nRec = 0, capRec = 0;
  size_t at = r->pos; U64 line = r->line; size_t rawSeq = 0;
  size_t *starts = 0, nStarts = 0, si = 0;                       /* FASTA: where records start, found by the pool */
  if (!r->isFastq) nStarts = findFastaStarts (r, &starts);
  while (cutByPool (r, maxBases, starts, nStarts, &recs, &nRec, &capRec, &at, &rawSeq, &line)) ;
  while (at < r->len && (int64_t) rawSeq < maxBases)
    { RawRec rec;
      int ok;
      if (r->isFastq) ok = cutFastq (r, at, &rec, line);
      else
        { while (si < nStarts && starts[si] < at) ++si;
          const size_t next = (si < nStarts && starts[si] == at && si + 1 < nStarts) ? starts[si + 1] : 0;
          ok = cutFasta (r, at, next, &rec, line);
        }
      if (!ok)
        { if (!r->eof)
            { if (nRec) break;                                   /* hand over what is whole; the rest next time */
              size_t have = r->len - r->pos;
              refill (r, have * 2 > want ? have * 2 : want);     /* one record larger than the window */
              at = r->pos;
              if (!r->isFastq) { free (starts); nStarts = findFastaStarts (r, &starts); si = 0; }
              continue;
            }
          /* FASTA lines of this batch are only counted by the pool below: count here what was cut so far */
          fprintf (stderr, "incomplete sequence record line %llu\n",
                   (unsigned long long) (r->line + countLines (r->buf + r->pos, r->buf + r->len)));
          r->finished = 1;
          break;
        }
      if (nRec == capRec) { capRec = capRec ? capRec * 2 : 1024; recs = (RawRec *) realloc (recs, capRec * sizeof (RawRec)); }
      recs[nRec++] = rec;
      rawSeq += rec.seqEnd - rec.seq;
      if (r->isFastq) line += 4;
      at = rec.end;
    }
  free (starts);
  if (at >= r->len && r->eof) r->finished = 1;
  if (!nRec) { free (recs); r->pos = at; r->line = line; return 0; }
  TIMING (1, "cut");

  /* units: raw ranges of at most UNIT_BYTES, never across records */
  /* per-record bookkeeping, by the pool over contiguous runs of records: sums per run, a serial prefix over the
     (few) runs, then every run fills its own part -- units and id slots first, output offsets after the count */
  Book bk; memset (&bk, 0, sizeof (bk));
  bk.recs = recs; bk.nRec = nRec; bk.isFastq = r->isFastq;
  bk.nRuns = (size_t) r->nThreads * 4 < nRec ? (size_t) r->nThreads * 4 : (nRec < 1 ? 1 : nRec);
  if (nRec < 4096) bk.nRuns = 1;
  bk.run = (BookRun *) calloc (bk.nRuns + 1, sizeof (BookRun));
  const int bookThreads = bk.nRuns > 1 ? r->nThreads : 1;
  bk.phase = 0; bk.next = 0; runPool (bookThreads, bookWorker, &bk);
  size_t nUnits = 0, nameBytes = 0;
  for (size_t c = 0 ; c < bk.nRuns ; ++c)
    { size_t nu = bk.run[c].units, nb = bk.run[c].nameBytes;
      bk.run[c].units = nUnits; bk.run[c].nameBytes = nameBytes; nUnits += nu; nameBytes += nb;
    }
  bk.run[bk.nRuns].units = nUnits;
  Unit *units = (Unit *) malloc ((nUnits ? nUnits : 1) * sizeof (Unit));
  out->nSeq = (int) nRec;
  out->offsets = (int64_t *) malloc ((nRec + 1) * sizeof (int64_t));
  out->names = (char **) malloc (nRec * sizeof (char *));
  size_t *unitDst = (size_t *) malloc ((nUnits ? nUnits : 1) * sizeof (size_t));
  bk.units = units; bk.names = out->names; bk.offsets = out->offsets; bk.unitDst = unitDst;
  bk.nameArena = (char *) malloc (nameBytes ? nameBytes : 1);    /* one block for all ids (names[0] is its start), filled by the pool with the bases */
  bk.phase = 1; bk.next = 0; runPool (bookThreads, bookWorker, &bk);
  Job j; memset (&j, 0, sizeof (j));
  j.raw = r->buf; j.units = units; j.nUnits = nUnits; j.isFastq = r->isFastq;
  j.phase = 0;
  if (!r->isFastq) runJob (&j, r->nThreads);                     /* a FASTQ line keeps every byte: nothing to count */
  TIMING (2, "count");

  bk.phase = 2; bk.next = 0; runPool (bookThreads, bookWorker, &bk);
  size_t total = 0;
  for (size_t c = 0 ; c < bk.nRuns ; ++c)
    { size_t nb = bk.run[c].bases; bk.run[c].bases = total; total += nb; line += bk.run[c].lines; }
  if (!r->isFastq) line += nRec;                                 /* the header lines */
  bk.phase = 3; bk.next = 0; runPool (bookThreads, bookWorker, &bk);
  free (bk.run);
  out->offsets[nRec] = (int64_t) total;
  out->total = (int64_t) total;
  out->basesCap = (int64_t) (total ? total : 1);
  TIMING (4, "offsets");
  out->bases = (char *) bigAlloc ((size_t) out->basesCap);
  TIMING (5, "alloc");
  j.phase = 1; j.dst = out->bases; j.unitDst = unitDst; j.recs = recs; j.names = out->names;
  runJob (&j, r->nThreads);
  TIMING (3, "convert");
  free (unitDst); free (units); free (recs);

  r->pos = at; r->line = line; r->nSeq += nRec;
  return (int) nRec;
}

void mgSeqBatchFree (MgSeqBatch *b)
{
  if (!b) return;
  if (b->names && b->nSeq > 0) free (b->names[0]);               /* the ids share one block */
  free (b->names); free (b->offsets); bigFree (b->bases, (size_t) b->basesCap);
  memset (b, 0, sizeof (*b));
}

/* ---- the callers' file loops ---- */

typedef struct { MgSeqReader *r; int64_t maxBases; MgSeqBatch batch; int n; } Prefetch;
static void *prefetchMain (void *arg) { Prefetch *p = (Prefetch *) arg; p->n = mgSeqNextBatch (p->r, p->maxBases, &p->batch); return 0; }

static int64_t batchBases (void)
{
  const long fk = mgKnobs ()->fileBatchMbp;
  long mbp = fk != MG_KNOB_UNSET ? fk : 512;          /* parse of the next batch overlaps the GPU work on this one */
  if (mbp < 1) mbp = 1;
  return (int64_t) mbp * 1000000;
}

/* every batch of the file through fn, the next batch being parsed while fn works on this one */
static int forEachBatchFrom (const char *filename, size_t startOff, U64 startLine, U64 startSeq, int (*fn) (MgSeqBatch *, void *), void *ctx)
{
  MgSeqReader *r = seqOpenAt (filename, startOff, startLine, startSeq);
  if (!r) return -1;
  Prefetch p; memset (&p, 0, sizeof (p)); p.r = r; p.maxBases = batchBases ();
  prefetchMain (&p);
  int rc = 0;
  while (p.n > 0 && !rc)
    { MgSeqBatch cur = p.batch;
      pthread_t th;
      memset (&p.batch, 0, sizeof (p.batch)); p.n = 0;
      int threaded = pthread_create (&th, 0, prefetchMain, &p) == 0;
      double tLast = nowS ();
      rc = fn (&cur, ctx);
      TIMING (6, "consume");
      if (threaded) pthread_join (th, 0); else prefetchMain (&p);
      TIMING (7, "wait");
      mgSeqBatchFree (&cur);
    }
  if (p.n > 0) mgSeqBatchFree (&p.batch);
  mgSeqClose (r);
  return rc;
}

static int forEachBatch (const char *filename, int (*fn) (MgSeqBatch *, void *), void *ctx)
{ return forEachBatchFrom (filename, 0, 1, 0, fn, ctx); }
/* for the callers in other files (mg_readset.c): the host parser from byte startOff on (0, line 1, no sequence before: the whole file) */
int mgSeqForEachBatchFrom (const char *filename, size_t startOff, U64 startLine, U64 startSeq, int (*fn) (MgSeqBatch *, void *), void *ctx)
{ return forEachBatchFrom (filename, startOff, startLine, startSeq, fn, ctx); }

typedef struct { Modset *ms; U64 nSeq, totLen, totHash; } AddCtx;
static int addBatch (MgSeqBatch *b, void *v)
{
  AddCtx *c = (AddCtx *) v;
  int64_t h = mgAddSequenceBatch (c->ms, b->bases, b->offsets, b->nSeq);
  if (h < 0) return -1;
  c->nSeq += (U64) b->nSeq; c->totLen += (U64) b->total; c->totHash += (U64) h;
  return 0;
}

int mgAddSequenceFile (Modset *ms, const char *filename, FILE *out)            /* modutils.c:33-51 */
{
  AddCtx c; memset (&c, 0, sizeof (c)); c.ms = ms;
  /* plain FASTA / FASTQ text: the device parses it (mg_textgpu.hip), the host only moves the bytes; everything else -- gzip, a
     last line without its newline -- through the host parser below */
  U64 resumeOff = 0, resumeLine = 1;
  int rc = mgAddSequenceFileDevice (ms, filename, &c.nSeq, &c.totLen, &c.totHash, &resumeOff, &resumeLine);
  if (rc == -1) return -1;
  if (rc == -2) rc = forEachBatch (filename, addBatch, &c);
  else if (rc == -3) rc = forEachBatchFrom (filename, (size_t) resumeOff, resumeLine, c.nSeq, addBatch, &c);   /* the device parser met FASTQ text it leaves to this one (a record that breaks the rules, an unfinished one): from the first record it has not added */
  if (rc) return rc;
  fprintf (out, "added %llu sequences total length %llu total hashes %llu, new max %u\n",
           (unsigned long long) c.nSeq, (unsigned long long) c.totLen, (unsigned long long) c.totHash, ms->max);
  return 0;
}

/* the ids of a device batch as the pointer array the callers take */
static const char **idPointers (const char *idBytes, const U64 *idOff, U32 n)
{
  const char **p = (const char **) malloc (((size_t) n + 1) * sizeof (char *));
  for (U32 i = 0 ; i < n ; ++i) p[i] = idBytes + idOff[i];
  return p;
}

typedef struct { MgReference *ref; bool isAdd; } RefDevCtx;
static int refDeviceBatch (void *v, const U32 *dPacked, U64 total, const U64 *dOff, U32 nReads, const char *idBytes, const U64 *idOff, void *stream)
{
  RefDevCtx *c = (RefDevCtx *) v; (void) stream;
  const char **names = idPointers (idBytes, idOff, nReads);
  int rc = mgReferenceAddDevice (c->ref, dPacked, total, dOff, (int) nReads, names, c->isAdd);
  free (names);
  return rc;
}

/* modmap reads the whole reference before it classifies and packs (modmap.c:93-134), and its report
 * lines are per file.  Plain FASTA text is parsed on the device (mg_textgpu.hip), batch by batch -- first-occurrence
 * indices do not depend on where a stream is cut -- everything else by the host parser as one batch holding every record */
int mgReferenceFastaRead (MgReference *ref, const char *filename, bool isAdd, FILE *out)
{
  { RefDevCtx c; c.ref = ref; c.isAdd = isAdd;
    U64 nSeq = 0, totLen = 0, resumeOff = 0, resumeLine = 1;
    int first = -1;
    { FILE *f = fopen (filename, "rb"); if (f) { first = fgetc (f); fclose (f); } }
    if (first == '>')                                     /* (a FASTQ reference goes the host way: the hand-over in the middle of a file is not worth having here) */
      { const int rc = mgTextForEachBatchDevice (filename, refDeviceBatch, &c, 0, 0, &nSeq, &totLen, &resumeOff, &resumeLine);
        if (rc == -1) return -1;
        if (rc == 0) { mgReferenceFinish (ref, totLen, isAdd, out); return 0; }
      }
  }
  MgSeqReader *r = mgSeqOpen (filename);
  if (!r) { fprintf (stderr, "FATAL ERROR: failed to read reference sequence file %s\n", filename); exit (-1); }   /* modmap.c:99 */
  MgSeqBatch b;
  int n = mgSeqNextBatch (r, INT64_MAX, &b);
  int rc = mgReferenceRead (ref, b.bases, b.offsets, n, (const char **) b.names, isAdd, out);
  mgSeqBatchFree (&b);
  mgSeqClose (r);
  return rc;
}

typedef struct { MgReference *ref; FILE *out; } QueryCtx;
static int queryBatch (MgSeqBatch *b, void *v)
{ QueryCtx *c = (QueryCtx *) v; return mgQueryProcess (c->ref, b->bases, b->offsets, b->nSeq, (const char **) b->names, c->out); }

static int queryDeviceBatch (void *v, const U32 *dPacked, U64 total, const U64 *dOff, U32 nReads, const char *idBytes, const U64 *idOff, void *stream)
{ (void) stream; return mgQueryPipePush ((MgQueryPipe *) v, dPacked, total, dOff, (int) nReads, idBytes, idOff); }

/* short reads: a query batch per window of the file (128 MiB of text; a batch is handed on at the first window's end at which it holds
   this many bases and records): the lines of one batch are formatted and written while the next one is read, parsed and queried.
   Long reads (few lines): batches of the default size, 1 Gbp -- the chaining of a batch (mg_chain.hip) takes as long as the serial
   walk of its longest read, whatever else it holds: 38 batches of 134 Mbp spent 17 ms there per 5 Gbp */
#define MG_QUERY_FILE_BATCH 32000000ull
#define MG_QUERY_FILE_BATCH_RECS 200000ull

int mgQueryFile (MgReference *ref, const char *filename, FILE *out)            /* modmap.c:188-196 */
{
  QueryCtx c; c.ref = ref; c.out = out;
  /* plain FASTA / FASTQ text: parsed on the device, the batches stay there (no 1-byte-per-base upload); gzip, a last line without
     its newline, FASTQ that breaks a rule: the host parser (from the first record the device parser has not handed on) */
  U64 nSeq = 0, totLen = 0, resumeOff = 0, resumeLine = 1;
  MgQueryPipe *pipe = mgQueryPipeOpen (ref, out);
  int rc = mgTextForEachBatchDevice (filename, queryDeviceBatch, pipe, MG_QUERY_FILE_BATCH, MG_QUERY_FILE_BATCH_RECS, &nSeq, &totLen, &resumeOff, &resumeLine);
  mgQueryPipeClose (pipe);                                 /* every line of the device parser's batches is out */
  if (rc == 0 || rc == -1) return rc;
  if (rc == -3) return forEachBatchFrom (filename, (size_t) resumeOff, resumeLine, nSeq, queryBatch, &c);
  rc = forEachBatch (filename, queryBatch, &c);
  if (rc == -1 && access (filename, R_OK)) { fprintf (stderr, "FATAL ERROR: failed to read query sequence file %s\n", filename); exit (-1); }   /* modmap.c:196 */
  return rc;
}
