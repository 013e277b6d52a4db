/* mg_pgzip.c — the gzip stream of the reference's files, written by a team of threads.
 *
 * The reference writes .mod / .ref / .readset through fzopen (utils.c:107-127): ONE gzwrite stream, which is one core of zlib
 * behind a set that the device built in milliseconds (a config-2 .mod is 104 + 4 * 2^30 + 11 * 1.03e8 bytes = 5.4 GB).  The gzip
 * format allows a file to be several members one after the other, and gzread -- hence the reference's own fzopen "r", and gunzip --
 * decodes them as one stream.  So: the bytes handed to the FILE * are cut into members of MG_PGZ_MEMBER uncompressed bytes, every
 * member is deflated by itself (zlib, gzip wrapper, the level gzopen "w" uses), the members are written in order.  A large fwrite
 * (modsetWrite's index[] is one fwrite of 4 GiB) is compressed straight out of the caller's array, nothing is copied; small
 * writes collect in a member-sized buffer.  What comes out decompresses to exactly what the single stream would have held.
 */
#define _GNU_SOURCE             /* fopencookie */
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>
#include "modgpu.h"
#include "mg_internal.h"

#define MG_PGZ_MEMBER ((size_t) 16 << 20)       /* uncompressed bytes per member */
#define MG_PGZ_MAXT   32

typedef struct
{ int fd; int level; int T; int err;
  unsigned char *pend; size_t pendLen, pendCap;   /* small writes collect here: room for a member per thread, so that a stream of small writes is compressed by the whole team too */
  unsigned long long members, rawBytes, zBytes;
} MgPgz;

/* Every member says how long it is: its gzip header carries an extra field (RFC 1952 2.3.1.1, as bgzip's blocks do) with the subfield
   'M' 'G': the member's compressed size in the file and its uncompressed size, 4 bytes each.  gzread and gunzip skip extra fields; the
   library's own reader (mgGzipOpenRead) walks them to find every member without inflating any, and inflates them in parallel. */
#define PGZ_HDR 24                                /* 10 fixed bytes, XLEN (2), subfield id (2) + length (2) + 8 bytes of sizes */
static void put32 (unsigned char *p, unsigned v) { p[0] = (unsigned char) v; p[1] = (unsigned char) (v >> 8); p[2] = (unsigned char) (v >> 16); p[3] = (unsigned char) (v >> 24); }
static unsigned get32 (const unsigned char *p) { return (unsigned) p[0] | (unsigned) p[1] << 8 | (unsigned) p[2] << 16 | (unsigned) p[3] << 24; }

/* What these files hold -- index[]: a 4 * 2^bits byte table nine tenths zero; value[]: k-mers of 2k bits in 64-bit words; counts, offsets
   -- is runs of zero bytes between bytes that repeat nothing.  zlib's Z_RLE strategy (matches of distance one only) finds exactly those
   runs and skips the hash-chain search that finds nothing else: on index[] it is 5 times faster than the default strategy AND 4 % smaller,
   on value[] 7.7 times faster and 1 % larger (zlib level 6, one thread: 59 -> 291 and 13 -> 100 MB/s).  Data of another kind (small counts
   that alternate, text) compresses better the default way, so every member decides for itself on its first 128 KiB: RLE unless that
   costs more than 5 % of size. */
static int pgzStrategy (const unsigned char *src, size_t n, int level)
{
  const size_t sample = n < ((size_t) 128 << 10) ? n : (size_t) 128 << 10;
  if (sample < 4096) return Z_DEFAULT_STRATEGY;
  size_t len[2] = { 0, 0 };
  for (int which = 0 ; which < 2 ; ++which)
    { z_stream z; memset (&z, 0, sizeof (z));
      if (deflateInit2 (&z, level, Z_DEFLATED, -15, 8, which ? Z_RLE : Z_DEFAULT_STRATEGY) != Z_OK) return Z_DEFAULT_STRATEGY;
      const size_t cap = deflateBound (&z, (uLong) sample) + 64;
      unsigned char *out = (unsigned char *) malloc (cap);
      if (!out) { deflateEnd (&z); return Z_DEFAULT_STRATEGY; }
      z.next_in = (Bytef *) src; z.avail_in = (uInt) sample; z.next_out = out; z.avail_out = (uInt) cap;
      (void) deflate (&z, Z_FINISH);
      len[which] = cap - z.avail_out;
      deflateEnd (&z); free (out);
    }
  return len[1] * 100 <= len[0] * 105 ? Z_RLE : Z_DEFAULT_STRATEGY;
}

/* one member: gzip header (with the sizes) + deflate + crc32 + length, into a malloc ()ed block */
static unsigned char *pgzMember (const unsigned char *src, size_t n, int level, size_t *outLen)
{
  z_stream z; memset (&z, 0, sizeof (z));
  if (deflateInit2 (&z, level, Z_DEFLATED, 15 + 16, 8, pgzStrategy (src, n, level)) != Z_OK) return 0;
  unsigned char extra[12] = { 'M', 'G', 8, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
  gz_header gh; memset (&gh, 0, sizeof (gh));
  gh.os = 3; gh.extra = extra; gh.extra_len = sizeof (extra);
  if (deflateSetHeader (&z, &gh) != Z_OK) { deflateEnd (&z); return 0; }
  const size_t cap = deflateBound (&z, (uLong) n) + 64 + PGZ_HDR;
  unsigned char *out = (unsigned char *) malloc (cap);
  if (!out) { deflateEnd (&z); return 0; }
  z.next_in = (Bytef *) src; z.avail_in = (uInt) n; z.next_out = out; z.avail_out = (uInt) cap;
  const int rc = deflate (&z, Z_FINISH);
  *outLen = cap - z.avail_out;
  deflateEnd (&z);
  if (rc != Z_STREAM_END || *outLen < PGZ_HDR + 8 || out[3] != 4 || out[12] != 'M' || out[13] != 'G') { free (out); return 0; }
  put32 (out + 16, (unsigned) *outLen); put32 (out + 20, (unsigned) n);      /* (no header CRC: the field can be filled in afterwards) */
  return out;
}

static int pgzPut (MgPgz *p, const unsigned char *buf, size_t n)
{
  while (n)
    { const ssize_t w = write (p->fd, buf, n);
      if (w < 0) { if (errno == EINTR) continue; return -1; }
      buf += w; n -= (size_t) w;
    }
  return 0;
}

/* a run of whole members out of one array: threads take member numbers from a counter, the caller's thread writes them in order */
typedef struct
{ MgPgz *p; const unsigned char *src; size_t n, nMembers;
  pthread_mutex_t mu; pthread_cond_t cv;
  size_t next, written;                           /* next member to take; members written so far */
  unsigned char **out; size_t *outLen; int failed;
} PgzRun;
#define PGZ_AHEAD 3                               /* a thread does not start member j before member j - PGZ_AHEAD * T is written (bounds the memory) */

static void *pgzWorker (void *v)
{
  PgzRun *r = (PgzRun *) v;
  for (;;)
    { pthread_mutex_lock (&r->mu);
      while (!r->failed && r->next < r->nMembers && r->next >= r->written + (size_t) PGZ_AHEAD * r->p->T) pthread_cond_wait (&r->cv, &r->mu);
      if (r->failed || r->next >= r->nMembers) { pthread_mutex_unlock (&r->mu); return 0; }
      const size_t j = r->next++;
      pthread_mutex_unlock (&r->mu);
      const size_t off = j * MG_PGZ_MEMBER, len = r->n - off < MG_PGZ_MEMBER ? r->n - off : MG_PGZ_MEMBER;
      size_t zl = 0;
      unsigned char *z = pgzMember (r->src + off, len, r->p->level, &zl);
      pthread_mutex_lock (&r->mu);
      if (!z) r->failed = 1; else { r->out[j] = z; r->outLen[j] = zl; }
      pthread_cond_broadcast (&r->cv);
      pthread_mutex_unlock (&r->mu);
    }
}

static int pgzRun (MgPgz *p, const unsigned char *src, size_t n)
{
  if (!n) return 0;
  PgzRun r; memset (&r, 0, sizeof (r));
  r.p = p; r.src = src; r.n = n; r.nMembers = (n + MG_PGZ_MEMBER - 1) / MG_PGZ_MEMBER;
  r.out = (unsigned char **) calloc (r.nMembers, sizeof (unsigned char *)); r.outLen = (size_t *) calloc (r.nMembers, sizeof (size_t));
  if (!r.out || !r.outLen) { free (r.out); free (r.outLen); return -1; }
  pthread_mutex_init (&r.mu, 0); pthread_cond_init (&r.cv, 0);
  int T = p->T; if ((size_t) T > r.nMembers) T = (int) r.nMembers;
  pthread_t th[MG_PGZ_MAXT]; int started = 0;
  for (int t = 0 ; t < T ; ++t) if (pthread_create (&th[started], 0, pgzWorker, &r) == 0) ++started;
  int rc = 0;
  for (size_t j = 0 ; j < r.nMembers && !rc ; ++j)
    { unsigned char *z = 0; size_t zl = 0;
      if (!started)                                               /* no thread to be had: member by member, here */
        { const size_t off = j * MG_PGZ_MEMBER, len = n - off < MG_PGZ_MEMBER ? n - off : MG_PGZ_MEMBER;
          if (!(z = pgzMember (src + off, len, p->level, &zl))) { rc = -1; break; }
        }
      else
        { pthread_mutex_lock (&r.mu);
          while (!r.failed && !r.out[j]) pthread_cond_wait (&r.cv, &r.mu);
          z = r.out[j]; zl = r.outLen[j]; r.out[j] = 0;
          pthread_mutex_unlock (&r.mu);
          if (!z) { rc = -1; break; }
        }
      if (pgzPut (p, z, zl)) rc = -1;
      free (z);
      p->zBytes += zl; ++p->members;
      if (started) { pthread_mutex_lock (&r.mu); r.written = j + 1; pthread_cond_broadcast (&r.cv); pthread_mutex_unlock (&r.mu); }
    }
  if (rc) { pthread_mutex_lock (&r.mu); r.failed = 1; pthread_cond_broadcast (&r.cv); pthread_mutex_unlock (&r.mu); }
  for (int t = 0 ; t < started ; ++t) pthread_join (th[t], 0);
  for (size_t j = 0 ; j < r.nMembers ; ++j) free (r.out[j]);
  free (r.out); free (r.outLen);
  pthread_mutex_destroy (&r.mu); pthread_cond_destroy (&r.cv);
  p->rawBytes += n;
  return rc;
}

static ssize_t pgzCookieWrite (void *c, const char *buf, size_t n)
{
  MgPgz *p = (MgPgz *) c;
  if (p->err) return 0;
  const unsigned char *b = (const unsigned char *) buf; size_t left = n;
  if (p->pendLen)                                                 /* top the collected bytes up to a member first: the order of the bytes is the file's */
    { const size_t take = p->pendCap - p->pendLen < left ? p->pendCap - p->pendLen : left;
      memcpy (p->pend + p->pendLen, b, take); p->pendLen += take; b += take; left -= take;
      if (p->pendLen == p->pendCap) { if (pgzRun (p, p->pend, p->pendLen)) { p->err = 1; return 0; } p->pendLen = 0; }
    }
  const size_t whole = left / MG_PGZ_MEMBER * MG_PGZ_MEMBER;      /* whole members straight from the caller's array */
  if (whole) { if (pgzRun (p, b, whole)) { p->err = 1; return 0; } b += whole; left -= whole; }
  if (left) { memcpy (p->pend + p->pendLen, b, left); p->pendLen += left; }      /* (less than a member; pend is empty or has room: pendCap >= 2 members) */
  return (ssize_t) n;
}

static int pgzCookieClose (void *c)
{
  MgPgz *p = (MgPgz *) c;
  int rc = p->err ? -1 : 0;
  if (!rc && (p->pendLen || !p->members))                         /* (an empty file is still one -- empty -- member, as gzopen "w" + gzclose leaves it) */
    { if (p->pendLen) rc = pgzRun (p, p->pend, p->pendLen);
      else { size_t zl = 0; unsigned char *z = pgzMember ((const unsigned char *) "", 0, p->level, &zl); if (!z || pgzPut (p, z, zl)) rc = -1; free (z); }
    }
  if (mgKnobs ()->uploadTiming == 1 && p->rawBytes)
    fprintf (stderr, "mgPgz: %llu members, %llu -> %llu bytes, %d threads\n", p->members, p->rawBytes, p->zBytes, p->T);
  if (close (p->fd)) rc = -1;
  free (p->pend); free (p);
  return rc;
}

/* `name` opened for writing as a gzip file of independent members; 0 if it cannot be */
FILE *mgGzipOpenWrite (const char *name)
{
  const int fd = open (name, O_WRONLY | O_CREAT | O_TRUNC, 0666);
  if (fd < 0) return 0;
  MgPgz *p = (MgPgz *) calloc (1, sizeof (MgPgz));
  if (!p) { close (fd); return 0; }
  long t = mgKnobs ()->gzipThreads; if (t == MG_KNOB_UNSET || t <= 0) t = mgCpuBudget ();
  p->T = t > MG_PGZ_MAXT ? MG_PGZ_MAXT : (int) t;
  p->pendCap = (size_t) (p->T < 2 ? 2 : p->T) * MG_PGZ_MEMBER;    /* (virtual until written: a file of small writes touches what it fills) */
  p->pend = (unsigned char *) malloc (p->pendCap);
  if (!p->pend) { free (p); close (fd); return 0; }
  p->fd = fd; p->level = Z_DEFAULT_COMPRESSION;
  cookie_io_functions_t io = { 0, pgzCookieWrite, 0, pgzCookieClose };
  FILE *f = fopencookie (p, "w", io);
  if (!f) { free (p->pend); free (p); close (fd); return 0; }
  return f;
}


/* ---- reading: the members found by their extra fields, inflated by the team ----
 * modsetRead (modset.c:90-104) is a few small freads and then index[] / value[] / depth[] / info[] in one fread each: gigabytes that one
 * inflate stream delivers at a few hundred MB/s.  A file the writer above made lists its members' sizes, so: the table of members is
 * built by walking the headers (24 bytes read per member), a read that covers whole members has them inflated straight into the
 * caller's array by the team, and what is left of a read -- the head and tail of a large one, all of a small one -- is served from
 * one member inflated into a buffer of the reader's.  Any other gzip file (or plain file) goes through gzopen as before. */
typedef struct { off_t off; unsigned csize, usize; } PgzMem;
typedef struct
{ int fd; int T; PgzMem *mem; size_t nMem; unsigned most;
  size_t cur; unsigned at;                        /* the position: member, byte inside it */
  unsigned char *win; size_t winFirst, winCount;  /* members winFirst .. winFirst + winCount - 1 inflated, `most` bytes apart: the read-ahead window, a member per thread */
  int err;
} MgPgzIn;

static int pgzInflateMember (int fd, const PgzMem *m, unsigned char *dst)
{
  unsigned char *z = (unsigned char *) malloc (m->csize);
  if (!z) return -1;
  size_t got = 0;
  while (got < m->csize)
    { const ssize_t r = pread (fd, z + got, m->csize - got, m->off + (off_t) got);
      if (r <= 0) { if (r < 0 && errno == EINTR) continue; free (z); return -1; }
      got += (size_t) r;
    }
  z_stream s; memset (&s, 0, sizeof (s));
  int rc = -1;
  if (inflateInit2 (&s, 15 + 16) == Z_OK)
    { s.next_in = z; s.avail_in = m->csize; s.next_out = dst; s.avail_out = m->usize;
      const int r = inflate (&s, Z_FINISH);       /* (the gzip wrapper: zlib checks the member's CRC and length itself) */
      if (r == Z_STREAM_END && s.avail_out == 0) rc = 0;
      inflateEnd (&s);
    }
  free (z);
  return rc;
}

/* members first .. first + n - 1 inflated by the team: one after the other into dst (stride 0), or `stride` bytes apart */
typedef struct { MgPgzIn *p; size_t first, n; unsigned char *dst; size_t stride; size_t next; pthread_mutex_t mu; int failed; } PgzInRun;
static void *pgzInWorker (void *v)
{
  PgzInRun *r = (PgzInRun *) v;
  for (;;)
    { pthread_mutex_lock (&r->mu);
      const size_t j = r->next++;
      const int stop = r->failed;
      pthread_mutex_unlock (&r->mu);
      if (j >= r->n || stop) return 0;
      size_t at = j * r->stride;
      if (!r->stride) for (size_t q = 0 ; q < j ; ++q) at += r->p->mem[r->first + q].usize;
      if (pgzInflateMember (r->p->fd, &r->p->mem[r->first + j], r->dst + at))
        { pthread_mutex_lock (&r->mu); r->failed = 1; pthread_mutex_unlock (&r->mu); }
    }
}
static int pgzInRun (MgPgzIn *p, size_t first, size_t n, unsigned char *dst, size_t stride)
{
  PgzInRun r; memset (&r, 0, sizeof (r));
  r.p = p; r.first = first; r.n = n; r.dst = dst; r.stride = stride;
  pthread_mutex_init (&r.mu, 0);
  int T = p->T; if ((size_t) T > n) T = (int) n;
  pthread_t th[MG_PGZ_MAXT]; int started = 0;
  for (int t = 1 ; t < T ; ++t) if (pthread_create (&th[started], 0, pgzInWorker, &r) == 0) ++started;
  pgzInWorker (&r);
  for (int t = 0 ; t < started ; ++t) pthread_join (th[t], 0);
  pthread_mutex_destroy (&r.mu);
  return r.failed ? -1 : 0;
}

/* (glibc hands a cookie's read function its own buffer's worth at a time -- fread on such a FILE goes through the buffer whatever
   the size asked for -- so the reads that arrive here are small: what makes the team useful is the window, a member per thread
   inflated ahead; a read that does cover whole members, e.g. with a large setvbuf, has them inflated in place) */
static ssize_t pgzCookieRead (void *c, char *out, size_t n)
{
  MgPgzIn *p = (MgPgzIn *) c;
  if (p->err) return -1;
  size_t done = 0;
  while (done < n && p->cur < p->nMem)
    { const PgzMem *m = &p->mem[p->cur];
      const int inWin = p->cur >= p->winFirst && p->cur < p->winFirst + p->winCount;
      if (!inWin && p->at == 0)
        { size_t k = 0, bytes = 0;
          while (p->cur + k < p->nMem && bytes + p->mem[p->cur + k].usize <= n - done) { bytes += p->mem[p->cur + k].usize; ++k; }
          if (k)
            { if (pgzInRun (p, p->cur, k, (unsigned char *) out + done, 0)) { p->err = 1; return -1; }
              done += bytes; p->cur += k;
              continue;
            }
        }
      if (!inWin)
        { const size_t k = p->nMem - p->cur < (size_t) p->T ? p->nMem - p->cur : (size_t) p->T;
          if (pgzInRun (p, p->cur, k, p->win, p->most)) { p->err = 1; return -1; }
          p->winFirst = p->cur; p->winCount = k;
        }
      const unsigned char *src = p->win + (p->cur - p->winFirst) * (size_t) p->most;
      const size_t take = m->usize - p->at < n - done ? m->usize - p->at : n - done;
      memcpy (out + done, src + p->at, take);
      done += take; p->at += (unsigned) take;
      if (p->at == m->usize) { ++p->cur; p->at = 0; }
    }
  return (ssize_t) done;
}
static int pgzInClose (void *c)
{ MgPgzIn *p = (MgPgzIn *) c; const int rc = p->err ? -1 : 0; close (p->fd); free (p->mem); free (p->win); free (p); return rc; }

/* `name` opened for reading if it is a file of this writer's members from its first byte to its last; 0 otherwise (the caller then takes gzopen) */
FILE *mgGzipOpenRead (const char *name)
{
  const int fd = open (name, O_RDONLY);
  if (fd < 0) return 0;
  const off_t end = lseek (fd, 0, SEEK_END);
  PgzMem *mem = 0; size_t nMem = 0, cap = 0; unsigned most = 0;
  off_t at = 0;
  int ok = end >= PGZ_HDR;
  while (ok && at < end)
    { unsigned char h[PGZ_HDR];
      if (pread (fd, h, PGZ_HDR, at) != PGZ_HDR) { ok = 0; break; }
      if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || h[3] != 4 || h[10] != 12 || h[11] != 0 || h[12] != 'M' || h[13] != 'G' || h[14] != 8 || h[15] != 0) { ok = 0; break; }
      const unsigned cs = get32 (h + 16), us = get32 (h + 20);
      if (cs < PGZ_HDR + 8 || at + (off_t) cs > end || us > 8 * MG_PGZ_MEMBER) { ok = 0; break; }      /* (a size this writer never makes: not a file of its own) */
      if (nMem == cap) { cap = cap ? 2 * cap : 256; PgzMem *q = (PgzMem *) realloc (mem, cap * sizeof (PgzMem)); if (!q) { ok = 0; break; } mem = q; }
      mem[nMem].off = at; mem[nMem].csize = cs; mem[nMem].usize = us; ++nMem;
      if (us > most) most = us;
      at += cs;
    }
  if (!ok || at != end || !nMem) { free (mem); close (fd); return 0; }
  MgPgzIn *p = (MgPgzIn *) calloc (1, sizeof (MgPgzIn));
  if (!p) { free (mem); close (fd); return 0; }
  long t = mgKnobs ()->gzipThreads; if (t == MG_KNOB_UNSET || t <= 0) t = mgCpuBudget ();
  p->T = t > MG_PGZ_MAXT ? MG_PGZ_MAXT : (int) t;
  if ((size_t) p->T > nMem) p->T = (int) nMem;
  p->fd = fd; p->mem = mem; p->nMem = nMem; p->most = most ? most : 1;
  p->win = (unsigned char *) malloc ((size_t) p->T * p->most);       /* (virtual until a window is inflated into it) */
  if (!p->win) { free (p); free (mem); close (fd); return 0; }
  cookie_io_functions_t io = { pgzCookieRead, 0, 0, pgzInClose };
  FILE *f = fopencookie (p, "r", io);
  if (!f) { free (p->win); free (p); free (mem); close (fd); return 0; }
  (void) setvbuf (f, 0, _IOFBF, (size_t) 1 << 20);                  /* a megabyte a call instead of glibc's 4 KiB */
  return f;
}
