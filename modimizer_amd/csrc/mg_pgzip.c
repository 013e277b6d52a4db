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

/* one member: gzip header + deflate + crc32 + length, into a malloc ()ed block */
static unsigned char *pgzMember (const unsigned char *src, size_t n, int level, size_t *outLen)
{
  z_stream z; memset (&z, 0, sizeof (z));
  if (deflateInit2 (&z, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return 0;
  const size_t cap = deflateBound (&z, (uLong) n) + 64;
  unsigned char *out = (unsigned char *) malloc (cap);
  if (!out) { deflateEnd (&z); return 0; }
  z.next_in = (Bytef *) src; z.avail_in = (uInt) n; z.next_out = out; z.avail_out = (uInt) cap;
  const int rc = deflate (&z, Z_FINISH);
  *outLen = cap - z.avail_out;
  deflateEnd (&z);
  if (rc != Z_STREAM_END) { free (out); return 0; }
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
