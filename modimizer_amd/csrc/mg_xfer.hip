/* mg_xfer.hip — whole arrays between HBM and the caller's (pageable) host arrays at bus speed.
 *
 * The reference's Modset is transparent (modset.h:17-28): its callers index ms->value / depth / index themselves
 * (modutils.c:26,69,186-198, modset.c:79-88), so what the device built has to be IN those malloc ()ed arrays before
 * control returns to them.  A hipMemcpy into pageable memory goes through the runtime's single staging path at a few
 * GB/s; here a team of host threads moves the array in pieces: every thread owns a slice of the pieces, two page-locked
 * blocks, a stream and two events; it has the copy engine fill one block while it empties the other into the
 * destination (memcpy, or a saturating 16-bit add: depth[i] = min (65535, depth[i] + pending[i]), modutils.c:26 applied
 * `pending` times).  Nothing is allocated per call: the blocks, streams and events are made once per device and stay
 * until mgReleaseBuffers ().  Nothing crosses between the threads: no queue, no hand-off, each one's pieces are its own.
 * There is one such set per DEVICE, with a lock of its own: the host threads of a process that drives several GPUs
 * (examples/multi_gpu.c, one thread per device) mirror their results side by side, not one after the other.
 */
#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <mutex>
#include <thread>
#include <vector>
#include "mg_common.h"
#include "mg_internal.h"
#include "mg_xfer.h"

#define MG_XFER_MAXT   16
#define MG_XFER_MAXDEV 128        /* (an 8-GPU node in its 8-partition mode shows 64 devices) */
#define MG_XFER_PIECE_MAX ((size_t) 16 << 20)

/* A lane has a copy stream of its own: four queues keep the link at 52 - 54 GB/s, all lanes on the device's default stream reach 46
   (tools/xfer_probe.py) -- and a call that follows such a transfer on the default stream was seen to wait 30 ms for it to drain.
   But the first streams a process creates cost 10 ms apiece (42 ms for four: half of a first mgReferenceRead's mirror), so they are
   made AHEAD of their first use, by a thread started when a Modset gets its device table (mgXferWarm): by the time there is
   something to mirror they exist.  MODGPU_XFER_STREAMS=0 puts the copies on the default stream (dev). */
struct MgXferLane { hipStream_t st; bool own, made; char *pin[2]; hipEvent_t ev[2]; };
struct MgXferCtx
{ int T = 0;
  size_t piece = (size_t) 4 << 20;                         /* bytes per piece: 80 us on the link, long enough to hide a copy call (measured, 4 threads with a stream each: 1 MiB 54 GB/s, 4 MiB 49 - 54, 16 MiB 52; MODGPU_XFER_PIECE_KB: dev) */
  MgXferLane lane[MG_XFER_MAXT] = {};
  std::mutex lock;
  std::mutex warmLock; std::thread warm;                   /* the thread that makes the lanes ahead of their first use: joined before another is started, on release, */
  ~MgXferCtx () { if (warm.joinable ()) warm.join (); }    /* and when the library's statics go (a std::thread destroyed while it can still be joined ends the process) */
};
static MgXferCtx gXs[MG_XFER_MAXDEV];                     /* by device number */

int mgXferThreads (void)
{
  const long kv = mgKnobs ()->xferThreads;
  long v = (kv != MG_KNOB_UNSET && kv > 0) ? kv : (mgCpuBudget () < 4 ? mgCpuBudget () : 4);      /* measured (tools/xfer_probe.py, 1 GiB, 16 CPUs granted): 4 threads 52 GB/s device to host, 8: 47, 16: 43 -- four memcpy streams keep up with the link, more get in each other's way; host to device 55 either way */
  if (v > MG_XFER_MAXT) v = MG_XFER_MAXT;
  if (v < 1) v = 1;
  return (int) v;
}
extern "C" int mgXferThreadCount (void) { return mgXferThreads (); }

/* streams and events belong to a device: made and released with that device current */
static void mgXferDropLocked (MgXferCtx &X, int dev)
{
  if (!X.T) return;
  int before = -1;
  if (hipGetDevice (&before) != hipSuccess) { (void) hipGetLastError (); before = -1; }
  if (before != dev) (void) hipSetDevice (dev);
  for (int t = 0 ; t < X.T ; ++t)
    { MgXferLane &l = X.lane[t];
      if (l.made) (void) hipStreamSynchronize (l.st);
      if (l.own && l.st) (void) hipStreamDestroy (l.st);
      for (int b = 0 ; b < 2 ; ++b) { if (l.pin[b]) (void) hipHostFree (l.pin[b]); if (l.ev[b]) (void) hipEventDestroy (l.ev[b]); }
      memset (&l, 0, sizeof (l));
    }
  X.T = 0;
  if (before >= 0 && before != dev) (void) hipSetDevice (before);
}

extern "C" void mgXferReleaseBuffers (void)
{
  for (int dev = 0 ; dev < MG_XFER_MAXDEV ; ++dev)
    { MgXferCtx &X = gXs[dev];
      { std::lock_guard<std::mutex> w (X.warmLock); if (X.warm.joinable ()) X.warm.join (); }
      std::lock_guard<std::mutex> g (X.lock);
      mgXferDropLocked (X, dev);
    }
}

/* the current device's set (0: a device number this file has no room for) */
static MgXferCtx *mgXferCtxOf (int *devOut)
{
  int dev = 0;
  if (hipGetDevice (&dev) != hipSuccess) { (void) hipGetLastError (); return 0; }
  if (dev < 0 || dev >= MG_XFER_MAXDEV) return 0;
  *devOut = dev;
  return &gXs[dev];
}

/* room for T lanes (X.lock held): a lane's stream, blocks and events are made by the thread that runs the lane, on its first piece
   (the threads page-lock their blocks side by side instead of one after the other) */
static void mgXferPrepareLocked (MgXferCtx &X, int dev, int T)
{
  const long kb = mgKnobs ()->xferPieceKb;
  size_t piece = (kb != MG_KNOB_UNSET && kb >= 64) ? (size_t) kb << 10 : (size_t) 4 << 20;
  if (piece > MG_XFER_PIECE_MAX) piece = MG_XFER_PIECE_MAX;
  if (piece != X.piece) { mgXferDropLocked (X, dev); X.piece = piece; }      /* the blocks are a piece long */
  if (T > X.T) { for (int t = X.T ; t < T ; ++t) memset (&X.lane[t], 0, sizeof (MgXferLane)); X.T = T; }
}
static hipError_t mgXferLaneMake (MgXferLane &l, size_t piece)
{
  if (l.made) return hipSuccess;
  hipError_t e;
  struct timespec a, b1, c; clock_gettime (CLOCK_MONOTONIC, &a);
  if (!l.own && mgKnobs ()->xferStreams != 0) { if ((e = hipStreamCreateWithFlags (&l.st, hipStreamNonBlocking)) != hipSuccess) return e; l.own = true; }      /* (else, dev: the device's default stream, 0) */
  clock_gettime (CLOCK_MONOTONIC, &b1);
  for (int b = 0 ; b < 2 ; ++b)
    { if (!l.pin[b] && (e = hipHostMalloc ((void **) &l.pin[b], piece, hipHostMallocPortable)) != hipSuccess) return e;
      if (!l.ev[b] && (e = hipEventCreateWithFlags (&l.ev[b], hipEventDisableTiming)) != hipSuccess) return e;
    }
  l.made = true;
  clock_gettime (CLOCK_MONOTONIC, &c);
  if (mgKnobs ()->uploadTiming == 1)
    fprintf (stderr, "mgXferLaneMake: stream %.2f ms, blocks + events %.2f ms\n", (b1.tv_sec - a.tv_sec) * 1e3 + (b1.tv_nsec - a.tv_nsec) * 1e-6, (c.tv_sec - b1.tv_sec) * 1e3 + (c.tv_nsec - b1.tv_nsec) * 1e-6);
  return hipSuccess;
}

static inline void mgXferApply (char *dst, const char *src, size_t bytes, int op)
{
  if (op == MG_XFER_COPY) { memcpy (dst, src, bytes); return; }
  U16 *d = (U16 *) dst; const U16 *s = (const U16 *) src;
  const size_t n = bytes >> 1;
  for (size_t i = 0 ; i < n ; ++i) { const U32 v = (U32) d[i] + s[i]; d[i] = (U16) (v > 0xffffu ? 0xffffu : v); }   /* (the compiler makes paddusw of it) */
}

/* thread t of T, device to host: pieces t, t + T, t + 2T, ... */
static hipError_t mgXferLaneDown (int dev, MgXferLane &l, size_t piece, int t, int T, char *dst, const char *src, size_t bytes, int op)
{
  hipError_t e = hipSetDevice (dev);
  if (e != hipSuccess || (e = mgXferLaneMake (l, piece)) != hipSuccess) return e;
  const size_t nPieces = (bytes + piece - 1) / piece;
  size_t p = (size_t) t; int b = 0;
  auto issue = [&] (size_t pc, int buf) -> hipError_t
    { const size_t off = pc * piece, len = bytes - off < piece ? bytes - off : piece;
      hipError_t x = hipMemcpyAsync (l.pin[buf], src + off, len, hipMemcpyDeviceToHost, l.st);
      return x != hipSuccess ? x : hipEventRecord (l.ev[buf], l.st);
    };
  if (p < nPieces && (e = issue (p, b)) != hipSuccess) return e;
  for ( ; p < nPieces ; p += (size_t) T, b ^= 1)
    { const size_t next = p + (size_t) T;
      if (next < nPieces && (e = issue (next, b ^ 1)) != hipSuccess) return e;      /* the other block fills while this one is emptied */
      if ((e = hipEventSynchronize (l.ev[b])) != hipSuccess) return e;
      const size_t off = p * piece, len = bytes - off < piece ? bytes - off : piece;
      mgXferApply (dst + off, l.pin[b], len, op);
    }
  return hipSuccess;
}

/* The other direction, for the arrays a device table is (re)built from (ms->value, ms->depth): the team fills the
   page-locked blocks from the pageable source and the copy engine drains them. */
static hipError_t mgXferLaneUp (int dev, MgXferLane &l, size_t piece, int t, int T, char *dDst, const char *hSrc, size_t bytes)
{
  hipError_t e = hipSetDevice (dev);
  if (e != hipSuccess || (e = mgXferLaneMake (l, piece)) != hipSuccess) return e;
  const size_t nPieces = (bytes + piece - 1) / piece;
  int b = 0; size_t done = 0;
  for (size_t p = (size_t) t ; p < nPieces ; p += (size_t) T, b ^= 1, ++done)
    { if (done >= 2 && (e = hipEventSynchronize (l.ev[b])) != hipSuccess) return e;      /* the copy that last read this block */
      const size_t off = p * piece, len = bytes - off < piece ? bytes - off : piece;
      memcpy (l.pin[b], hSrc + off, len);
      if ((e = hipMemcpyAsync (dDst + off, l.pin[b], len, hipMemcpyHostToDevice, l.st)) != hipSuccess) return e;
      if ((e = hipEventRecord (l.ev[b], l.st)) != hipSuccess) return e;
    }
  return hipStreamSynchronize (l.st);
}

/* op < 0: host to device (a = device destination, b = host source); else device to host with that op (a = host destination, b = device source) */
static MgStatus mgXferRun (void *a, const void *b, size_t bytes, int op, const char *what)
{
  if (!bytes) return MG_OK;
  struct timespec q0; clock_gettime (CLOCK_MONOTONIC, &q0);
  int dev = 0;
  MgXferCtx *Xp = mgXferCtxOf (&dev);
  if (!Xp)                                                 /* more devices than this file has sets for: the runtime's own copy */
    { if (op == MG_XFER_SATADD16) { mgSetError ("%s: device number beyond %d", what, MG_XFER_MAXDEV - 1); return MG_ERR_ARG; }
      MG_HIP (hipMemcpy (a, b, bytes, op < 0 ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost));
      return MG_OK;
    }
  MgXferCtx &X = *Xp;
  std::lock_guard<std::mutex> g (X.lock);
  int T = mgXferThreads ();
  mgXferPrepareLocked (X, dev, T);
  const size_t piece = X.piece, nPieces = (bytes + piece - 1) / piece;
  if ((size_t) T > nPieces) T = (int) nPieces;
  hipError_t err[MG_XFER_MAXT];
  for (int t = 0 ; t < T ; ++t) err[t] = hipSuccess;
  auto run = [&] (int t)
    { err[t] = op < 0 ? mgXferLaneUp (dev, X.lane[t], piece, t, T, (char *) a, (const char *) b, bytes)
                      : mgXferLaneDown (dev, X.lane[t], piece, t, T, (char *) a, (const char *) b, bytes, op); };
  std::vector<std::thread> th;
  int started = 1;                                             /* lane 0 is the caller's */
  try { for (int t = 1 ; t < T ; ++t) { th.emplace_back (run, t); ++started; } }
  catch (...) { }                                              /* the system gave fewer threads: the pieces of those that did not start are done below */
  run (0);
  for (auto &x : th) x.join ();
  for (int t = started ; t < T ; ++t) run (t);
  for (int t = 0 ; t < T ; ++t) if (err[t] != hipSuccess) return mgHipFail (err[t], what);
  if (mgKnobs ()->uploadTiming == 1) { struct timespec q1; clock_gettime (CLOCK_MONOTONIC, &q1); fprintf (stderr, "%s: %zu bytes, device %d, %d threads, %.2f ms\n", what, bytes, dev, T, (q1.tv_sec - q0.tv_sec) * 1e3 + (q1.tv_nsec - q0.tv_nsec) * 1e-6); }
  return MG_OK;
}

MgStatus mgXferD2H (void *hostDst, const void *devSrc, size_t bytes, int op) { return mgXferRun (hostDst, devSrc, bytes, op, "mgXferD2H"); }
MgStatus mgXferH2D (void *devDst, const void *hostSrc, size_t bytes) { return mgXferRun (devDst, hostSrc, bytes, -1, "mgXferH2D"); }

/* public forms (include/modgpu.h) */
extern "C" MgStatus mgCopyD2HBig (void *hostDst, const void *devSrc, size_t bytes)
{ MgStatus s = mgEnsureDevice (); if (s) return s; MG_HIP (hipDeviceSynchronize ()); return mgXferD2H (hostDst, devSrc, bytes, MG_XFER_COPY); }
extern "C" MgStatus mgCopyH2DBig (void *devDst, const void *hostSrc, size_t bytes)
{ MgStatus s = mgEnsureDevice (); if (s) return s; MG_HIP (hipDeviceSynchronize ()); return mgXferH2D (devDst, hostSrc, bytes); }

/* Host to device for an array most of which may never have been written: pages of an anonymous mapping that were never touched read as
 * zero, and READING them maps the shared zero page -- after which the first write to each (the mirror coming back) is a copy-on-write
 * fault per 4 KiB instead of a fresh huge page (ms->info of a new Modset, calloc ()ed by modsetCreate: 47 MB came back at 4 GB/s).
 * /proc/self/pagemap says which pages exist (present or swapped); only those are read and sent, the rest of the device array is
 * cleared.  Only for private anonymous memory (checked: mgRangeIsPrivateAnon); anything else takes the plain copy. */
/* [a, a + bytes) lies entirely in private anonymous mappings (/proc/self/maps: perms "..p", inode 0) -- what malloc () / calloc () hand out.
   Only there does "neither present nor swapped" mean "never written, reads as zero": a file-backed or shared mapping a caller put in the
   place of a transparent struct's array (ms->info is the caller's to replace, modset.h:17-28) has non-resident pages with contents */
static bool mgRangeIsPrivateAnon (size_t a, size_t bytes)
{
  FILE *f = fopen ("/proc/self/maps", "r");
  if (!f) return false;
  char line[512];
  size_t at = a; const size_t end = a + bytes;
  bool ok = true;
  while (ok && at < end && fgets (line, sizeof line, f))
    { unsigned long lo, hi, off, ino; char perms[8]; unsigned maj, mnr;
      if (sscanf (line, "%lx-%lx %7s %lx %x:%x %lu", &lo, &hi, perms, &off, &maj, &mnr, &ino) != 7) continue;
      if (hi <= at) continue;
      if (lo > at) { ok = false; break; }                  /* a hole (the list is sorted by address) */
      if (perms[3] != 'p' || ino != 0) { ok = false; break; }
      at = hi;
    }
  fclose (f);
  return ok && at >= end;
}

MgStatus mgXferH2DSparse (void *devDst, const void *hostSrc, size_t bytes)
{
  if (bytes < ((size_t) 4 << 20)) return mgXferH2D (devDst, hostSrc, bytes);
  if (!mgRangeIsPrivateAnon ((size_t) hostSrc, bytes)) return mgXferH2D (devDst, hostSrc, bytes);
  const size_t pg = (size_t) sysconf (_SC_PAGESIZE);
  const size_t a0 = (size_t) hostSrc, firstPage = a0 / pg, lastPage = (a0 + bytes - 1) / pg, nPages = lastPage - firstPage + 1;
  U64 *ent = (U64 *) malloc (nPages * 8);
  const int fd = open ("/proc/self/pagemap", O_RDONLY);
  bool ok = ent && fd >= 0;
  for (size_t got = 0 ; ok && got < nPages * 8 ; )
    { const ssize_t r = pread (fd, (char *) ent + got, nPages * 8 - got, (off_t) (firstPage * 8 + got));
      if (r <= 0) ok = false; else got += (size_t) r;
    }
  if (fd >= 0) close (fd);
  if (!ok) { free (ent); return mgXferH2D (devDst, hostSrc, bytes); }
  MgStatus s = MG_OK;
  if (hipMemset (devDst, 0, bytes) != hipSuccess || hipDeviceSynchronize () != hipSuccess) { free (ent); return mgHipFail (hipGetLastError (), "mgXferH2DSparse"); }
  for (size_t p = 0 ; p < nPages && !s ; )
    { if (!(ent[p] >> 62)) { ++p; continue; }              /* bit 63 present, bit 62 swapped: neither = never written */
      size_t q = p; while (q < nPages && (ent[q] >> 62)) ++q;
      size_t b0 = (firstPage + p) * pg, b1 = (firstPage + q) * pg;
      if (b0 < a0) b0 = a0;
      if (b1 > a0 + bytes) b1 = a0 + bytes;
      s = mgXferH2D ((char *) devDst + (b0 - a0), (const char *) b0, b1 - b0);
      p = q;
    }
  free (ent);
  return s;
}

/* see MgXferLane: called where a transfer is certain to follow (a Modset's device table has just been made) */
void mgXferWarm (void)
{
  int dev = 0;
  MgXferCtx *Xp = mgXferCtxOf (&dev);
  if (!Xp) return;
  const int T = mgXferThreads ();
  { std::unique_lock<std::mutex> g (Xp->lock, std::try_to_lock);
    if (!g.owns_lock ()) return;                          /* a transfer (or a warm-up) is running: the lanes exist or are being made */
    if (Xp->T >= T) { bool all = true; for (int t = 0 ; t < T ; ++t) all = all && Xp->lane[t].made; if (all) return; }
  }
  std::lock_guard<std::mutex> w (Xp->warmLock);
  if (Xp->warm.joinable ()) Xp->warm.join ();
  try
    { Xp->warm = std::thread ([dev, T, Xp]
        { if (hipSetDevice (dev) != hipSuccess) return;
          std::lock_guard<std::mutex> g (Xp->lock);         /* a transfer that comes before this is done waits here, and finds the lanes made */
          mgXferPrepareLocked (*Xp, dev, T);
          for (int t = 0 ; t < T ; ++t) if (mgXferLaneMake (Xp->lane[t], Xp->piece) != hipSuccess) { (void) hipGetLastError (); return; }
        });
    }
  catch (...) { }                                          /* no thread to be had: the lanes are made when they are first used */
}
