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
 */
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <thread>
#include <vector>
#include "mg_common.h"
#include "mg_internal.h"
#include "mg_xfer.h"

#define MG_XFER_MAXT   16
#define MG_XFER_PIECE  ((size_t) 4 << 20)              /* bytes per piece: 80 us on the link, long enough to hide a copy call */

struct MgXferLane { hipStream_t st; char *pin[2]; hipEvent_t ev[2]; };
static struct MgXferCtx { int dev = -1; int T = 0; MgXferLane lane[MG_XFER_MAXT]; std::mutex lock; } gX;

static void mgXferDropLocked (void)
{
  for (int t = 0 ; t < gX.T ; ++t)
    { MgXferLane &l = gX.lane[t];
      if (l.st) { (void) hipStreamSynchronize (l.st); (void) hipStreamDestroy (l.st); }
      for (int b = 0 ; b < 2 ; ++b) { if (l.pin[b]) (void) hipHostFree (l.pin[b]); if (l.ev[b]) (void) hipEventDestroy (l.ev[b]); }
      memset (&l, 0, sizeof (l));
    }
  gX.T = 0; gX.dev = -1;
}

int mgXferThreads (void)
{
  const long kv = mgKnobs ()->xferThreads;
  long v = (kv != MG_KNOB_UNSET && kv > 0) ? kv : mgCpuBudget ();
  if (v > MG_XFER_MAXT) v = MG_XFER_MAXT;
  if (v < 1) v = 1;
  return (int) v;
}

extern "C" int mgXferThreadCount (void) { return mgXferThreads (); }
extern "C" void mgXferReleaseBuffers (void) { std::lock_guard<std::mutex> g (gX.lock); mgXferDropLocked (); }

/* lanes 0 .. T-1 on the current device (gX.lock held) */
static MgStatus mgXferPrepareLocked (int T)
{
  int dev = 0; MG_HIP (hipGetDevice (&dev));
  if (gX.dev >= 0 && gX.dev != dev) mgXferDropLocked ();      /* streams and events belong to a device */
  gX.dev = dev;
  for (int t = gX.T ; t < T ; ++t)
    { MgXferLane &l = gX.lane[t];
      memset (&l, 0, sizeof (l));
      gX.T = t + 1;                                           /* (a lane made by halves is still dropped whole) */
      MG_HIP (hipStreamCreateWithFlags (&l.st, hipStreamNonBlocking));
      for (int b = 0 ; b < 2 ; ++b)
        { MG_HIP (hipHostMalloc ((void **) &l.pin[b], MG_XFER_PIECE, hipHostMallocPortable));
          MG_HIP (hipEventCreateWithFlags (&l.ev[b], hipEventDisableTiming));
        }
    }
  return MG_OK;
}

static inline void mgXferApply (char *dst, const char *src, size_t bytes, int op)
{
  if (op == MG_XFER_COPY) { memcpy (dst, src, bytes); return; }
  U16 *d = (U16 *) dst; const U16 *s = (const U16 *) src;
  const size_t n = bytes >> 1;
  for (size_t i = 0 ; i < n ; ++i) { const U32 v = (U32) d[i] + s[i]; d[i] = (U16) (v > 0xffffu ? 0xffffu : v); }   /* (the compiler makes paddusw of it) */
}

/* thread t of T: pieces t, t + T, t + 2T, ... */
static hipError_t mgXferLaneRun (int dev, MgXferLane &l, int t, int T, char *dst, const char *src, size_t bytes, int op)
{
  hipError_t e = hipSetDevice (dev);
  if (e != hipSuccess) return e;
  const size_t nPieces = (bytes + MG_XFER_PIECE - 1) / MG_XFER_PIECE;
  size_t p = (size_t) t; int b = 0;
  auto issue = [&] (size_t piece, int buf) -> hipError_t
    { const size_t off = piece * MG_XFER_PIECE, len = bytes - off < MG_XFER_PIECE ? bytes - off : MG_XFER_PIECE;
      hipError_t x = hipMemcpyAsync (l.pin[buf], src + off, len, hipMemcpyDeviceToHost, l.st);
      return x != hipSuccess ? x : hipEventRecord (l.ev[buf], l.st);
    };
  if (p < nPieces && (e = issue (p, b)) != hipSuccess) return e;
  for ( ; p < nPieces ; p += (size_t) T, b ^= 1)
    { const size_t next = p + (size_t) T;
      if (next < nPieces && (e = issue (next, b ^ 1)) != hipSuccess) return e;      /* the other block fills while this one is emptied */
      if ((e = hipEventSynchronize (l.ev[b])) != hipSuccess) return e;
      const size_t off = p * MG_XFER_PIECE, len = bytes - off < MG_XFER_PIECE ? bytes - off : MG_XFER_PIECE;
      mgXferApply (dst + off, l.pin[b], len, op);
    }
  return hipSuccess;
}

MgStatus mgXferD2H (void *hostDst, const void *devSrc, size_t bytes, int op)
{
  if (!bytes) return MG_OK;
  std::lock_guard<std::mutex> g (gX.lock);
  const size_t nPieces = (bytes + MG_XFER_PIECE - 1) / MG_XFER_PIECE;
  int T = mgXferThreads ();
  if ((size_t) T > nPieces) T = (int) nPieces;
  MgStatus s = mgXferPrepareLocked (T); if (s) return s;
  const int dev = gX.dev;
  hipError_t err[MG_XFER_MAXT];
  for (int t = 0 ; t < T ; ++t) err[t] = hipSuccess;
  std::vector<std::thread> th;
  int started = 1;                                             /* lane 0 is the caller's */
  try { for (int t = 1 ; t < T ; ++t) { th.emplace_back ([&, t] { err[t] = mgXferLaneRun (dev, gX.lane[t], t, T, (char *) hostDst, (const char *) devSrc, bytes, op); }); ++started; } }
  catch (...) { }                                              /* the system gave fewer threads: the pieces of those that did not start are done below */
  err[0] = mgXferLaneRun (dev, gX.lane[0], 0, T, (char *) hostDst, (const char *) devSrc, bytes, op);
  for (auto &x : th) x.join ();
  for (int t = started ; t < T ; ++t) err[t] = mgXferLaneRun (dev, gX.lane[t], t, T, (char *) hostDst, (const char *) devSrc, bytes, op);
  for (int t = 0 ; t < T ; ++t) if (err[t] != hipSuccess) return mgHipFail (err[t], "mgXferD2H");
  return MG_OK;
}

/* The other direction, for the arrays a device table is (re)built from (ms->value, ms->depth): the team fills the
   page-locked blocks from the pageable source and the copy engine drains them. */
static hipError_t mgXferLaneUp (int dev, MgXferLane &l, int t, int T, char *dDst, const char *hSrc, size_t bytes)
{
  hipError_t e = hipSetDevice (dev);
  if (e != hipSuccess) return e;
  const size_t nPieces = (bytes + MG_XFER_PIECE - 1) / MG_XFER_PIECE;
  int b = 0; size_t done = 0;
  for (size_t p = (size_t) t ; p < nPieces ; p += (size_t) T, b ^= 1, ++done)
    { if (done >= 2 && (e = hipEventSynchronize (l.ev[b])) != hipSuccess) return e;      /* the copy that last read this block */
      const size_t off = p * MG_XFER_PIECE, len = bytes - off < MG_XFER_PIECE ? bytes - off : MG_XFER_PIECE;
      memcpy (l.pin[b], hSrc + off, len);
      if ((e = hipMemcpyAsync (dDst + off, l.pin[b], len, hipMemcpyHostToDevice, l.st)) != hipSuccess) return e;
      if ((e = hipEventRecord (l.ev[b], l.st)) != hipSuccess) return e;
    }
  return hipStreamSynchronize (l.st);
}

MgStatus mgXferH2D (void *devDst, const void *hostSrc, size_t bytes)
{
  if (!bytes) return MG_OK;
  std::lock_guard<std::mutex> g (gX.lock);
  const size_t nPieces = (bytes + MG_XFER_PIECE - 1) / MG_XFER_PIECE;
  int T = mgXferThreads ();
  if ((size_t) T > nPieces) T = (int) nPieces;
  MgStatus s = mgXferPrepareLocked (T); if (s) return s;
  const int dev = gX.dev;
  hipError_t err[MG_XFER_MAXT];
  for (int t = 0 ; t < T ; ++t) err[t] = hipSuccess;
  std::vector<std::thread> th;
  int started = 1;
  try { for (int t = 1 ; t < T ; ++t) { th.emplace_back ([&, t] { err[t] = mgXferLaneUp (dev, gX.lane[t], t, T, (char *) devDst, (const char *) hostSrc, bytes); }); ++started; } }
  catch (...) { }
  err[0] = mgXferLaneUp (dev, gX.lane[0], 0, T, (char *) devDst, (const char *) hostSrc, bytes);
  for (auto &x : th) x.join ();
  for (int t = started ; t < T ; ++t) err[t] = mgXferLaneUp (dev, gX.lane[t], t, T, (char *) devDst, (const char *) hostSrc, bytes);
  for (int t = 0 ; t < T ; ++t) if (err[t] != hipSuccess) return mgHipFail (err[t], "mgXferH2D");
  return MG_OK;
}
