/* mg_api.hip — host glue behind the C ABI of include/modgpu.h (layer 2, the batch/device path):
 * device bookkeeping, the per-Modset device table registry, and the composite entry points.
 * The reference-compatible scalar API (layer 1) lives in mg_host.c and calls the mgHook* functions
 * defined here to keep the host arrays and the device table coherent.
 */
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <unordered_map>
#include "mg_common.h"
#include "mg_internal.h"
#include "mg_xfer.h"

/* ---------------------------------------------------------------------------------------- */
/* errors / device                                                                            */

static thread_local char gErr[512] = "";

void mgSetError (const char *fmt, ...)
{ va_list ap; va_start (ap, fmt); vsnprintf (gErr, sizeof (gErr), fmt, ap); va_end (ap); }

extern "C" const char *mgLastError (void) { return gErr; }

MgStatus mgHipFail (hipError_t e, const char *what)
{ mgSetError ("HIP error %d (%s) in %s", (int) e, hipGetErrorString (e), what); return MG_ERR_HIP; }

extern "C" int mgDeviceCount (void)
{ int n = 0; if (hipGetDeviceCount (&n) != hipSuccess) { (void) hipGetLastError (); return 0; } return n; }

MgStatus mgEnsureDevice (void)
{
  static int known = -1;
  if (known < 0) known = mgDeviceCount ();
  if (known <= 0)
    { mgSetError ("no HIP device available: libmodgpu has no CPU fallback for the batch path");
      return MG_ERR_NO_DEVICE;
    }
  return MG_OK;
}

extern "C" MgStatus mgSetDevice (int device)
{ MgStatus s = mgEnsureDevice (); if (s) return s; MG_HIP (hipSetDevice (device)); return MG_OK; }

extern "C" MgStatus mgDeviceAlloc (void **dptr, size_t bytes)
{ MgStatus s = mgEnsureDevice (); if (s) return s; MG_HIP (hipMalloc (dptr, bytes ? bytes : 16)); return MG_OK; }
extern "C" MgStatus mgDeviceFree (void *dptr) { if (dptr) MG_HIP (hipFree (dptr)); return MG_OK; }
extern "C" MgStatus mgMemcpyH2D (void *dst, const void *src, size_t bytes, void *stream)
{ if (bytes) MG_HIP (hipMemcpyAsync (dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t) stream)); return MG_OK; }
extern "C" MgStatus mgMemcpyD2H (void *dst, const void *src, size_t bytes, void *stream)
{ if (bytes) MG_HIP (hipMemcpyAsync (dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t) stream));
  MG_HIP (hipStreamSynchronize ((hipStream_t) stream)); return MG_OK; }
/* page-locked host memory for the C callers' device-to-host copies (a copy into pageable memory goes through the runtime's staging
   buffers at a few GB/s); portable and mapped: the blocks are kept between calls, and the caller may have moved to another device */
extern "C" void *mgPinnedAlloc (size_t bytes)
{ void *p = 0; if (mgEnsureDevice () || hipHostMalloc (&p, bytes ? bytes : 16, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { (void) hipGetLastError (); return 0; } return p; }
extern "C" void mgPinnedFree (void *p) { if (p) (void) hipHostFree (p); }
/* device memory to a page-locked block by a kernel that writes host memory, then a wait for the stream: megabytes copied by the
   copy engine take their turn behind a text window (128 MiB) that is on its way to the device at the same time, stores from a kernel
   do not (tools/ubench_copyq.hip) */
__global__ void mgCopyOutKernel (const U64 *__restrict__ src, U64 *__restrict__ dstHost, U64 nWords)
{ for (U64 i = (U64) blockIdx.x * blockDim.x + threadIdx.x ; i < nWords ; i += (U64) gridDim.x * blockDim.x) dstHost[i] = src[i]; }
extern "C" MgStatus mgCopyOutPinned (void *dstPinned, const void *srcDev, size_t bytes, void *stream)
{
  if (!bytes) return MG_OK;
  void *dDst = 0;
  if ((bytes & 7) || ((uintptr_t) dstPinned & 7) || ((uintptr_t) srcDev & 7) || hipHostGetDevicePointer (&dDst, dstPinned, 0) != hipSuccess)
    { (void) hipGetLastError (); return mgMemcpyD2H (dstPinned, srcDev, bytes, stream); }
  const U64 nWords = bytes >> 3;
  unsigned grid = (unsigned) ((nWords + 255) / 256); if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL (mgCopyOutKernel, dim3 (grid), dim3 (256), 0, (hipStream_t) stream, (const U64 *) srcDev, (U64 *) dDst, nWords);
  MG_HIP (hipGetLastError ());
  MG_HIP (hipStreamSynchronize ((hipStream_t) stream));
  return MG_OK;
}
extern "C" MgStatus mgMemsetD (void *dst, int byte, size_t bytes, void *stream)
{ if (bytes) MG_HIP (hipMemsetAsync (dst, byte, bytes, (hipStream_t) stream)); return MG_OK; }
extern "C" MgStatus mgStreamSynchronize (void *stream)
{ MG_HIP (hipStreamSynchronize ((hipStream_t) stream)); return MG_OK; }

/* ---------------------------------------------------------------------------------------- */
/* per-kernel event timing (bench.py's roofline object)                                       */

static const char *gKernelNames[MG_K_COUNT] = {
  "mgPackKernel", "mgUnpackKernel", "mgTileInfoKernel", "mgScanKernel", "mgTableInsertKernel",
  "mgRankAssignKernel", "mgDirectFlagKernel", "mgTableFindKernel", "mgTableLoadKernel",
  "mgTableExportDepthKernel", "mgTableHistKernel", "mgReplayIndexKernel", "mgIndexFinishKernel",
  "mgSynthGenomeKernel", "mgSynthReadsKernel", "memset", "mgSegScanKernel", "mgSegCompactKernel",
  "mgPartChunks+ScanKernel", "mgPartHistKernel", "mgPartScatterKernel", "mgRankCountKernel", "mgRankScanKernel", "mgBucketDedupKernel",
  "mgBucketMergeKernel", "mgRankLookupKernel", "mgTableFindSegKernel", "mgHotPlan+ReduceKernel", "mgBucketFindKernel", "mgUnpartKernel", "mgChainKernel", "mgChainResolveKernel" };
#define MG_PROF_POOL 8192
struct MgProfRec { int id; hipEvent_t a, b; };
static struct {
  bool on = false;
  int only = -1;                 /* >= 0: bracket launches of this kernel id alone */
  MgProfRec rec[MG_PROF_POOL]; int used = 0, made = 0;
  double ms[MG_K_COUNT] = { 0 }; U64 n[MG_K_COUNT] = { 0 };
  int open = -1;
} gProf;

static void mgProfDrain (void)
{
  if (!gProf.used) return;
  (void) hipDeviceSynchronize ();
  for (int i = 0 ; i < gProf.used ; ++i)
    { float t = 0;
      if (hipEventElapsedTime (&t, gProf.rec[i].a, gProf.rec[i].b) == hipSuccess)
        { gProf.ms[gProf.rec[i].id] += t; ++gProf.n[gProf.rec[i].id]; }
    }
  gProf.used = 0;
}

void mgProfBegin (int id, hipStream_t st)
{
  if (!gProf.on || (gProf.only >= 0 && id != gProf.only)) return;
  if (gProf.used == MG_PROF_POOL) mgProfDrain ();
  MgProfRec &r = gProf.rec[gProf.used];
  if (gProf.used >= gProf.made)
    { if (hipEventCreate (&r.a) != hipSuccess || hipEventCreate (&r.b) != hipSuccess) { gProf.on = false; return; }
      gProf.made = gProf.used + 1;
    }
  r.id = id;
  (void) hipEventRecord (r.a, st);
  gProf.open = gProf.used;
}

void mgProfEnd (int id, hipStream_t st)
{
  if (!gProf.on || gProf.open < 0) return;
  (void) id;
  (void) hipEventRecord (gProf.rec[gProf.open].b, st);
  gProf.used = gProf.open + 1;
  gProf.open = -1;
}

extern "C" void mgProfileEnable (int on) { if (!on) mgProfDrain (); gProf.on = on != 0; }
extern "C" void mgProfileOnly (int kernelId) { mgProfDrain (); gProf.only = (kernelId >= 0 && kernelId < MG_K_COUNT) ? kernelId : -1; }
extern "C" void mgProfileReset (void)
{ mgProfDrain (); for (int i = 0 ; i < MG_K_COUNT ; ++i) { gProf.ms[i] = 0; gProf.n[i] = 0; } }
extern "C" int mgProfileKernels (void) { return MG_K_COUNT; }
extern "C" MgStatus mgProfileGet (int id, const char **name, double *totalMs, U64 *launches)
{
  if (id < 0 || id >= MG_K_COUNT) { mgSetError ("bad kernel id"); return MG_ERR_ARG; }
  mgProfDrain ();
  if (name) *name = gKernelNames[id];
  if (totalMs) *totalMs = gProf.ms[id];
  if (launches) *launches = gProf.n[id];
  return MG_OK;
}

/* ---------------------------------------------------------------------------------------- */
/* hash parameters                                                                            */

MgHashParams mgMakeParams (const Seqhash *sh)
{
  MgHashParams p;
  p.factor1 = sh->factor1; p.mask = sh->mask; p.k = sh->k; p.shift1 = sh->shift1;
  p.d = (U32) sh->w;
  p.dShift = __builtin_ctz (p.d);
  U64 odd = (U64) (p.d >> p.dShift);
  U64 x = odd;                                   /* Newton: inverse of an odd number mod 2^64 */
  for (int i = 0 ; i < 6 ; ++i) x *= 2 - odd * x;
  p.dOddInv = x;
  p.dOddLim = ~(U64) 0 / odd;
  p.small32 = (2 * sh->k <= 40 && odd < ((U64) 1 << 15)) ? 1u : 0u;
  p.c24 = (U32) (((U64) 1 << 24) % odd); p.inv32 = (U32) x; p.lim32 = (U32) (0xffffffffull / odd);
  return p;
}

static MgStatus mgCheckHasher (const Seqhash *sh)
{
  if (!sh || sh->k < 1 || sh->k >= 32 || sh->w < 1 || sh->shift1 != 64 - 2 * sh->k)
    { mgSetError ("invalid Seqhash (k %d w %d)", sh ? sh->k : -1, sh ? sh->w : -1); return MG_ERR_ARG; }
  return MG_OK;
}

/* ---------------------------------------------------------------------------------------- */
/* packing                                                                                    */

extern "C" size_t mgPackedWords (U64 nBases) { return (size_t) ((nBases + 15) / 16) + MG_PACK_PAD; }

/* mgPackHost, mgPackWords: mg_pack.c */
extern "C" void mgPackWords (const char *bases, U64 nBases, U32 *words);

extern "C" MgStatus mgPackDevice (const U8 *dBases, U64 nBases, U32 *dWords, void *stream)
{ MgStatus s = mgEnsureDevice (); if (s) return s; return mgLaunchPack (dBases, nBases, dWords, (hipStream_t) stream); }
extern "C" MgStatus mgUnpackDevice (const U32 *dWords, U64 nBases, U8 *dBases, void *stream)
{ MgStatus s = mgEnsureDevice (); if (s) return s; return mgLaunchUnpack (dWords, nBases, dBases, (hipStream_t) stream); }

/* ---------------------------------------------------------------------------------------- */
/* scan                                                                                       */

extern "C" MgStatus seqhashScanBatchDevice (const Seqhash *sh, const U32 *dPacked, U64 totalBases,
                                            const U64 *dReadOffsets, U32 nReads,
                                            U64 *dKmer, U32 *dPosF, U32 *dReadId, U64 capacity,
                                            U64 *dCount, void *dWork, void *stream)
{
  MgStatus s = mgEnsureDevice (); if (s) return s;
  if ((s = mgCheckHasher (sh))) return s;
  if (!dCount || (totalBases && (!dPacked || !dReadOffsets || !dWork)))
    { mgSetError ("seqhashScanBatchDevice: null argument"); return MG_ERR_ARG; }
  return mgLaunchScan (mgMakeParams (sh), dPacked, totalBases, dReadOffsets, nReads,
                       dKmer, dPosF, dReadId, capacity, dCount, dWork, (hipStream_t) stream);
}

/* grow-only device scratch */
struct MgArena {
  char *base = 0; size_t bytes = 0; size_t used = 0;
  MgStatus reserve (size_t need)
  { if (need <= bytes) return MG_OK;
    if (base) MG_HIP (hipFree (base));
    base = 0; bytes = 0;
    size_t want = need + need / 8 + (1 << 20);
    MG_HIP (hipMalloc ((void **) &base, want));
    bytes = want;
    return MG_OK;
  }
  void reset () { used = 0; }
  void *take (size_t n) { size_t a = (used + 255) & ~(size_t) 255; used = a + n; return base + a; }
  void release () { if (base) (void) hipFree (base); base = 0; bytes = used = 0; }
};
static inline size_t al256 (size_t n) { return (n + 255) & ~(size_t) 255; }

static U64 mgSurvivorGuess (const Seqhash *sh, U64 totalBases)
{
  U64 g = totalBases / (U64) sh->w;
  g += g / 4 + (1 << 16);
  return g < totalBases ? g : totalBases;
}

/* Host bytes (one base per byte) -> 2-bit packed words in HBM.  The bytes are packed ON THE HOST by a team of
 * threads (mg_pack.c: 128 bases per AVX2 step) into two pinned staging buffers, piece by piece, and each piece goes
 * across PCIe as packed words (a quarter of the bytes) with an asynchronous copy that overlaps the packing of the
 * next piece.  (Before: the bytes crossed as they were and were packed on the device: 1 byte per base on the link.) */
#include <pthread.h>
#include <atomic>
#include <thread>
#include <vector>
#include <unistd.h>
#define MG_UP_PIECE ((U64) 128 << 20)                 /* bases per piece (a multiple of 16): 32 MiB of packed words */
#define MG_MAXDEV 128            /* (an 8-GPU node in its 8-partition mode shows 64 devices) */
static struct MgUpStage { U32 *pin[2] = { 0, 0 }; hipEvent_t done[2]; bool ready = false; std::mutex lock; } gUps[MG_MAXDEV];   /* by device (the events belong to one): host threads that drive several GPUs do not take turns on one stage, nor re-make it at every switch */

static int mgHostThreads (void)
{
  static int n = 0;
  if (n) return n;
  const long kv = mgKnobs ()->packThreads;
  long v = kv != MG_KNOB_UNSET ? kv : 0;
  if (v <= 0) v = mgCpuBudget ();                                      /* (a cgroup CPU quota caps what threads can get) */
  if (kv == MG_KNOB_UNSET) v = v / 2;                                   /* measured (16-CPU quota): 8 packers keep the link busy, 16 starve the thread that issues the copies */
  if (v < 1) v = 1;
  if (v > 16) v = 16;
  return n = (int) v;
}

extern "C" MgStatus mgUploadPack (const char *bases, U64 nBases, U32 *dPacked, void *stream)
{
  MgStatus s = mgEnsureDevice (); if (s) return s;
  hipStream_t st = (hipStream_t) stream;
  if (!nBases) { MG_HIP (hipMemsetAsync (dPacked, 0, mgPackedWords (0) * 4, st)); return MG_OK; }
  int curDev = 0; MG_HIP (hipGetDevice (&curDev));
  if (curDev < 0 || curDev >= MG_MAXDEV) { mgSetError ("mgUploadPack: device number beyond %d", MG_MAXDEV - 1); return MG_ERR_ARG; }
  MgUpStage &gUp = gUps[curDev];
  std::lock_guard<std::mutex> g (gUp.lock);
  if (!gUp.ready)
    { for (int i = 0 ; i < 2 ; ++i)
        { if (!gUp.pin[i]) MG_HIP (hipHostMalloc ((void **) &gUp.pin[i], (MG_UP_PIECE / 16 + MG_PACK_PAD) * 4, hipHostMallocPortable));
          MG_HIP (hipEventCreateWithFlags (&gUp.done[i], hipEventDisableTiming));
        }
      gUp.ready = true;
    }
  const U64 nPieces = (nBases + MG_UP_PIECE - 1) / MG_UP_PIECE;
  int T = mgHostThreads ();
  if (nBases < ((U64) 1 << 20)) T = 1;
  pthread_barrier_t bar;
  /* every thread packs its share of every piece; thread 0 also waits for the staging buffer to be free before a
     piece and sends the piece off after it */
  volatile int failed = 0;
  auto work = [&] (int t)
    { for (U64 p = 0 ; p < nPieces ; ++p)
        { const U64 off = p * MG_UP_PIECE, len = nBases - off < MG_UP_PIECE ? nBases - off : MG_UP_PIECE;
          U32 *buf = gUp.pin[p & 1];
          if (t == 0 && p >= 2 && hipEventSynchronize (gUp.done[p & 1]) != hipSuccess) failed = 1;    /* the copy that last read this buffer */
          pthread_barrier_wait (&bar);
          const U64 words = (len + 15) / 16, per = ((words + T - 1) / T + 7) & ~(U64) 7;               /* whole words per thread, a multiple of 8 */
          const U64 w0 = per * (U64) t < words ? per * (U64) t : words, w1 = w0 + per < words ? w0 + per : words;
          if (w1 > w0)
            { const U64 b0 = w0 * 16, b1 = w1 * 16 < len ? w1 * 16 : len;
              mgPackWords (bases + off + b0, b1 - b0, buf + w0);
            }
          pthread_barrier_wait (&bar);
          if (t == 0)
            { U64 n = words;
              if (p + 1 == nPieces) { for (int j = 0 ; j < MG_PACK_PAD ; ++j) buf[words + j] = 0; n += MG_PACK_PAD; }   /* the pad words after the stream */
              if (hipMemcpyAsync (dPacked + off / 16, buf, n * 4, hipMemcpyHostToDevice, st) != hipSuccess
                  || hipEventRecord (gUp.done[p & 1], st) != hipSuccess) failed = 1;
            }
        }
    };
  /* the helpers are started first and held at a gate: if the system will not give T threads, those that did start are
     sent home and the caller packs alone (a thread constructor that throws must not leave threads at the barrier) */
  std::vector<std::thread> th;
  std::atomic<int> gate (0);
  auto helper = [&] (int t) { int v; while (!(v = gate.load (std::memory_order_acquire))) std::this_thread::yield (); if (v > 0) work (t); };
  try { for (int t = 1 ; t < T ; ++t) th.emplace_back (helper, t); }
  catch (...) { gate.store (-1, std::memory_order_release); for (auto &x : th) x.join (); th.clear (); T = 1; }
  pthread_barrier_init (&bar, 0, (unsigned) T);
  gate.store (1, std::memory_order_release);
  work (0);
  for (auto &x : th) x.join ();
  pthread_barrier_destroy (&bar);
  if (failed) return mgHipFail (hipGetLastError (), "mgUploadPack");
  MG_HIP (hipStreamSynchronize (st));
  return MG_OK;
}

extern "C" int64_t seqhashScanBatch (const Seqhash *sh, const char *bases, const int64_t *readOffsets, int nReads,
                                     U64 **kmerOut, int **posOut, bool **isFOut, int64_t **survStartOut)
{
  if (mgEnsureDevice () || mgCheckHasher (sh)) return -1;
  if (nReads < 0 || (nReads && (!readOffsets || readOffsets[0] != 0)))
    { mgSetError ("seqhashScanBatch: bad read offsets"); return -1; }
  U64 total = nReads ? (U64) readOffsets[nReads] : 0;
  size_t nw = mgPackedWords (total);
  U64 cap = mgSurvivorGuess (sh, total);
  MgArena ar;
  int64_t result = -1;
  U64 *hK = 0; U32 *hP = 0, *hR = 0;
  for (int attempt = 0 ; attempt < 3 ; ++attempt)
    { size_t need = al256 (nw * 4) + al256 (((size_t) nReads + 1) * 8) + al256 (cap * 8) + 2 * al256 (cap * 4)
                    + al256 (mgScanWorkBytes (total, (U32) nReads, cap)) + 4096;
      if (ar.reserve (need)) break;
      ar.reset ();
      U32 *dP = (U32 *) ar.take (nw * 4);
      U64 *dOff = (U64 *) ar.take (((size_t) nReads + 1) * 8);
      U64 *dK = (U64 *) ar.take (cap * 8);
      U32 *dPos = (U32 *) ar.take (cap * 4);
      U32 *dRid = (U32 *) ar.take (cap * 4);
      void *dWork = ar.take (mgScanWorkBytes (total, (U32) nReads, cap));
      U64 *dCount = (U64 *) ar.take (8 * MG_COUNT_WORDS);
      if (mgUploadPack (bases, total, dP, 0)) break;
      if (nReads && hipMemcpy (dOff, readOffsets, ((size_t) nReads + 1) * 8, hipMemcpyHostToDevice) != hipSuccess) break;
      if (seqhashScanBatchDevice (sh, dP, total, dOff, (U32) nReads, dK, dPos, dRid, cap, dCount, dWork, 0)) break;
      U64 cnt[MG_COUNT_WORDS];
      if (hipMemcpy (cnt, dCount, sizeof (cnt), hipMemcpyDeviceToHost) != hipSuccess) break;
      if (cnt[1] || cnt[0] > cap) { cap = cnt[3]; continue; }
      U64 n = cnt[0];
      hK = (U64 *) malloc ((n + 1) * 8); hP = (U32 *) malloc ((n + 1) * 4); hR = (U32 *) malloc ((n + 1) * 4);
      if (n)
        { if (hipMemcpy (hK, dK, n * 8, hipMemcpyDeviceToHost) != hipSuccess) break;
          if (hipMemcpy (hP, dPos, n * 4, hipMemcpyDeviceToHost) != hipSuccess) break;
          if (hipMemcpy (hR, dRid, n * 4, hipMemcpyDeviceToHost) != hipSuccess) break;
        }
      result = (int64_t) n;
      break;
    }
  ar.release ();
  if (result < 0)
    { if (!gErr[0]) mgSetError ("seqhashScanBatch: device failure");
      free (hK); free (hP); free (hR); return -1;
    }
  int64_t n = result;
  if (posOut) { int *p = (int *) malloc ((n + 1) * sizeof (int)); for (int64_t i = 0 ; i < n ; ++i) p[i] = (int) (hP[i] & MG_POS_MASK); *posOut = p; }
  if (isFOut) { bool *f = (bool *) malloc ((n + 1) * sizeof (bool)); for (int64_t i = 0 ; i < n ; ++i) f[i] = (hP[i] & MG_FWD_BIT) != 0; *isFOut = f; }
  if (survStartOut)
    { int64_t *st = (int64_t *) malloc (((size_t) nReads + 1) * sizeof (int64_t));
      int64_t i = 0;
      for (int r = 0 ; r <= nReads ; ++r) { while (i < n && (int64_t) hR[i] < r) ++i; st[r] = i; }
      *survStartOut = st;
    }
  if (kmerOut) *kmerOut = hK; else free (hK);
  free (hP); free (hR);
  return n;
}

/* ---------------------------------------------------------------------------------------- */
/* minimizers (seqhash.c:83-152) */

extern "C" MgStatus seqhashMinimizerBatchDevice (const Seqhash *sh, const U32 *dPacked, U64 totalBases,
                                                 const U64 *dReadOffsets, U32 nReads,
                                                 U64 *dHash, U32 *dPosF, U64 *dReadStart, U64 capacity,
                                                 U64 *nOut, void *stream)
{
  MgStatus s = mgEnsureDevice (); if (s) return s;
  if ((s = mgCheckHasher (sh))) return s;
  (void) totalBases;
  U64 total = 0;
  s = mgLaunchMinimizers (mgMakeParams (sh), (U32) sh->w, dPacked, dReadOffsets, nReads, dHash, dPosF, dReadStart,
                          capacity, &total, (hipStream_t) stream);
  if (nOut) *nOut = total;
  return s;
}

/* host buffers in, host arrays out; *startOut[r] .. [r+1] are read r's minimizers.  Returns their number, -1 on error. */
static int64_t mgMinimizersHost (const Seqhash *sh, const char *bases, const int64_t *readOffsets, int nReads,
                                 U64 **hashOut, U32 **posFOut, int64_t **startOut)
{
  if (mgEnsureDevice () || mgCheckHasher (sh)) return -1;
  if (nReads < 0 || (nReads && (!readOffsets || readOffsets[0] != 0))) { mgSetError ("minimizer batch: bad read offsets"); return -1; }
  const U64 total = nReads ? (U64) readOffsets[nReads] : 0;
  const size_t nw = mgPackedWords (total);
  U64 cap = total / ((U64) sh->w / 2 + 1) + total / 8 + (U64) nReads + 1024;
  int64_t result = -1;
  U64 *hH = 0; U32 *hP = 0; int64_t *hS = 0;
  MgArena ar;
  for (int attempt = 0 ; attempt < 2 ; ++attempt)
    { size_t need = al256 (nw * 4) + 2 * al256 (((size_t) nReads + 2) * 8) + al256 (cap * 8) + al256 (cap * 4) + 4096;
      if (ar.reserve (need)) break;
      ar.reset ();
      U32 *dP = (U32 *) ar.take (nw * 4);
      U64 *dOff = (U64 *) ar.take (((size_t) nReads + 2) * 8);
      U64 *dStart = (U64 *) ar.take (((size_t) nReads + 2) * 8);
      U64 *dH = (U64 *) ar.take (cap * 8);
      U32 *dQ = (U32 *) ar.take (cap * 4);
      if (mgUploadPack (bases, total, dP, 0)) break;
      if (nReads && hipMemcpy (dOff, readOffsets, ((size_t) nReads + 1) * 8, hipMemcpyHostToDevice) != hipSuccess) break;
      U64 n = 0;
      MgStatus s = seqhashMinimizerBatchDevice (sh, dP, total, dOff, (U32) nReads, dH, dQ, dStart, cap, &n, 0);
      if (s == MG_ERR_CAPACITY && attempt == 0) { cap = n; continue; }
      if (s) break;
      hH = (U64 *) malloc ((n + 1) * 8); hP = (U32 *) malloc ((n + 1) * 4); hS = (int64_t *) malloc (((size_t) nReads + 1) * 8);
      if (hipStreamSynchronize (0) != hipSuccess) break;
      if (n && (hipMemcpy (hH, dH, n * 8, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy (hP, dQ, n * 4, hipMemcpyDeviceToHost) != hipSuccess)) break;
      if (hipMemcpy (hS, dStart, ((size_t) nReads + 1) * 8, hipMemcpyDeviceToHost) != hipSuccess) break;
      result = (int64_t) n;
      break;
    }
  ar.release ();
  if (result < 0)
    { if (!gErr[0]) mgSetError ("minimizer batch: device failure");
      free (hH); free (hP); free (hS); return -1;
    }
  *hashOut = hH; *posFOut = hP; *startOut = hS;
  return result;
}

extern "C" int64_t seqhashMinimizerBatch (const Seqhash *sh, const char *bases, const int64_t *readOffsets, int nReads,
                                          U64 **hashOut, int **posOut, bool **isFOut, int64_t **startOut)
{
  U64 *hH = 0; U32 *hP = 0; int64_t *hS = 0;
  int64_t n = mgMinimizersHost (sh, bases, readOffsets, nReads, &hH, &hP, &hS);
  if (n < 0) return -1;
  if (posOut) { int *p = (int *) malloc ((n + 1) * sizeof (int)); for (int64_t i = 0 ; i < n ; ++i) p[i] = (int) (hP[i] & MG_POS_MASK); *posOut = p; }
  if (isFOut) { bool *f = (bool *) malloc ((n + 1) * sizeof (bool)); for (int64_t i = 0 ; i < n ; ++i) f[i] = (hP[i] & MG_FWD_BIT) != 0; *isFOut = f; }
  if (hashOut) *hashOut = hH; else free (hH);
  if (startOut) *startOut = hS; else free (hS);
  free (hP);
  return n;
}

/* one read for the iterator facade: *rec = malloc()ed {n hashes (U64), n posF (U32)} */
extern "C" int mgIterMinScan (Seqhash *sh, const char *s, int len, U64 **rec, U64 *nOut)
{
  *rec = 0; *nOut = 0;
  if (len < sh->k) return mgEnsureDevice () ? -1 : 0;
  int64_t off[2] = { 0, len };
  U64 *hH = 0; U32 *hP = 0; int64_t *hS = 0;
  int64_t n = mgMinimizersHost (sh, s, off, 1, &hH, &hP, &hS);
  if (n < 0) return -1;
  U64 *blk = (U64 *) malloc ((size_t) n * 12 + 16);
  memcpy (blk, hH, (size_t) n * 8);
  memcpy (blk + n, hP, (size_t) n * 4);
  free (hH); free (hP); free (hS);
  *rec = blk; *nOut = (U64) n;
  return 0;
}

/* ---------------------------------------------------------------------------------------- */
/* per-Modset device state                                                                    */

struct MgDev {
  MgTable t;
  MgArena arena;           /* the scratch of the call in flight (slot 0 of the query pipeline) */
  MgArena arena2;          /* slot 1: the scan of the NEXT query batch runs into it while the lookups of this one read slot 0's segments (mgQueryReadsDeviceAsync) */
  hipStream_t side = 0;    /* the stream those scans run on */
  hipEvent_t scanned[2] = { 0, 0 }, inputReady = 0;
  int ticketsOut = 0, nextSlot = 0; bool slotBusy[2] = { false, false };
  U64 *hPin;               /* pinned host words the per-call counters come back into: a device-to-host copy into pageable memory
                              goes through the runtime's staging path, whose wake-up was seen to take 10 - 25 ms now and then */
  U32 hostIndexMax;        /* entries 1..hostIndexMax are present in the host index[] table */
  bool built;
  int device = -1;         /* the GPU the table lives on: the one that was current when it was built */
};
/* the calling thread on a modset's own GPU for the length of a scope (the scalar API's hooks and the mirror may be reached from a thread
   that has since moved to another GPU with mgSetDevice) */
struct MgOnDevice
{ int before = -1; bool moved = false;
  explicit MgOnDevice (int dev) { if (dev >= 0 && hipGetDevice (&before) == hipSuccess && before != dev) moved = hipSetDevice (dev) == hipSuccess; }
  ~MgOnDevice () { if (moved) (void) hipSetDevice (before); }
};

static std::mutex gRegLock;
static std::unordered_map<const Modset *, MgDev *> gReg;
extern "C" { volatile int mgLiveDeviceModsets = 0; }

static MgDev *mgDevLookup (const Modset *ms)
{ std::lock_guard<std::mutex> g (gRegLock); auto it = gReg.find (ms); return it == gReg.end () ? 0 : it->second; }

static void mgDevFree (MgDev *d)
{
  if (!d) return;
  MgOnDevice here (d->device);
  if (d->hPin) (void) hipHostFree (d->hPin);
  if (d->built)
    { (void) hipFree (d->t.slots); (void) hipFree (d->t.value); (void) hipFree (d->t.occ); (void) hipFree (d->t.find8);
      (void) hipFree (d->t.baseDepth); (void) hipFree (d->t.counters); (void) hipFree (d->t.liveHist);
    }
  if (d->side) { (void) hipStreamSynchronize (d->side); (void) hipStreamDestroy (d->side); }      /* (a scan started by mgQueryReadsDeviceAsync and never waited for still writes into an arena) */
  d->arena.release (); d->arena2.release ();
  for (int i = 0 ; i < 2 ; ++i) if (d->scanned[i]) (void) hipEventDestroy (d->scanned[i]);
  if (d->inputReady) (void) hipEventDestroy (d->inputReady);
  delete d;
}

/* the load a table is brought to before lookups (a probe that misses walks to the next empty slot, and a workgroup of the partitioned
   lookups waits for its longest walk).  Rounds 1-5 asked for 60 and got 35-60: the slot count was rounded up to a power of two (config 3:
   0.35).  With the slot count free the figure is the load: 60 cost config 3's bucket lookups 0.99 -> 1.41 ms; 40 is what they had. */
#define MG_LOOKUP_LOAD_PCT 40
static MgStatus mgDevBuild (Modset *ms, MgDev *d, hipStream_t st)
{
  MgTable &t = d->t;
  if (ms->tableBits < 20 || ms->tableBits > 32)
    { mgSetError ("device modset supports table bits 20..32 (got %d)", ms->tableBits); return MG_ERR_ARG; }
  memset (&t, 0, sizeof (t));
  t.maxLog2Slots = ms->tableBits - 1;                 /* load <= 0.5 at the largest legal fill */
  t.kbits = 2 * ms->hasher->k;
  t.wantR = 4096;                                      /* 64 KiB of LDS per bucket, 1024-thread workgroups (measured best) */
  { const long r = mgKnobs ()->bucketR; if (r != MG_KNOB_UNSET && r >= 256) t.wantR = (U32) r; }     /* test knob */
  t.size = ms->size;
  t.pin = d->hPin + 24;                                /* (hPin: 32 words; 0-1 the add's counters, 8-23 the scans' counts) */
  U64 cap = (ms->tableSize >> 2);                      /* device arrays cover the largest legal size */
  MG_HIP (hipMalloc ((void **) &t.value, cap * sizeof (U64)));
  MG_HIP (hipMalloc ((void **) &t.baseDepth, cap * sizeof (U16)));
  MG_HIP (hipMalloc ((void **) &t.counters, 64));
  d->built = true;
  MG_HIP (hipMemsetAsync (t.baseDepth, 0, cap * sizeof (U16), st));
  MG_HIP (hipMemsetAsync (t.counters, 0, 64, st));
  { MgStatus s0 = mgTableEnsure (&t, ms->max, st); if (s0) return s0; }   /* slots sized to the content; grows on demand */
  t.max = 0; t.syncedMax = 0;
  t.baseZero = !ms->max;
  if (ms->max)
    { MG_HIP (hipStreamSynchronize (st));               /* (the memsets above: the team copies on streams of its own) */
      MgStatus s;
      if ((s = mgXferH2D (t.value, ms->value, ((size_t) ms->max + 1) * sizeof (U64)))) return s;
      if ((s = mgXferH2D (t.baseDepth, ms->depth, ((size_t) ms->max + 1) * sizeof (U16)))) return s;
      if ((s = mgTableLoadHost (&t, t.value, 1, ms->max, st))) return s;
      MG_HIP (hipStreamSynchronize (st));
      t.max = t.syncedMax = ms->max;
    }
  d->hostIndexMax = ms->max;
  return MG_OK;
}

/* get (and bring up to date) the device state of ms */
static MgStatus mgDevGet (Modset *ms, MgDev **out, hipStream_t st)
{
  MgStatus s = mgEnsureDevice (); if (s) return s;
  if (!ms || !ms->hasher) { mgSetError ("null Modset"); return MG_ERR_ARG; }
  if ((s = mgCheckHasher (ms->hasher))) return s;
  MgDev *d = mgDevLookup (ms);
  if (!d)
    { d = new MgDev (); d->built = false; d->hostIndexMax = 0; d->hPin = 0;
      if (hipGetDevice (&d->device) != hipSuccess) { delete d; return mgHipFail (hipGetLastError (), "hipGetDevice"); }
      if (hipHostMalloc ((void **) &d->hPin, 256, hipHostMallocDefault) != hipSuccess) { delete d; return mgHipFail (hipGetLastError (), "hipHostMalloc"); }
      if ((s = mgDevBuild (ms, d, st))) { mgDevFree (d); return s; }
      mgXferWarm ();                                       /* what is built on the device comes back through mg_xfer.hip: its streams are made meanwhile */
      std::lock_guard<std::mutex> g (gRegLock); gReg[ms] = d; mgLiveDeviceModsets = (int) gReg.size ();
    }
  else
    { int cur = -1;
      if (hipGetDevice (&cur) != hipSuccess || cur != d->device)
        { mgSetError ("this Modset's table lives on GPU %d, the calling thread is on GPU %d (mgSetDevice)", d->device, cur); return MG_ERR_ARG; }
    }
  if (d->built && ms->max > d->t.max)
    { /* the host appended entries through the scalar API (modsetIndexFind isAdd): mirror them */
      U32 first = d->t.max + 1, last = ms->max;
      if (last >= (ms->tableSize >> 2)) { mgSetError ("modset max %u beyond table capacity", last); return MG_ERR_CAPACITY; }
      MG_HIP (hipMemcpyAsync (d->t.value + first, ms->value + first, (size_t) (last - first + 1) * sizeof (U64), hipMemcpyHostToDevice, st));
      MG_HIP (hipMemcpyAsync (d->t.baseDepth + first, ms->depth + first, (size_t) (last - first + 1) * sizeof (U16), hipMemcpyHostToDevice, st));
      d->t.baseZero = false;
      if ((s = mgTableLoadHost (&d->t, d->t.value, first, last, st))) return s;
      MG_HIP (hipStreamSynchronize (st));
      d->t.max = d->t.syncedMax = last;
      if (d->hostIndexMax == first - 1) d->hostIndexMax = last;
    }
  d->t.size = ms->size;
  *out = d;
  return MG_OK;
}

extern "C" MgStatus mgModsetDeviceRelease (Modset *ms)
{
  MgDev *d = mgDevLookup (ms);
  if (!d) return MG_OK;
  MgStatus s = modsetSyncToHost (ms, 1);
  { std::lock_guard<std::mutex> g (gRegLock); gReg.erase (ms); mgLiveDeviceModsets = (int) gReg.size (); }
  mgDevFree (d);
  return s;
}

extern "C" void mgModsetHostChanged (Modset *ms)
{
  MgDev *d = mgDevLookup (ms);
  if (!d) return;
  { std::lock_guard<std::mutex> g (gRegLock); gReg.erase (ms); mgLiveDeviceModsets = (int) gReg.size (); }
  mgDevFree (d);
}

extern "C" MgStatus mgModsetClear (Modset *ms, void *stream)
{
  hipStream_t st = (hipStream_t) stream;
  MgDev *d = mgDevLookup (ms);
  if (d)
    { int cur = -1;
      if (hipGetDevice (&cur) != hipSuccess || cur != d->device)
        { mgSetError ("this Modset's table lives on GPU %d, the calling thread is on GPU %d (mgSetDevice)", d->device, cur); return MG_ERR_ARG; }
    }
  /* Host arrays only hold what the host was given: with a device table, entries beyond
     syncedMax exist on the device alone and index[] is populated up to hostIndexMax. */
  U32 hostIndexed = d ? d->hostIndexMax : ms->max;
  U32 hostUsed = d ? d->t.syncedMax : ms->max;
  if (d)
    { MgTable &t = d->t;
      mgTableForget (&t, st);          /* buckets are re-initialised by whoever writes them next (no 8 GB memset) */
      if (!t.baseZero) MG_HIP (hipMemsetAsync (t.baseDepth, 0, ((size_t) (t.max > t.syncedMax ? t.max : t.syncedMax) + 1) * sizeof (U16), st));
      t.baseZero = true;
      t.max = t.syncedMax = 0;
      t.pendingDepth = false;
      d->hostIndexMax = 0;
    }
  if (hostIndexed) memset (ms->index, 0, ms->tableSize * sizeof (U32));
  if (hostUsed)
    { memset (ms->depth, 0, ((size_t) hostUsed + 1) * sizeof (U16));
      memset (ms->info, 0, (size_t) hostUsed + 1);
    }
  ms->max = 0;
  return MG_OK;
}

/* device forms of the whole-set passes, used when the modset lives on the device ------------- */
static MgStatus mgAddBatch (Modset *ms, MgDev *d, const U64 *dKmer, U64 n, U32 *dIndexOut, int withDepth,
                            bool arenaLive, hipStream_t st, const MgHistReq *counted = 0, const MgSegSrc *segSrc = 0);

/* fold the pending device counts into baseDepth (afterwards baseDepth[i] IS depth[i]).  The callers that go on to REWRITE the host's depth[]
   wholesale from baseDepth (merge, prune) pass no `host`; one that leaves the host arrays alone (mgHookDeviceView) passes the Modset, and
   the pending counts reach its depth[] as modsetSyncToHost would have brought them -- the fold clears pendingDepth, so no later sync would
   (a sender of a rank-order merge was left with stale host depths) */
static MgStatus mgFoldCounts (MgDev *d, hipStream_t st, Modset *host = 0)
{
  MgTable &t = d->t;
  if (!t.max) return MG_OK;
  if (!t.pendingDepth) return MG_OK;                   /* nothing has counted since the last fold */
  MgStatus s = d->arena.reserve (al256 ((size_t) t.max * sizeof (U16)) + 4096); if (s) return s;      /* (no call is using the arena while a whole-set pass runs) */
  d->arena.reset ();
  U16 *dDelta = (U16 *) d->arena.take ((size_t) t.max * sizeof (U16));
  s = mgTableExportDepth (&t, dDelta, st);
  MG_HIP (hipStreamSynchronize (st));
  if (!s && host) s = mgXferD2H (host->depth + 1, dDelta, (size_t) t.max * sizeof (U16), MG_XFER_SATADD16);      /* modutils.c:26, `pending` times */
  return s;
}

/* modset.c:106-128 with ms1 on the device.  The caller (mg_host.c) has checked the hashers and
 * regrown ms1's host arrays; ms2's host arrays are current.  Returns 0 on success. */
/* ms2's entries 1 .. n2 as device arrays (entry i at [i - 1]) merged into ms1, which is on the device */
static int mgMergeDeviceCore (Modset *ms1, MgDev *d, const U64 *dV2, const U16 *dD2, const U8 *dI2, U32 n2, hipStream_t st)
{
  MgTable &t = d->t;
  const U32 max1 = t.max;
  U8 *dI1 = 0; U32 *dIdx = 0;
  int rc = -1;
  const size_t cap1 = (size_t) max1 + n2 + 2;
  do {
    if (hipMalloc ((void **) &dI1, cap1) || hipMalloc ((void **) &dIdx, (size_t) n2 * 4)) break;
    if (hipMemset (dI1, 0, cap1) || hipDeviceSynchronize ()) break;
    if (mgXferH2D (dI1, ms1->info, (size_t) max1 + 1)) break;
    /* ms2's values in ms2 index order: existing ones are found, new ones get max1+1, max1+2, ... (modset.c:120) */
    MgStatus as = mgAddBatch (ms1, d, dV2, n2, dIdx, 0, false, st);
    if (as == MG_ERR_CAPACITY) { fprintf (stderr, "FATAL ERROR: %s\n", mgLastError ()); exit (-1); }
    if (as) break;
    t.baseZero = false; t.liveHistValid = false;
    if (mgTableMergeApply (dIdx, dD2, dI2, n2, t.baseDepth, dI1, st)) break;
    if (hipStreamSynchronize (st)) break;
    /* bring the host mirror up to date wholesale: values of the new entries, all depths and info */
    if (t.max > t.syncedMax)
      { if (mgXferD2H (ms1->value + t.syncedMax + 1, t.value + t.syncedMax + 1, (size_t) (t.max - t.syncedMax) * 8, MG_XFER_COPY)) break;
        t.syncedMax = t.max;
      }
    if (mgXferD2H (ms1->depth, t.baseDepth, ((size_t) t.max + 1) * 2, MG_XFER_COPY)) break;
    if (mgXferD2H (ms1->info, dI1, (size_t) t.max + 1, MG_XFER_COPY)) break;
    ms1->depth[0] = 0;
    ms1->max = t.max;
    rc = 0;
  } while (0);
  if (rc && !gErr[0]) mgSetError ("device merge failed");
  (void) hipFree (dI1); (void) hipFree (dIdx);
  return rc;
}

extern "C" int mgHookMergeDevice (Modset *ms1, Modset *ms2)
{
  hipStream_t st = 0;
  MgDev *d; if (mgDevGet (ms1, &d, st)) return -1;
  const U32 n2 = ms2->max;
  if (mgFoldCounts (d, st)) return -1;
  if (!n2) return 0;
  U64 *dV2 = 0; U16 *dD2 = 0; U8 *dI2 = 0;
  int rc = -1;
  do {
    if (hipMalloc ((void **) &dV2, (size_t) n2 * 8) || hipMalloc ((void **) &dD2, (size_t) n2 * 2) || hipMalloc ((void **) &dI2, n2)) break;
    if (mgXferH2D (dV2, ms2->value + 1, (size_t) n2 * 8) || mgXferH2D (dD2, ms2->depth + 1, (size_t) n2 * 2) || mgXferH2D (dI2, ms2->info + 1, n2)) break;
    rc = mgMergeDeviceCore (ms1, d, dV2, dD2, dI2, n2, st);
  } while (0);
  if (rc && !gErr[0]) mgSetError ("device merge failed");
  (void) hipFree (dV2); (void) hipFree (dD2); (void) hipFree (dI2);
  return rc;
}

/* the same with the second set's arrays already in device memory (what a rank receives from its peers over xGMI, mg_comm.hip):
   nothing of the second set crosses the host link */
extern "C" int mgHookMergeDeviceArrays (Modset *ms1, const U64 *dValue2, const U16 *dDepth2, const U8 *dInfo2, U32 n2)
{
  hipStream_t st = 0;
  MgDev *d; if (mgDevGet (ms1, &d, st)) return -1;
  if (mgFoldCounts (d, st)) return -1;
  if (!n2) return 0;
  return mgMergeDeviceCore (ms1, d, dValue2, dDepth2, dInfo2, n2, st);
}

/* modset.c:64-77 with ms on the device: survivors keep their order, the table is rebuilt from them */
extern "C" int mgHookPruneDevice (Modset *ms, int lo, int hi)
{
  hipStream_t st = 0;
  MgDev *d; if (mgDevGet (ms, &d, st)) return -1;
  MgTable &t = d->t;
  const U32 n = t.max;
  if (mgFoldCounts (d, st)) return -1;
  /* values of device-only entries must reach the host before the arrays are rewritten?  No: the
     survivors are compacted on the device and copied back as a whole. */
  U8 *dInfo = 0, *dNewInfo = 0; U64 *dNewValue = 0; U16 *dNewDepth = 0; void *scratch = 0;
  int rc = -1;
  do {
    if (hipMalloc ((void **) &dInfo, (size_t) n + 1) || hipMalloc ((void **) &dNewInfo, (size_t) n + 2) || hipMalloc ((void **) &dNewValue, ((size_t) n + 2) * 8)
        || hipMalloc ((void **) &dNewDepth, ((size_t) n + 2) * 2) || hipMalloc (&scratch, mgTablePruneScratchBytes (n ? n : 1))) break;
    if (mgXferH2D (dInfo, ms->info, (size_t) n + 1)) break;
    if (mgTablePrune (&t, dInfo, lo, hi, dNewValue, dNewDepth, dNewInfo, scratch, st)) break;
    U64 c[2];
    if (hipMemcpyAsync (c, t.counters, 16, hipMemcpyDeviceToHost, st) || hipStreamSynchronize (st)) break;
    const U32 m = (U32) c[0];
    /* new arrays replace the old ones on both sides */
    if (m)
      { t.baseZero = false; t.liveHistValid = false;
        if (hipMemcpy (t.value + 1, dNewValue + 1, (size_t) m * 8, hipMemcpyDeviceToDevice) || hipMemcpy (t.baseDepth + 1, dNewDepth + 1, (size_t) m * 2, hipMemcpyDeviceToDevice)
            || mgXferD2H (ms->value + 1, dNewValue + 1, (size_t) m * 8, MG_XFER_COPY) || mgXferD2H (ms->depth + 1, dNewDepth + 1, (size_t) m * 2, MG_XFER_COPY)
            || mgXferD2H (ms->info + 1, dNewInfo + 1, m, MG_XFER_COPY)) break;
      }
    if (n > m && hipMemset (t.baseDepth + m + 1, 0, (size_t) (n - m) * 2)) break;
    mgTableForget (&t, st);
    t.max = 0;
    if (mgTableEnsure (&t, m, st)) break;
    if (m && mgTableLoadHost (&t, t.value, 1, m, st)) break;
    if (hipStreamSynchronize (st)) break;
    t.max = t.syncedMax = m;
    ms->max = m;
    d->hostIndexMax = 0;                 /* index[] is rebuilt (replayed) when somebody needs it */
    rc = 0;
  } while (0);
  if (rc && !gErr[0]) mgSetError ("device prune failed");
  (void) hipFree (dInfo); (void) hipFree (dNewInfo); (void) hipFree (dNewValue); (void) hipFree (dNewDepth); (void) hipFree (scratch);
  return rc;
}

/* hooks for mg_host.c */
extern "C" void mgHookDestroy (Modset *ms) { mgModsetHostChanged (ms); }
extern "C" void mgHookHostRewrote (Modset *ms) { mgModsetHostChanged (ms); }
extern "C" void mgHookNeedHostAll (Modset *ms, int wantIndex)
{
  MgDev *d = mgDevLookup (ms);
  if (!d) return;
  if (modsetSyncToHost (ms, wantIndex) != MG_OK)
    { fprintf (stderr, "FATAL ERROR: %s\n", mgLastError ()); exit (-1); }
}
extern "C" void mgHookNeedHost (Modset *ms, int wantIndex)
{
  MgDev *d = mgDevLookup (ms);
  if (!d) return;
  /* current when no entry, no index slot and no depth count is pending on the device (a batch that only re-hits
     existing k-mers leaves max alone but not depth[], which callers read and bump directly: modutils.c:26) */
  if (!d->t.pendingDepth && d->t.syncedMax == d->t.max && (!wantIndex || d->hostIndexMax >= d->t.max)) return;
  mgHookNeedHostAll (ms, wantIndex);
}
extern "C" int mgHookHasDevice (Modset *ms) { return mgDevLookup (ms) != 0; }

/* ---------------------------------------------------------------------------------------- */
/* batch insert / find                                                                        */

/* modimizers per insert pass (tokens are 31-bit ordinals); MODGPU_ADD_CHUNK shrinks it for tests */
static U64 mgAddChunkSize (void)
{
  const long c = mgKnobs ()->addChunk;
  U64 v = c != MG_KNOB_UNSET && c > 0 ? (U64) c : ((U64) 1 << 30);
  if (v > ((U64) 1 << 30)) v = (U64) 1 << 30;
  return v;
}
#define MG_ADD_CHUNK (mgAddChunkSize ())

static MgStatus mgAddChunk (Modset *ms, MgDev *d, const U64 *dKmer, U64 n, U32 *dIndexOut, int withDepth,
                            void *scratch, hipStream_t st, const MgHistReq *counted = 0, const MgSegSrc *segSrc = 0)
{
  MgTable &t = d->t;
  MgStatus s;
  MG_HIP (hipMemsetAsync (t.counters, 0, 16, st));
  if ((s = mgTableAdd (&t, dKmer, n, withDepth, scratch, st, counted, segSrc))) return s;
  volatile U64 *c = d->hPin;
  MG_HIP (hipMemcpyAsync (d->hPin, t.counters, 16, hipMemcpyDeviceToHost, st));
  MG_HIP (hipStreamSynchronize (st));
  U64 newMax = (U64) t.max + c[0];
  if (c[1] || newMax >= t.size)
    { /* modset.c:58 */
      mgSetError ("hashTableSize %u is too small for %llu", t.size, (unsigned long long) newMax);
      return MG_ERR_CAPACITY;
    }
  t.max = (U32) newMax;
  ms->max = t.max;
  if (n >= 4096) t.newPct = (int) (c[0] * 100 / n);
  if (dIndexOut && (s = mgTableFind (&t, dKmer, n, dIndexOut, st))) return s;
  return MG_OK;
}

/* arenaLive: dKmer itself lives in d->arena (mgAddReadsDevice), so the arena must not be reset
 * or reallocated; the caller reserved room for the temporaries taken here. */
static MgStatus mgAddBatch (Modset *ms, MgDev *d, const U64 *dKmer, U64 n, U32 *dIndexOut, int withDepth,
                            bool arenaLive, hipStream_t st, const MgHistReq *counted, const MgSegSrc *segSrc)
{
  if (!n) return MG_OK;
  if (d->ticketsOut) { mgSetError ("a query batch of this modset is in flight (mgQueryReadsDeviceAsync without its mgQueryReadsDeviceWait)"); return MG_ERR_ARG; }
  U64 chunk = n < MG_ADD_CHUNK ? n : MG_ADD_CHUNK;
  MgStatus s = mgTableEnsure (&d->t, chunk, st); if (s) return s;
  size_t need = mgTableAddScratchBytes (&d->t, chunk) + 4096;
  if (!arenaLive)
    { if ((s = d->arena.reserve (need))) return s;
      d->arena.reset ();
    }
  else if (d->arena.bytes - d->arena.used < need)
    { mgSetError ("internal: arena too small for insert temporaries"); return MG_ERR_NOMEM; }
  void *scratch = d->arena.take (mgTableAddScratchBytes (&d->t, chunk));
  for (U64 off = 0 ; off < n && !s ; off += chunk)
    { U64 m = n - off < chunk ? n - off : chunk;
      if (off) s = mgTableEnsure (&d->t, m, st);
      if (!s) s = mgAddChunk (ms, d, dKmer + off, m, dIndexOut ? dIndexOut + off : 0, withDepth, scratch, st,
                              (counted && n <= chunk) ? counted : 0,       /* the counts cover the whole batch */
                              n <= chunk ? segSrc : 0);
    }
  return s;
}

extern "C" MgStatus modsetAddBatchDevice (Modset *ms, const U64 *dKmer, U64 n, U32 *dIndexOut, int withDepth, void *stream)
{
  hipStream_t st = (hipStream_t) stream;
  MgDev *d; MgStatus s = mgDevGet (ms, &d, st); if (s) return s;
  return mgAddBatch (ms, d, dKmer, n, dIndexOut, withDepth, false, st);
}

extern "C" MgStatus modsetFindBatchDevice (Modset *ms, const U64 *dKmer, U64 n, U32 *dIndexOut, void *stream)
{
  hipStream_t st = (hipStream_t) stream;
  MgDev *d; MgStatus s = mgDevGet (ms, &d, st); if (s) return s;
  d->t.loadPct = MG_LOOKUP_LOAD_PCT;
  if (d->t.slots && (s = mgTableEnsure (&d->t, 0, st))) return s;
  return mgTableFind (&d->t, dKmer, n, dIndexOut, st);
}

/* What the device holds and the host arrays do not yet: value[] of the entries past syncedMax, the pending depth counts, the
 * index[] layout.  The callers read those arrays themselves (modset.h:17-28; modutils.c:26,69,186-198; modset.c:79-88), so this IS
 * the last step of the path for them.  Everything moves through mg_xfer.hip's team (page-locked pieces, several copy streams, the
 * final memcpy / saturating add spread over the host's threads); the device-side temporaries (the 16-bit counts, the replayed
 * index[]) come out of the modset's grow-only arena, which no call is using while this one runs: nothing is allocated per call.
 * The host's depth[] stays the authority (a caller may have bumped it directly, modutils.c:26): pending counts are ADDED to it,
 * saturating -- the thread that empties a page-locked piece does that instead of a memcpy, at the same memory traffic. */
extern "C" MgStatus modsetSyncToHost (Modset *ms, int wantIndex)
{
  MgDev *d = mgDevLookup (ms);
  if (!d) return MG_OK;
  MgOnDevice here (d->device);
  MgTable &t = d->t;
  hipStream_t st = 0;
  MgStatus s;
  MG_HIP (hipDeviceSynchronize ());
  if (d->ticketsOut && (t.pendingDepth || (wantIndex && d->hostIndexMax < t.max)))
    { mgSetError ("a query batch of this modset is in flight (its scratch is what a sync would use)"); return MG_ERR_ARG; }
  if (t.max > t.syncedMax)
    { const U32 first = t.syncedMax + 1;
      if ((s = mgXferD2H (ms->value + first, t.value + first, (size_t) (t.max - first + 1) * sizeof (U64), MG_XFER_COPY))) return s;
      t.syncedMax = t.max;
    }
  ms->max = t.max;
  if (t.max && t.pendingDepth)
    { /* depth[i] = min (65535, depth[i] + pending)   (modutils.c:26 applied `pending` times) */
      if ((s = d->arena.reserve (al256 ((size_t) t.max * sizeof (U16)) + 4096))) return s;
      d->arena.reset ();
      U16 *dDelta = (U16 *) d->arena.take ((size_t) t.max * sizeof (U16));
      if ((s = mgTableExportDepth (&t, dDelta, st))) return s;
      MG_HIP (hipStreamSynchronize (st));
      if ((s = mgXferD2H (ms->depth + 1, dDelta, (size_t) t.max * sizeof (U16), MG_XFER_SATADD16))) return s;
    }
  if (wantIndex && d->hostIndexMax < t.max)
    { if ((s = d->arena.reserve (al256 (ms->tableSize * sizeof (U32)) + 4096))) return s;
      d->arena.reset ();
      U32 *dIndex = (U32 *) d->arena.take (ms->tableSize * sizeof (U32));
      if ((s = mgTableReplayIndex (&t, mgMakeParams (ms->hasher), ms->tableBits, dIndex, st))) return s;
      MG_HIP (hipStreamSynchronize (st));
      if ((s = mgXferD2H (ms->index, dIndex, ms->tableSize * sizeof (U32), MG_XFER_COPY))) return s;
      d->hostIndexMax = t.max;
    }
  return MG_OK;
}

/* depth[] of ms has been remade from what the device counted elsewhere (modasm's read ingest, mg_refpack.hip): the device table's own copy
   follows -- dDepth16[0 .. max] on the device -- instead of the whole table being dropped and rebuilt from the host arrays on its next use.
   No count is pending in the table (the caller synced before it started).  0 if there is no device table (nothing to do). */
extern "C" MgStatus mgModsetAdoptDepthDevice (Modset *ms, const U16 *dDepth16)
{
  MgDev *d = mgDevLookup (ms);
  if (!d || !d->built) return MG_OK;
  MgTable &t = d->t;
  if (t.pendingDepth || t.max != ms->max) { mgModsetHostChanged (ms); return MG_OK; }      /* (not in step with the host: start again from its arrays) */
  MG_HIP (hipMemcpy (t.baseDepth, dDepth16, ((size_t) t.max + 1) * sizeof (U16), hipMemcpyDeviceToDevice));
  t.baseZero = false; t.liveHistValid = false;
  return MG_OK;
}

/* ms's entries 1 .. max where the device table keeps them (value, depth with every pending count folded in), for a sender that wants
   them on the device anyway (mg_comm.hip); -1: no device table */
extern "C" int mgHookDeviceView (Modset *ms, const U64 **dValue1, const U16 **dDepth1, U32 *max)
{
  MgDev *d = mgDevLookup (ms);
  if (!d || !d->built || d->ticketsOut) return -1;
  { int cur = -1; if (hipGetDevice (&cur) != hipSuccess || cur != d->device) return -1; }      /* (a sender stages from the host then) */
  if (hipDeviceSynchronize () != hipSuccess || mgFoldCounts (d, 0, ms)) return -1;
  *dValue1 = d->t.value + 1; *dDepth1 = d->t.baseDepth + 1; *max = d->t.max;
  return 0;
}

extern "C" U64 mgModsetDeviceSlots (Modset *ms) { MgDev *d = mgDevLookup (ms); return d && d->built ? d->t.nSlots : 0; }

extern "C" MgStatus modsetDepthHistogramDevice (Modset *ms, U64 *dHist, void *stream)
{
  hipStream_t st = (hipStream_t) stream;
  MgDev *d; MgStatus s = mgDevGet (ms, &d, st); if (s) return s;
  return mgTableHistogram (&d->t, dHist, st);
}

/* ---------------------------------------------------------------------------------------- */
/* composite: scan a device-resident batch straight into the modset                           */

struct MgScanBufs { U64 *kmer; U32 *posF; U32 *rid; void *work; U64 *count; U64 cap; MgHistReq counted;
                    MgSegSrc seg; bool lazy; void *findScratch, *findScratch2; };      /* lazy: kmer[] has not been written, the modimizers are in seg */

/* scan into arena buffers, growing once if the survivor guess was too small */
/* outPosF / outRid (with room for outCap entries): the caller's own arrays; when they are large enough for the scan's
 * capacity the compaction writes pos / read straight into them (a device-to-device copy of 1.2 GB per 10 Gbp batch
 * took longer than the scan itself) */
struct MgScanReq { const Seqhash *sh; const U32 *dPacked; U64 totalBases; const U64 *dReadOffsets; U32 nReads; bool wantPos; size_t extraPerSurvivor;
                   U32 *outPosF, *outRid; U64 outCap; bool lazy; };
static inline MgArena &mgArenaOf (MgDev *d, int slot) { return slot ? d->arena2 : d->arena; }
static inline U64 *mgCountPin (MgDev *d, int slot) { return d->hPin + 8 + 8 * slot; }

/* launch: buffers out of the slot's arena for `cap` modimizers, the scan, its counters on their way to the slot's pinned words.  No
   wait: mgScanFinish does that */
static MgStatus mgScanStart (MgDev *d, int slot, const MgScanReq &q, U64 cap, MgScanBufs *b, hipStream_t st)
{
  b->lazy = false; b->seg.nSegs = 0; b->seg.segKmer = 0; b->findScratch = 0; b->findScratch2 = 0;
  MgArena &ar = mgArenaOf (d, slot);
  /* a lookup batch whose k-mers stay in the segments may take the partitioned path (mgTableFindPartitioned): it needs the first digit's counts */
  const long fp = mgKnobs ()->findPath;
  const bool lookupHist = q.lazy && q.wantPos && !q.extraPerSurvivor && d->t.slots && fp != 'd'
                          && (fp == 'p' || fp == '2' || (cap >= ((U64) 1 << 24) && d->t.nSlots >= ((U64) 1 << 24) && d->t.log2NB > 9));
  const bool lookup2 = lookupHist && fp != 'p';
  MgHashParams p = mgMakeParams (q.sh);
  const size_t perS = 8 + (q.wantPos ? 8 : 0) + q.extraPerSurvivor;
  const size_t need = al256 (cap * perS) + 4 * 4096 + 512 * MG_HIST_STRIDE * 4 + al256 (mgScanWorkBytes (q.totalBases, q.nReads, cap))
                      + (q.extraPerSurvivor ? mgTableAddScratchBytes (&d->t, cap < MG_ADD_CHUNK ? cap : MG_ADD_CHUNK) + 8192 : 0) + 8 * 256
                      + (lookupHist ? mgTableFindPartScratchBytes (cap) + 8192 : 0) + (lookup2 ? mgTableFindPart2ScratchBytes (cap) + 8192 : 0);
  MgStatus s = ar.reserve (need); if (s) return s;
  ar.reset ();
  b->cap = cap;
  b->kmer = (U64 *) ar.take (cap * 8);
  const bool direct = q.wantPos && q.outPosF && q.outRid && q.outCap >= cap;
  b->posF = !q.wantPos ? 0 : (direct ? q.outPosF : (U32 *) ar.take (cap * 4));
  b->rid = !q.wantPos ? 0 : (direct ? q.outRid : (U32 *) ar.take (cap * 4));
  b->work = ar.take (mgScanWorkBytes (q.totalBases, q.nReads, cap));
  b->count = (U64 *) ar.take (8 * MG_COUNT_WORDS);
  b->counted.log2NB = 0; b->counted.kbits = 64; b->counted.binCount = 0; b->counted.hiB = 0;
  if (q.extraPerSurvivor || lookupHist)              /* the survivors go into the modset, or through the partitioned lookup: have the scan count the first partition digit */
    { b->counted.binCount = (U32 *) ar.take (512 * MG_HIST_STRIDE * sizeof (U32));
      b->counted.log2NB = d->t.log2NB; b->counted.kbits = d->t.kbits;
      if (lookupHist) b->counted.hiB = mgTableFindDigitBits (&d->t);
    }
  b->findScratch = lookupHist ? ar.take (mgTableFindPartScratchBytes (cap)) : 0;
  b->findScratch2 = lookup2 ? ar.take (mgTableFindPart2ScratchBytes (cap)) : 0;
  const bool lz = q.lazy && (q.wantPos ? q.extraPerSurvivor == 0 : b->counted.binCount != 0);   /* a build needs the digit counts; a pure lookup just the segments */
  if ((s = mgLaunchScan (p, q.dPacked, q.totalBases, q.dReadOffsets, q.nReads, b->kmer, b->posF, b->rid, cap, b->count, b->work, st,
                         b->counted.binCount ? &b->counted : 0, lz ? &b->seg : 0))) return s;
  b->lazy = lz;
  MG_HIP (hipMemcpyAsync (mgCountPin (d, slot), b->count, MG_COUNT_WORDS * sizeof (U64), hipMemcpyDeviceToHost, st));
  return MG_OK;
}
/* wait for the scan; *nOut = its modimizers, or *retryCap != 0: the capacity was too small, start again with that one */
static MgStatus mgScanFinish (MgDev *d, int slot, const MgScanBufs *b, U64 *nOut, U64 *retryCap, hipStream_t st, hipEvent_t done = 0)
{
  volatile U64 *c = mgCountPin (d, slot);
  if (done) MG_HIP (hipEventSynchronize (done));           /* (this scan alone: the stream may hold the next batch's scan behind it) */
  else MG_HIP (hipStreamSynchronize (st));
  *retryCap = 0;
  if (!c[1] && c[0] <= b->cap) { *nOut = c[0]; return MG_OK; }
  *nOut = c[0]; *retryCap = c[3];
  return MG_OK;
}

static MgStatus mgScanIntoArena (MgDev *d, const Seqhash *sh, const U32 *dPacked, U64 totalBases,
                                 const U64 *dReadOffsets, U32 nReads, bool wantPos, size_t extraPerSurvivor,
                                 MgScanBufs *b, U64 *nOut, hipStream_t st,
                                 U32 *outPosF = 0, U32 *outRid = 0, U64 outCap = 0, bool lazy = false)
{
  if (d->ticketsOut) { mgSetError ("a query batch of this modset is in flight (mgQueryReadsDeviceAsync without its mgQueryReadsDeviceWait)"); return MG_ERR_ARG; }
  MgScanReq q = { sh, dPacked, totalBases, dReadOffsets, nReads, wantPos, extraPerSurvivor, outPosF, outRid, outCap, lazy };
  U64 cap = mgSurvivorGuess (sh, totalBases);
  if (extraPerSurvivor)
    { /* size the table now for the expected number of modimizers (N/d): the insert would do it anyway once the
         count is known, and with the geometry fixed the scan (or, for 2k < 24, the compaction kernel) can count for the first partition pass */
      U64 expect = totalBases / (U64) (sh->w > 0 ? sh->w : 1) + 1;
      if (expect > MG_ADD_CHUNK) expect = MG_ADD_CHUNK;
      MgStatus es = mgTableEnsure (&d->t, expect, st); if (es) return es;
    }
  for (int attempt = 0 ; attempt < 3 ; ++attempt)
    { MgStatus s = mgScanStart (d, 0, q, cap, b, st); if (s) return s;
      U64 retry = 0;
      if ((s = mgScanFinish (d, 0, b, nOut, &retry, st))) return s;
      if (!retry) return MG_OK;
      cap = retry;
    }
  mgSetError ("scan capacity could not be established");
  return MG_ERR_CAPACITY;
}

extern "C" MgStatus mgAddReadsDevice (Modset *ms, const U32 *dPacked, U64 totalBases,
                                      const U64 *dReadOffsets, U32 nReads, U64 *nHash, void *stream)
{
  hipStream_t st = (hipStream_t) stream;
  MgDev *d; MgStatus s = mgDevGet (ms, &d, st); if (s) return s;
  if (nHash) *nHash = 0;
  if (!totalBases || !nReads) return MG_OK;
  MgScanBufs b; U64 n = 0;
  d->t.loadPct = 75;                                   /* built and counted: see MgTable.loadPct */
  /* pos / isF / read are not needed by addSequence (modutils.c:24 passes 0 for isF and ignores pos) */
  /* the dense k-mer array is only made if the insert turns out to need it (a small batch, the table regrown): a large
     batch's first partition pass and index assignment read the scan's segments */
  if ((s = mgScanIntoArena (d, ms->hasher, dPacked, totalBases, dReadOffsets, nReads, false, 4, &b, &n, st, 0, 0, 0, true))) return s;
  if (nHash) *nHash = n;
  if (!n) return MG_OK;
  if (b.lazy)
    { const U64 chunk = n < MG_ADD_CHUNK ? n : MG_ADD_CHUNK;
      if ((s = mgTableEnsure (&d->t, chunk, st))) return s;
      if (n <= MG_ADD_CHUNK && mgTableAddTakesSegments (&d->t, n, &b.counted))
        return mgAddBatch (ms, d, 0, n, 0, 1, true, st, &b.counted, &b.seg);
      if ((s = mgLaunchSegCompact (b.seg, b.kmer, b.cap, b.count, st))) return s;
    }
  return mgAddBatch (ms, d, b.kmer, n, 0, 1, true, st, &b.counted);
}

/* mode 0: lookup only (modmap.c:202); mode 1: insert without depth (modmap.c:109) */
static MgStatus mgSeedReads (Modset *ms, int mode, const U32 *dPacked, U64 totalBases,
                             const U64 *dReadOffsets, U32 nReads,
                             U32 *dSeedIndex, U32 *dSeedPosF, U32 *dSeedRead, U64 capacity,
                             U64 *nSeeds, hipStream_t st)
{
  MgDev *d; MgStatus s = mgDevGet (ms, &d, st); if (s) return s;
  if (nSeeds) *nSeeds = 0;
  if (!totalBases || !nReads) return MG_OK;
  MgScanBufs b; U64 n = 0;
  d->t.loadPct = MG_LOOKUP_LOAD_PCT;                                   /* lookups follow (or are this call): see MgTable.loadPct */
  if (mode == 0 && d->t.slots && (s = mgTableEnsure (&d->t, 0, st))) return s;
  const int timing = mgKnobs ()->seedTiming == 1;   /* dev */
  struct timespec q0, q1, q2; if (timing) clock_gettime (CLOCK_MONOTONIC, &q0);
  /* lookups read the k-mers from the scan's segments (no dense copy of them); pos / read are compacted as before */
  if ((s = mgScanIntoArena (d, ms->hasher, dPacked, totalBases, dReadOffsets, nReads, true, mode ? 4 : 0, &b, &n, st,
                            dSeedPosF, dSeedRead, capacity, mode == 0))) return s;
  if (timing) clock_gettime (CLOCK_MONOTONIC, &q1);
  if (nSeeds) *nSeeds = n;
  if (n > capacity)
    { mgSetError ("%llu seeds exceed the caller's capacity %llu", (unsigned long long) n, (unsigned long long) capacity); return MG_ERR_CAPACITY; }
  if (mode == 0)
    { if (b.lazy && b.findScratch && mgTableFindTakesPartition (&d->t, n, &b.counted))
        s = mgTableFindPartitioned (&d->t, b.seg, n, &b.counted, dSeedIndex, b.kmer, b.findScratch, st, b.findScratch2);     /* (b.kmer: the dense array a lazy scan leaves unused) */
      else s = b.lazy ? mgTableFindSegments (&d->t, b.seg, n, dSeedIndex, st) : mgTableFind (&d->t, b.kmer, n, dSeedIndex, st);
    }
  else s = mgAddBatch (ms, d, b.kmer, n, dSeedIndex, 0, true, st, &b.counted);
  if (s) return s;
  if (dSeedPosF && b.posF != dSeedPosF) MG_HIP (hipMemcpyAsync (dSeedPosF, b.posF, n * 4, hipMemcpyDeviceToDevice, st));
  if (dSeedRead && b.rid != dSeedRead) MG_HIP (hipMemcpyAsync (dSeedRead, b.rid, n * 4, hipMemcpyDeviceToDevice, st));
  MG_HIP (hipStreamSynchronize (st));
  if (timing)
    { clock_gettime (CLOCK_MONOTONIC, &q2);
      fprintf (stderr, "mgSeedReads: scan+count %.2f ms, lookup+sync %.2f ms\n", (q1.tv_sec - q0.tv_sec) * 1e3 + (q1.tv_nsec - q0.tv_nsec) * 1e-6,
               (q2.tv_sec - q1.tv_sec) * 1e3 + (q2.tv_nsec - q1.tv_nsec) * 1e-6);
    }
  return MG_OK;
}

extern "C" MgStatus mgQueryReadsDevice (Modset *ms, const U32 *dPacked, U64 totalBases,
                                        const U64 *dReadOffsets, U32 nReads,
                                        U32 *dSeedIndex, U32 *dSeedPosF, U32 *dSeedRead, U64 capacity,
                                        U64 *nSeeds, void *stream)
{ return mgSeedReads (ms, 0, dPacked, totalBases, dReadOffsets, nReads, dSeedIndex, dSeedPosF, dSeedRead, capacity, nSeeds, (hipStream_t) stream); }

extern "C" MgStatus mgInsertReadsDevice (Modset *ms, const U32 *dPacked, U64 totalBases,
                                         const U64 *dReadOffsets, U32 nReads,
                                         U32 *dSeedIndex, U32 *dSeedPosF, U32 *dSeedRead, U64 capacity,
                                         U64 *nSeeds, void *stream)
{ return mgSeedReads (ms, 1, dPacked, totalBases, dReadOffsets, nReads, dSeedIndex, dSeedPosF, dSeedRead, capacity, nSeeds, (hipStream_t) stream); }

/* The lookup loop of queryProcess (modmap.c:197-206) for batches that follow one another, in two halves: Async starts the batch's SCAN
 * on a stream of the library's own, into the other of two scratch arenas, and returns; Wait runs its LOOKUPS on the caller's stream and
 * returns when the seeds are complete.  Called as  Async (0); for i: Async (i + 1); Wait (i)  the scan of batch i + 1 -- bound by
 * instruction issue -- runs beside the lookups of batch i -- bound by memory requests (profiles/r03_corun_matrix.txt: 7 - 8 % on the
 * pair).  At most two batches in flight, waited for in the order they were started; each needs its own output arrays; until the last
 * ticket is waited for, the modset takes no other batch call. */
struct MgQueryTicket { Modset *ms; MgDev *d; int slot; MgScanReq q; MgScanBufs b; U32 *dSeedIndex, *dSeedPosF, *dSeedRead; U64 capacity; };

extern "C" MgStatus mgQueryReadsDeviceAsync (Modset *ms, const U32 *dPacked, U64 totalBases, const U64 *dReadOffsets, U32 nReads,
                                             U32 *dSeedIndex, U32 *dSeedPosF, U32 *dSeedRead, U64 capacity, void **ticket, void *stream)
{
  hipStream_t st = (hipStream_t) stream;
  if (!ticket) { mgSetError ("mgQueryReadsDeviceAsync: null ticket"); return MG_ERR_ARG; }
  *ticket = 0;
  MgDev *d; MgStatus s = mgDevGet (ms, &d, st); if (s) return s;
  if (d->ticketsOut >= 2) { mgSetError ("two query batches are in flight already"); return MG_ERR_ARG; }
  d->t.loadPct = MG_LOOKUP_LOAD_PCT;
  if (d->t.slots && !d->ticketsOut && (s = mgTableEnsure (&d->t, 0, st))) return s;      /* (with a batch in flight the table has its lookup shape already) */
  if (!d->side)
    { /* lowest priority: the lookups are a chain of short kernels that should not queue behind the scan's workgroups; the scan fills what they leave */
      int lo = 0, hi = 0; (void) hipDeviceGetStreamPriorityRange (&lo, &hi);
      MG_HIP (hipStreamCreateWithPriority (&d->side, hipStreamNonBlocking, lo));      /* (default priority instead: no difference measured, DESIGN_EXPERIMENTS §J) */
      for (int i = 0 ; i < 2 ; ++i) MG_HIP (hipEventCreateWithFlags (&d->scanned[i], hipEventDisableTiming));
      MG_HIP (hipEventCreateWithFlags (&d->inputReady, hipEventDisableTiming));
    }
  const int slot = d->slotBusy[d->nextSlot] ? d->nextSlot ^ 1 : d->nextSlot;
  MgQueryTicket *t = new MgQueryTicket ();
  t->ms = ms; t->d = d; t->slot = slot; t->dSeedIndex = dSeedIndex; t->dSeedPosF = dSeedPosF; t->dSeedRead = dSeedRead; t->capacity = capacity;
  t->q = MgScanReq { ms->hasher, dPacked, totalBases, dReadOffsets, nReads, true, 0, dSeedPosF, dSeedRead, capacity, true };
  t->b = MgScanBufs (); t->b.cap = 0;
  if (totalBases && nReads)
    { /* the batch (and a table that has just been reshaped) is ready when the caller's stream gets here: the scan's stream waits for that */
      hipError_t e = hipEventRecord (d->inputReady, st);
      if (e == hipSuccess) e = hipStreamWaitEvent (d->side, d->inputReady, 0);
      if (e != hipSuccess) { delete t; return mgHipFail (e, "mgQueryReadsDeviceAsync"); }
      if ((s = mgScanStart (d, slot, t->q, mgSurvivorGuess (ms->hasher, totalBases), &t->b, d->side))) { delete t; return s; }
      if ((e = hipEventRecord (d->scanned[slot], d->side)) != hipSuccess) { delete t; return mgHipFail (e, "mgQueryReadsDeviceAsync"); }
    }
  d->slotBusy[slot] = true; d->nextSlot = slot ^ 1; ++d->ticketsOut;
  *ticket = t;
  return MG_OK;
}

extern "C" MgStatus mgQueryReadsDeviceWait (void *ticket, U64 *nSeeds, void *stream)
{
  hipStream_t st = (hipStream_t) stream;
  MgQueryTicket *t = (MgQueryTicket *) ticket;
  if (!t) { mgSetError ("mgQueryReadsDeviceWait: null ticket"); return MG_ERR_ARG; }
  MgDev *d = t->d;
  if (nSeeds) *nSeeds = 0;
  MgStatus s = MG_OK;
  U64 n = 0;
  if (t->q.totalBases && t->q.nReads)
    do {
      U64 retry = 0;
      if ((s = mgScanFinish (d, t->slot, &t->b, &n, &retry, d->side, d->scanned[t->slot]))) break;
      for (int attempt = 0 ; retry && attempt < 2 && !s ; ++attempt)      /* the guess was too small: again, with what the scan asked for */
        { if ((s = mgScanStart (d, t->slot, t->q, retry, &t->b, d->side))) break;
          s = mgScanFinish (d, t->slot, &t->b, &n, &retry, d->side);
        }
      if (s) break;
      if (retry) { mgSetError ("scan capacity could not be established"); s = MG_ERR_CAPACITY; break; }
      if (nSeeds) *nSeeds = n;
      if (n > t->capacity) { mgSetError ("%llu seeds exceed the caller's capacity %llu", (unsigned long long) n, (unsigned long long) t->capacity); s = MG_ERR_CAPACITY; break; }
      /* (the scan is complete: the host has waited for it, so the caller's stream need not) */
      MgScanBufs &b = t->b;
      if (b.lazy && b.findScratch && mgTableFindTakesPartition (&d->t, n, &b.counted))
        s = mgTableFindPartitioned (&d->t, b.seg, n, &b.counted, t->dSeedIndex, b.kmer, b.findScratch, st, b.findScratch2);
      else s = b.lazy ? mgTableFindSegments (&d->t, b.seg, n, t->dSeedIndex, st) : mgTableFind (&d->t, b.kmer, n, t->dSeedIndex, st);
      if (s) break;
      hipError_t e = hipSuccess;
      if (t->dSeedPosF && b.posF != t->dSeedPosF) e = hipMemcpyAsync (t->dSeedPosF, b.posF, n * 4, hipMemcpyDeviceToDevice, st);
      if (e == hipSuccess && t->dSeedRead && b.rid != t->dSeedRead) e = hipMemcpyAsync (t->dSeedRead, b.rid, n * 4, hipMemcpyDeviceToDevice, st);
      if (e == hipSuccess) e = hipStreamSynchronize (st);
      if (e != hipSuccess) s = mgHipFail (e, "mgQueryReadsDeviceWait");
    } while (0);
  d->slotBusy[t->slot] = false; --d->ticketsOut;
  delete t;
  return s;
}

/* ---------------------------------------------------------------------------------------- */
/* host-buffer mirrors of the reference callers' loops                                        */

/* grow-only device buffers for the host-buffer entry points (a hipMalloc + hipFree of a gigabyte per call costs
 * milliseconds): the packed reads and their offsets of the batch in flight */
static struct MgHostBatchBufs { U32 *dP = 0; size_t words = 0; U64 *dOff = 0; size_t offs = 0; std::mutex lock; } gHbs[MG_MAXDEV];   /* by device */
extern "C" void mgHostBatchRelease (void)
{
  int before = -1; if (hipGetDevice (&before) != hipSuccess) { (void) hipGetLastError (); before = -1; }
  for (int dev = 0 ; dev < MG_MAXDEV ; ++dev)
    { MgHostBatchBufs &b = gHbs[dev];
      std::lock_guard<std::mutex> g (b.lock);
      if (!b.dP && !b.dOff) continue;
      (void) hipSetDevice (dev);
      if (b.dP) (void) hipFree (b.dP);
      if (b.dOff) (void) hipFree (b.dOff);
      b.dP = 0; b.words = 0; b.dOff = 0; b.offs = 0;
    }
  if (before >= 0) (void) hipSetDevice (before);
}
static double mgNowS (void) { struct timespec t; clock_gettime (CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

extern "C" int64_t mgAddSequenceBatch (Modset *ms, const char *bases, const int64_t *readOffsets, int nReads)
{
  if (mgEnsureDevice ()) return -1;
  if (nReads <= 0) return 0;
  const int timing = mgKnobs ()->uploadTiming == 1;   /* dev */
  int curDev = 0; if (hipGetDevice (&curDev) != hipSuccess) { mgSetError ("mgAddSequenceBatch: no current device"); return -1; }
  if (curDev < 0 || curDev >= MG_MAXDEV) { mgSetError ("mgAddSequenceBatch: device number beyond %d", MG_MAXDEV - 1); return -1; }
  MgHostBatchBufs &gHb = gHbs[curDev];
  std::lock_guard<std::mutex> g (gHb.lock);            /* the buffers of this device: one batch at a time through them; a thread on another GPU has that GPU's */
  U64 total = (U64) readOffsets[nReads];
  size_t nw = mgPackedWords (total);
  if (nw > gHb.words)
    { if (gHb.dP) (void) hipFree (gHb.dP);
      gHb.dP = 0; gHb.words = 0;
      if (hipMalloc ((void **) &gHb.dP, (nw + nw / 8) * 4) != hipSuccess) { mgSetError ("mgAddSequenceBatch: device allocation failed"); return -1; }
      gHb.words = nw + nw / 8;
    }
  if ((size_t) nReads + 1 > gHb.offs)
    { if (gHb.dOff) (void) hipFree (gHb.dOff);
      gHb.dOff = 0; gHb.offs = 0;
      if (hipMalloc ((void **) &gHb.dOff, ((size_t) nReads + 1) * 2 * 8) != hipSuccess) { mgSetError ("mgAddSequenceBatch: device allocation failed"); return -1; }
      gHb.offs = ((size_t) nReads + 1) * 2;
    }
  int64_t res = -1;
  const double t0 = mgNowS ();
  if (hipMemcpyAsync (gHb.dOff, readOffsets, ((size_t) nReads + 1) * 8, hipMemcpyHostToDevice, 0) == hipSuccess
      && mgUploadPack (bases, total, gHb.dP, 0) == MG_OK)
    { const double t1 = mgNowS ();
      U64 nHash = 0;
      if (mgAddReadsDevice (ms, gHb.dP, total, gHb.dOff, (U32) nReads, &nHash, 0) == MG_OK) res = (int64_t) nHash;
      if (timing) fprintf (stderr, "mgAddSequenceBatch: %.3f Gbp  pack+upload %.2f ms  scan+build %.2f ms\n", total / 1e9, (t1 - t0) * 1e3, (mgNowS () - t1) * 1e3);
    }
  else mgSetError ("mgAddSequenceBatch: copy to the device failed");
  return res;
}

extern "C" void mgDepthHistogram (Modset *ms, FILE *f)
{
  U64 *hist = (U64 *) calloc (65536, sizeof (U64));
  MgDev *d = mgDevLookup (ms);
  bool done = false;
  if (d)
    { U64 *dHist = 0;
      if (hipMalloc ((void **) &dHist, 65536 * 8) == hipSuccess)
        { if (hipMemset (dHist, 0, 65536 * 8) == hipSuccess && modsetDepthHistogramDevice (ms, dHist, 0) == MG_OK
              && hipMemcpy (hist, dHist, 65536 * 8, hipMemcpyDeviceToHost) == hipSuccess) done = true;
          (void) hipFree (dHist);
        }
      if (!done) { fprintf (stderr, "FATAL ERROR: depth histogram on device failed: %s\n", mgLastError ()); exit (-1); }
    }
  else
    for (U32 i = 1 ; i <= ms->max ; ++i) ++hist[ms->depth[i]];      /* host-only modset: nothing on the device */
  for (U32 b = 0 ; b < 65536 ; ++b)
    if (hist[b]) fprintf (f, "DP\t%u\t%u\n", b, (U32) hist[b]);       /* modutils.c:61 */
  free (hist);
}

/* ---------------------------------------------------------------------------------------- */
/* per-read iterator facade: one GPU scan per modRCiterator call, replayed by modRCnext       */

/* The reference's callers scan read by read (modutils.c:19-31, modmap.c:197-206): modRCiterator, modRCnext until false,
 * destroy.  One call here = pack the read into a pinned host buffer the GPU reads in place, ONE kernel launch
 * (mgIterScanKernel: scan + ordered concatenation + completion flag), the replay block {n, k-mers, pos|isF} written by the
 * kernel straight into pinned host memory, the host polling the flag: no host-to-device or device-to-host copy call and
 * no stream wait on the way.  Reads longer than mgIterMaxBases () take the batch scan with the block copied back.
 * The scratch is per host thread and per device. */
#include <immintrin.h>
static bool gRuntimeAlive = true;                          /* false once the library is being unloaded: thread-local destructors that run after that leave HIP alone */
__attribute__ ((destructor)) static void mgRuntimeDown (void) { gRuntimeAlive = false; }
static bool mgRuntimeAlive (void) { return gRuntimeAlive; }
struct MgIterScratch {
  int dev = -1; hipStream_t st = 0;
  char *hIn = 0; char *dIn = 0; size_t inBytes = 0;          /* pinned: {0, len} (2 x U64), then the packed words */
  U64 *hOut = 0; U64 *dOut = 0; size_t outEntries = 0;       /* pinned: the replay block */
  U64 *hFlag = 0; U64 *dFlag = 0; U64 seq = 0;               /* pinned: the kernel's completion flag */
  U64 *dSegK = 0; U32 *dSegP = 0;                            /* device: the workers' segments */
  /* the long-read path */
  U32 *dPacked = 0; U64 *dOff = 0; U64 *dKmer = 0; U32 *dPosF = 0; void *dWork = 0; U64 *dCount = 0;
  size_t wordsCap = 0, survCap = 0, workCap = 0;
  U32 *hPacked = 0; size_t hWordsCap = 0;
  ~MgIterScratch () { if (dev >= 0 && mgRuntimeAlive ()) release (); }      /* a host thread that ends gives its pinned buffers and its stream back */
  void release ()
  { if (st) (void) hipStreamSynchronize (st);              /* no kernel of this scratch is still writing into what is freed */
    if (hIn) (void) hipHostFree (hIn);
    if (hOut) (void) hipHostFree (hOut);
    if (hFlag) (void) hipHostFree (hFlag);
    (void) hipFree (dSegK); (void) hipFree (dSegP); (void) hipFree (dPacked); (void) hipFree (dOff); (void) hipFree (dKmer);
    (void) hipFree (dPosF); (void) hipFree (dWork); (void) hipFree (dCount);
    if (st) (void) hipStreamDestroy (st);
    free (hPacked);
    *this = MgIterScratch ();
  }
};
static thread_local MgIterScratch gIt;

static int mgIterPinned (void **h, void **d, size_t bytes)
{
  if (hipHostMalloc (h, bytes, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return -1;
  if (hipHostGetDevicePointer (d, *h, 0) != hipSuccess) return -1;
  return 0;
}

/* the scratch for the calling thread on the current device */
static int mgIterPrepare (MgIterScratch &g)
{
  int dev = 0;
  if (hipGetDevice (&dev) != hipSuccess) return -1;
  if (g.dev == dev) return 0;
  if (g.dev >= 0) g.release ();                        /* the thread moved to another GPU (mgSetDevice) */
  g.inBytes = 16 + mgPackedWords (mgIterMaxBases ()) * 4;
  if (hipStreamCreateWithFlags (&g.st, hipStreamNonBlocking) != hipSuccess) return -1;
  if (mgIterPinned ((void **) &g.hIn, (void **) &g.dIn, g.inBytes)) return -1;
  if (mgIterPinned ((void **) &g.hFlag, (void **) &g.dFlag, 64)) return -1;
  g.hFlag[0] = 0;
  if (hipMalloc ((void **) &g.dSegK, mgIterSegEntries () * 8) != hipSuccess || hipMalloc ((void **) &g.dSegP, mgIterSegEntries () * 4) != hipSuccess) return -1;
  g.dev = dev;
  return 0;
}

static int mgIterOutReserve (MgIterScratch &g, size_t entries)
{
  if (entries <= g.outEntries) return 0;
  if (g.hOut) { (void) hipStreamSynchronize (g.st); (void) hipHostFree (g.hOut); g.hOut = 0; g.outEntries = 0; }
  const size_t want = entries + entries / 4 + 4096;
  if (mgIterPinned ((void **) &g.hOut, (void **) &g.dOut, (want + 2) * 12)) return -1;
  g.outEntries = want;
  return 0;
}

/* reads beyond the one-launch kernel's reach: the batch scan, the block copied back */
static int mgIterScanLong (MgIterScratch &g, Seqhash *sh, const char *s, U64 total, U64 **blkOut)
{
  size_t nw = mgPackedWords (total);
  if (nw > g.hWordsCap) { free (g.hPacked); g.hPacked = (U32 *) malloc (2 * nw * 4); g.hWordsCap = 2 * nw; }
  mgPackHost (s, total, g.hPacked);
  if (!g.dOff) { if (hipMalloc ((void **) &g.dOff, 16) != hipSuccess || hipMalloc ((void **) &g.dCount, 8 * MG_COUNT_WORDS) != hipSuccess) return -1; }
  if (nw > g.wordsCap)
    { if (g.dPacked) (void) hipFree (g.dPacked);
      if (hipMalloc ((void **) &g.dPacked, 2 * nw * 4) != hipSuccess) return -1;
      g.wordsCap = 2 * nw;
    }
  U64 cap = mgSurvivorGuess (sh, total);
  MgHashParams p = mgMakeParams (sh);
  U64 off[2] = { 0, total };
  for (int attempt = 0 ; attempt < 3 ; ++attempt)
    { if (cap > g.survCap)
        { if (g.dKmer) (void) hipFree (g.dKmer);
          if (g.dPosF) (void) hipFree (g.dPosF);
          if (hipMalloc ((void **) &g.dKmer, 2 * cap * 8) != hipSuccess || hipMalloc ((void **) &g.dPosF, 2 * cap * 4) != hipSuccess) return -1;
          g.survCap = 2 * cap;
        }
      size_t wb = mgScanWorkBytes (total, 1, g.survCap);
      if (wb > g.workCap)
        { if (g.dWork) (void) hipFree (g.dWork);
          if (hipMalloc (&g.dWork, 2 * wb) != hipSuccess) return -1;
          g.workCap = 2 * wb;
        }
      if (hipMemcpyAsync (g.dPacked, g.hPacked, nw * 4, hipMemcpyHostToDevice, g.st) != hipSuccess) return -1;
      if (hipMemcpyAsync (g.dOff, off, 16, hipMemcpyHostToDevice, g.st) != hipSuccess) return -1;
      if (mgLaunchScan (p, g.dPacked, total, g.dOff, 1, g.dKmer, g.dPosF, 0, g.survCap, g.dCount, g.dWork, g.st)) return -1;
      U64 c[MG_COUNT_WORDS];
      if (hipMemcpyAsync (c, g.dCount, sizeof (c), hipMemcpyDeviceToHost, g.st) != hipSuccess || hipStreamSynchronize (g.st) != hipSuccess) return -1;
      if (c[1] || c[0] > g.survCap) { cap = c[3]; continue; }
      const U64 n = c[0];
      U64 *blk = (U64 *) malloc ((size_t) (n + 1) * 8 + (size_t) n * 4 + 8);
      if (!blk) return -1;
      blk[0] = n;
      if (n && (hipMemcpyAsync (blk + 1, g.dKmer, n * 8, hipMemcpyDeviceToHost, g.st) != hipSuccess
                || hipMemcpyAsync (blk + 1 + n, g.dPosF, n * 4, hipMemcpyDeviceToHost, g.st) != hipSuccess
                || hipStreamSynchronize (g.st) != hipSuccess)) { free (blk); return -1; }
      *blkOut = blk;
      return 0;
    }
  return -1;
}

/* returns 0 on success; *blkOut is the malloc()ed replay block {n, n k-mers (U64), n pos | isF << 31 (U32)} -- what the
   iterator's hashBuf points to (the reference's callers free () it: seqhash.h:54-55) */
extern "C" int mgIterScan (Seqhash *sh, const char *s, int len, U64 **blkOut)
{
  *blkOut = 0;
  if (mgEnsureDevice ()) return -1;
  if (len < sh->k)
    { U64 *blk = (U64 *) malloc (16); if (!blk) return -1;
      blk[0] = 0; *blkOut = blk; return 0;
    }
  MgIterScratch &g = gIt;
  if (mgIterPrepare (g)) { if (!gErr[0]) mgSetError ("iterator scratch: %s", hipGetErrorString (hipGetLastError ())); return -1; }
  const U64 total = (U64) len;
  if (total > mgIterMaxBases ()) return mgIterScanLong (g, sh, s, total, blkOut);
  U64 *hdr = (U64 *) g.hIn; hdr[0] = 0; hdr[1] = total;
  mgPackHost (s, total, (U32 *) (g.hIn + 16));
  U64 want = total / (U64) sh->w + total / (4 * (U64) sh->w) + 256;      /* expected modimizers, a quarter more; the kernel says so if it was not enough */
  const MgHashParams p = mgMakeParams (sh);
  for (int attempt = 0 ; attempt < 2 ; ++attempt)
    { if (mgIterOutReserve (g, want)) { mgSetError ("iterator scratch: pinned allocation failed"); return -1; }
      const U64 seq = ++g.seq;
      if (mgLaunchIterScan (p, (const U32 *) (g.dIn + 16), total, (const U64 *) g.dIn, g.dSegK, g.dSegP, g.dOut, g.outEntries, g.dFlag, seq, g.st)) return -1;
      U64 *flag = g.hFlag;
      bool done = false;
      /* acquire loads: the block the kernel wrote before its release store of the flag is read after the flag is seen */
      for (int spin = 0 ; spin < (1 << 22) ; ++spin) { if (__atomic_load_n (flag, __ATOMIC_ACQUIRE) == seq) { done = true; break; } _mm_pause (); }
      if (!done)                                           /* a kernel that takes this long, or one that failed: ask the runtime */
        { hipError_t e = hipStreamSynchronize (g.st);
          if (e != hipSuccess) { mgHipFail (e, "iterator scan"); g.release (); return -1; }      /* (nothing of this scratch is reused after a failure) */
          if (__atomic_load_n (flag, __ATOMIC_ACQUIRE) != seq) { mgSetError ("iterator scan: no completion flag"); g.release (); return -1; }
        }
      const U64 n = g.hOut[0];
      if (n > g.outEntries) { want = n; continue; }
      const size_t bytes = (size_t) (n + 1) * 8 + (size_t) n * 4;
      U64 *blk = (U64 *) malloc (bytes + 8);
      if (!blk) return -1;
      memcpy (blk, g.hOut, bytes);
      *blkOut = blk;
      return 0;
    }
  mgSetError ("iterator scan: output capacity could not be established");
  return -1;
}

extern "C" int mgIterRequireDevice (void) { return mgEnsureDevice () ? -1 : 0; }

extern "C" void mgIterReleaseBuffers (void) { if (gIt.dev >= 0) gIt.release (); }

extern "C" void mgSeqReleaseBuffers (void);
extern "C" void mgReleaseBuffers (void)
{
  mgSeqReleaseBuffers ();
  mgTextReleaseBuffers ();
  mgQueryReleaseBuffers ();
  mgHostBatchRelease ();
  mgIterReleaseBuffers ();
  mgXferReleaseBuffers ();
  int before = -1; if (hipGetDevice (&before) != hipSuccess) { (void) hipGetLastError (); before = -1; }
  for (int dev = 0 ; dev < MG_MAXDEV ; ++dev)
    { MgUpStage &u = gUps[dev];
      std::lock_guard<std::mutex> g (u.lock);
      if (!u.ready && !u.pin[0] && !u.pin[1]) continue;
      (void) hipSetDevice (dev);
      for (int i = 0 ; i < 2 ; ++i) { if (u.ready) (void) hipEventDestroy (u.done[i]); if (u.pin[i]) (void) hipHostFree (u.pin[i]); u.pin[i] = 0; }
      u.ready = false;
    }
  if (before >= 0) (void) hipSetDevice (before);
}

/* ---------------------------------------------------------------------------------------- */
/* synthetic data                                                                             */

extern "C" MgStatus mgSynthGenome (U32 *dPacked, U64 nBases, U64 seed, void *stream)
{ MgStatus s = mgEnsureDevice (); if (s) return s; return mgLaunchSynthGenome (dPacked, nBases, seed, (hipStream_t) stream); }

extern "C" MgStatus mgSynthReads (const U32 *dGenomePacked, U64 genomeBases,
                                  const U64 *dReadStart, const U64 *dReadOffsets, const U8 *dStrand, U32 nReads,
                                  U64 totalBases, double errRate, U64 seed, U32 *dPackedOut, void *stream)
{ MgStatus s = mgEnsureDevice (); if (s) return s;
  return mgLaunchSynthReads (dGenomePacked, genomeBases, dReadStart, dReadOffsets, dStrand, nReads, totalBases,
                             errRate, seed, dPackedOut, (hipStream_t) stream);
}
