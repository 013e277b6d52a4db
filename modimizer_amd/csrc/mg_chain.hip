/* mg_chain.hip — modmap's queryProcess (modmap.c:188-281) for a batch of reads, on the device:
 * the seed list of every read (scan + lookup, mgQueryReadsDevice) and, new here, the tallies of the
 * "Q" line and the chaining of seeds into "M" blocks (modmap.c:213-276).  The chaining is serial
 * inside a read — every step looks at the block built so far — but reads are independent: one lane
 * per read walks its seeds.  What comes back to the host is a few integers per read and per block;
 * the host only formats the lines.
 *
 * The reference's rules, kept as they are (they decide what is printed):
 *   - seeds that miss, and copy-M seeds, are skipped; a seed's place in the reference is its FIRST
 *     occurrence, rev[loc[index]] (modmap.c:219-221);
 *   - occurrence number 0 doubles as "no block open" (modmap.c:232);
 *   - a block ends when the next seed is on another sequence, goes backwards along the block's
 *     direction, or when the block's extent in the reference and in the read differ by more than 50
 *     seeds (modmap.c:233-241); a copy-2 seed that would end the block is retried at its second
 *     occurrence (modmap.c:242-254);
 *   - a block is reported when it ENDS only if it holds more than two copy-1 seeds, and the block
 *     still open at the end of the read only if it holds more than two copy-2 seeds (modmap.c:256,269).
 */
#include <hip/hip_runtime.h>
#include <mutex>
#include <unordered_map>
#include "mg_common.h"
#include "mg_internal.h"
#include "mg_xfer.h"
#include "mg_ref.h"

/* seedStart[r] = first seed of read r (seeds are in read order); seedStart[nReads] = nSeeds */
__global__ void mgSeedStartKernel (const U32 *__restrict__ seedRead, U64 nSeeds, U32 nReads, U64 *__restrict__ seedStart)
{
  const U64 stride = (U64) gridDim.x * blockDim.x;
  for (U64 i = (U64) blockIdx.x * blockDim.x + threadIdx.x ; i <= nSeeds ; i += stride)
    { const U32 cur = i < nSeeds ? seedRead[i] : nReads;
      const U32 prev = i ? seedRead[i - 1] + 1 : 0;            /* reads prev .. cur start here (empty ones included) */
      for (U32 r = prev ; r <= cur && r <= nReads ; ++r) seedStart[r] = i;
    }
}

__device__ __forceinline__ bool mgBlockEnds (const MgRefDev &d, U32 loc, U32 loc0, U32 locN, U32 i0, U32 iN, bool withUnset)
{
  if (withUnset && !loc0) return true;
  if (d.id[loc] != d.id[loc0]) return true;
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

__global__ __launch_bounds__ (256)
void mgChainKernel (const U32 *__restrict__ seedIx, const U32 *__restrict__ seedPos, const U64 *__restrict__ seedStart,
                    U32 nReads, const MgRefDev d, MgChainQ *__restrict__ q, MgChainM *__restrict__ mRec, U32 maxM,
                    U32 *__restrict__ overflow)
{
  const U32 r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nReads) return;
  const U64 s0 = seedStart[r], s1 = seedStart[r + 1];
  const U32 ns = (U32) (s1 - s0);
  MgChainQ qq; qq.nSeeds = ns; qq.missed = 0; qq.copy1 = qq.copy2 = qq.copyM = 0; qq.nM = 0;
  U32 loc0 = 0, locN = 0, i0 = 0, iN = 0;
  int n1 = 0, n2 = 0;
  MgChainM *mine = mRec + (U64) r * maxM;
#define MG_EMIT() do { if (qq.nM < maxM) { MgChainM e; e.pos0 = seedPos[s0 + i0] & MG_POS_MASK; e.posN = seedPos[s0 + iN] & MG_POS_MASK; \
      e.id0 = d.id[loc0]; e.off0 = d.offset[loc0]; e.offN = d.offset[locN]; e.n1 = n1; e.n2 = n2; \
      e.span = locN > loc0 ? locN - loc0 : loc0 - locN; mine[qq.nM] = e; } else *overflow = 1; ++qq.nM; } while (0)
  for (U32 i = 0 ; i < ns ; ++i)
    { const U32 x = seedIx[s0 + i];
      if (!x) { ++qq.missed; continue; }
      const int c = d.info[x] & 3;
      if (c == 1) ++qq.copy1; else if (c == 2) ++qq.copy2; else if (c == 3) ++qq.copyM;
      if (c == 3) continue;
      U32 loc = d.rev[d.loc[x]];
      const bool is1 = c == 1;
      bool end = mgBlockEnds (d, loc, loc0, locN, i0, iN, true);
      if (end && loc0 && !is1)
        { loc = d.rev[d.loc[x] + 1];
          end = mgBlockEnds (d, loc, loc0, locN, i0, iN, false);
        }
      if (end)
        { if (n1 > 2) MG_EMIT ();
          n1 = n2 = 0; loc0 = loc; i0 = i;
        }
      if (is1) ++n1; else ++n2;
      locN = loc; iN = i;
    }
  if (n2 > 2) MG_EMIT ();
#undef MG_EMIT
  q[r] = qq;
}

/* the blocks of all reads, densely: read r's min (nM, maxM) blocks go to mStart[r] .. (the kernel above leaves them in slots of maxM per read:
   a short-read batch of 4e6 reads would copy 2 GB of mostly empty slots to the host) */
__global__ __launch_bounds__ (256)
void mgChainCompactKernel (const MgChainQ *__restrict__ q, const MgChainM *__restrict__ mRec, U32 maxM, const U64 *__restrict__ mStart, U32 nReads,
                           MgChainM *__restrict__ out)
{
  const U32 r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nReads) return;
  const U32 n = q[r].nM < maxM ? q[r].nM : maxM;
  const U64 at = mStart[r];
  for (U32 j = 0 ; j < n ; ++j) out[at + j] = mRec[(U64) r * maxM + j];
}

/* the device arrays of a query batch.  A call allocates and frees them; a caller with many batches in a row (mgQueryFile: one per
   window of the file) keeps them between its calls (mgChainScratchKeep): ten allocations and frees a batch are milliseconds.  What
   such a caller leaves stays allocated, like the text parser's windows, until mgReleaseBuffers () or a call that does not keep.
   The scratch belongs to the calling HOST THREAD and to the device it was allocated on (like the iterator's scratch, mg_api.hip): a
   thread that has moved to another GPU (mgSetDevice) drops it and starts again there, two threads driving two GPUs share nothing
   and do not wait for each other, and a thread that ends gives its blocks back. */
enum { CS_IX, CS_POS, CS_RID, CS_START, CS_Q, CS_M, CS_OV, CS_MC, CS_N };
static bool gChainAlive = true;                            /* false once the library is being unloaded: thread-local destructors that run after that leave HIP alone */
__attribute__ ((destructor)) static void mgChainDown (void) { gChainAlive = false; }
struct MgChainScratch
{ void *p[CS_N] = { 0 }; size_t cap[CS_N] = { 0 }; int keep = 0; int dev = -1;
  void drop (int i) { if (p[i]) (void) hipFree (p[i]); p[i] = 0; cap[i] = 0; }
  void dropAll () { for (int i = 0 ; i < CS_N ; ++i) drop (i); dev = -1; }
  ~MgChainScratch () { if (gChainAlive) dropAll (); }
};
static thread_local MgChainScratch gCs;
static int csPrepare (void)                                /* the scratch on the calling thread's current device */
{
  int dev = 0;
  if (hipGetDevice (&dev) != hipSuccess) return -1;
  if (gCs.dev >= 0 && gCs.dev != dev) gCs.dropAll ();
  gCs.dev = dev;
  return 0;
}
static void *csGet (int i, size_t bytes)
{
  if (gCs.cap[i] < bytes)
    { gCs.drop (i);
      const size_t want = gCs.keep ? bytes + bytes / 4 : bytes;
      if (hipMalloc (&gCs.p[i], want) != hipSuccess) return 0;
      gCs.cap[i] = want;
    }
  return gCs.p[i];
}
extern "C" void mgChainScratchKeep (int on)               /* (what a keeper leaves stays for the thread's next one: mgReleaseBuffers () frees it) */
{ if (on) ++gCs.keep; else if (gCs.keep) --gCs.keep; }
extern "C" void mgChainReleaseBuffers (void) { if (!gCs.keep) gCs.dropAll (); }

/* Q tallies and M blocks of every read of a device-resident batch.  hQ[nReads] is filled here; *hMOut is a malloc ()ed array
 * of all reads' blocks in read order (read r's are the next min (hQ[r].nM, maxM) entries), 0 when there is none; returns 1 if
 * some read had more than maxM blocks (the caller then does that batch the long way), 0 on success, -1 on error. */
extern "C" int mgChainQueryDevice (const MgReference *ref, const U32 *dPacked, U64 totalBases, const U64 *dReadOffsets, U32 nReads,
                                   MgChainQ *hQ, MgChainM **hMOut, U32 maxM, int hQPinned)
{
  *hMOut = 0;
  Modset *ms = ref->ms;
  MgRefDev d;
  if (mgRefDevGet (ref, &d)) return -1;
  U64 guess = totalBases / (U64) ms->hasher->w; guess += guess / 2 + 65536; if (guess > totalBases + 1) guess = totalBases + 1;
  if (csPrepare ()) { mgSetError ("query chaining: no current device"); return -1; }
  U32 *dIx = 0, *dPos = 0, *dRid = 0; U64 *dStart = 0; MgChainQ *dQ = 0; MgChainM *dM = 0, *dMc = 0; U32 *dOv = 0;
  U64 n = 0;
  int rc = -1;
  do {
    for (int attempt = 0 ; attempt < 2 ; ++attempt)
      { dIx = (U32 *) csGet (CS_IX, guess * 4); dPos = (U32 *) csGet (CS_POS, guess * 4); dRid = (U32 *) csGet (CS_RID, guess * 4);
        if (!dIx || !dPos || !dRid) { dIx = 0; break; }
        MgStatus s = mgQueryReadsDevice (ms, dPacked, totalBases, dReadOffsets, nReads, dIx, dPos, dRid, guess, &n, 0);
        if (s == MG_OK) break;
        dIx = 0;
        if (s == MG_ERR_CAPACITY && attempt == 0) { guess = n; continue; }
        break;
      }
    if (!dIx) break;
    dStart = (U64 *) csGet (CS_START, ((size_t) nReads + 2) * 8); dQ = (MgChainQ *) csGet (CS_Q, (size_t) nReads * sizeof (MgChainQ));
    dM = (MgChainM *) csGet (CS_M, (size_t) nReads * maxM * sizeof (MgChainM)); dOv = (U32 *) csGet (CS_OV, 4);
    if (!dStart || !dQ || !dM || !dOv) break;
    if (hipMemset (dOv, 0, 4)) break;
    unsigned grid = (unsigned) ((n + 1 + 255) / 256); if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL (mgSeedStartKernel, dim3 (grid), dim3 (256), 0, 0, dRid, n, nReads, dStart);
    hipLaunchKernelGGL (mgChainKernel, dim3 ((nReads + 255) / 256), dim3 (256), 0, 0, dIx, dPos, dStart, nReads, d, dQ, dM, maxM, dOv);
    if (hipGetLastError () != hipSuccess) break;
    U32 ov = 0;
    if (hipMemcpy (&ov, dOv, 4, hipMemcpyDeviceToHost)) break;
    if (hQPinned ? mgCopyOutPinned (hQ, dQ, (size_t) nReads * sizeof (MgChainQ), 0) != MG_OK : hipMemcpy (hQ, dQ, (size_t) nReads * sizeof (MgChainQ), hipMemcpyDeviceToHost) != hipSuccess) break;
    if (!ov)
      { /* where every read's blocks go in the dense array (dStart is free: the chain kernel is done with it) */
        U64 *hStart = (U64 *) malloc (((size_t) nReads + 1) * 8);
        if (!hStart) break;
        U64 tot = 0;
        for (U32 r = 0 ; r < nReads ; ++r) { hStart[r] = tot; tot += hQ[r].nM < maxM ? hQ[r].nM : maxM; }
        hStart[nReads] = tot;
        bool ok = true;
        if (tot)
          { MgChainM *hM = (MgChainM *) malloc ((size_t) tot * sizeof (MgChainM));
            dMc = (MgChainM *) csGet (CS_MC, (size_t) tot * sizeof (MgChainM));
            ok = hM && dMc && hipMemcpy (dStart, hStart, ((size_t) nReads + 1) * 8, hipMemcpyHostToDevice) == hipSuccess;
            if (ok)
              { hipLaunchKernelGGL (mgChainCompactKernel, dim3 ((nReads + 255) / 256), dim3 (256), 0, 0, dQ, dM, maxM, dStart, nReads, dMc);
                ok = hipGetLastError () == hipSuccess && hipMemcpy (hM, dMc, (size_t) tot * sizeof (MgChainM), hipMemcpyDeviceToHost) == hipSuccess;
              }
            if (ok) *hMOut = hM; else free (hM);
          }
        free (hStart);
        if (!ok) break;
      }
    rc = ov ? 1 : 0;
  } while (0);
  if (!gCs.keep) gCs.dropAll ();
  if (rc < 0 && !mgLastError ()[0]) mgSetError ("query chaining on the device failed");
  return rc;
}

/* ---------------------------------------------------------------------------------------- */
/* modasm's readsetFileRead (modasm.c:161-188) for a batch of reads: per read the hits (modset index,
 * top bit = forward), the 16-bit distance of each hit to the previous hit of the read, hit / miss counts,
 * and per mod the number of hits.  One lane per read walks its seeds twice (count, then write at the
 * offsets an exclusive scan of the counts gives). */

template <bool WRITE>
__global__ __launch_bounds__ (256)
void mgReadsetKernel (const U32 *__restrict__ seedIx, const U32 *__restrict__ seedPosF, const U64 *__restrict__ seedStart, U32 nReads,
                      U64 *__restrict__ hitStart, U32 *__restrict__ nMiss,
                      U32 *__restrict__ hit, unsigned short *__restrict__ dx, U32 *__restrict__ depthCount)
{
  const U32 r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nReads) return;
  const U64 s0 = seedStart[r], s1 = seedStart[r + 1];
  U64 out = WRITE ? hitStart[r] : 0;
  U32 miss = 0; int lastPos = 0;
  for (U64 i = s0 ; i < s1 ; ++i)
    { const U32 x = seedIx[i];
      if (!x) { ++miss; continue; }
      if (WRITE)
        { const U32 pf = seedPosF[i];
          const int pos = (int) (pf & MG_POS_MASK);
          hit[out] = (pf & MG_FWD_BIT) ? (x | 0x80000000u) : x;                  /* modasm.c:171 */
          dx[out] = (unsigned short) (pos - lastPos); lastPos = pos;             /* modasm.c:172 */
          atomicAdd (&depthCount[x], 1u);                                        /* modasm.c:174, saturated by the caller */
        }
      ++out;
    }
  if (!WRITE) { hitStart[r] = out; nMiss[r] = miss; }
}

/* exclusive scan of n counts in place, a[n] = total (one workgroup) */
__global__ __launch_bounds__ (1024)
void mgChainScanKernel (U64 *__restrict__ a, U32 n)
{
  __shared__ U64 sPart[1024];
  const int tid = threadIdx.x;
  const U32 per = (n + 1023) / 1024;
  U64 sum = 0;
  for (U32 i = 0 ; i < per ; ++i) { U32 j = tid * per + i; if (j < n) sum += a[j]; }
  sPart[tid] = sum;
  __syncthreads ();
  for (int off = 1 ; off < 1024 ; off <<= 1)
    { U64 v = tid >= off ? sPart[tid - off] : 0;
      __syncthreads ();
      sPart[tid] += v;
      __syncthreads ();
    }
  U64 run = sPart[tid] - sum;
  for (U32 i = 0 ; i < per ; ++i) { U32 j = tid * per + i; if (j < n) { U64 c = a[j]; a[j] = run; run += c; } }
  if (tid == 1023) a[n] = sPart[1023];
}

/* hHitStart[nReads+1], hNMiss[nReads], *hHit / *hDx malloc()ed here (totHit entries), hDepthCount[ms->max+1]
 * (hits per mod, unsaturated).  Returns 0, -1 on error. */
extern "C" int mgReadsetSeedsDevice (Modset *ms, const U32 *dPacked, U64 totalBases, const U64 *dReadOffsets, U32 nReads,
                                     U64 *hHitStart, U32 *hNMiss, U32 **hHit, unsigned short **hDx, U32 *hDepthCount)
{
  *hHit = 0; *hDx = 0;
  U64 guess = totalBases / (U64) ms->hasher->w; guess += guess / 2 + 65536; if (guess > totalBases + 1) guess = totalBases + 1;
  U32 *dIx = 0, *dPos = 0, *dRid = 0, *dMiss = 0, *dHit = 0, *dDepth = 0; unsigned short *dDx = 0; U64 *dStart = 0, *dHitStart = 0;
  U64 n = 0;
  int rc = -1;
  const size_t m = (size_t) ms->max + 1;
  do {
    for (int attempt = 0 ; attempt < 2 ; ++attempt)
      { if (hipMalloc ((void **) &dIx, guess * 4) || hipMalloc ((void **) &dPos, guess * 4) || hipMalloc ((void **) &dRid, guess * 4)) break;
        MgStatus s = mgQueryReadsDevice (ms, dPacked, totalBases, dReadOffsets, nReads, dIx, dPos, dRid, guess, &n, 0);
        if (s == MG_OK) break;
        (void) hipFree (dIx); (void) hipFree (dPos); (void) hipFree (dRid); dIx = dPos = dRid = 0;
        if (s == MG_ERR_CAPACITY && attempt == 0) { guess = n; continue; }
        break;
      }
    if (!dIx) break;
    if (hipMalloc ((void **) &dStart, ((size_t) nReads + 2) * 8) || hipMalloc ((void **) &dHitStart, ((size_t) nReads + 2) * 8)
        || hipMalloc ((void **) &dMiss, ((size_t) nReads + 1) * 4) || hipMalloc ((void **) &dDepth, m * 4)) break;
    if (hipMemset (dDepth, 0, m * 4)) break;
    unsigned grid = (unsigned) ((n + 1 + 255) / 256); if (grid > 16384) grid = 16384;
    const unsigned rgrid = (nReads + 255) / 256;
    hipLaunchKernelGGL (mgSeedStartKernel, dim3 (grid), dim3 (256), 0, 0, dRid, n, nReads, dStart);
    hipLaunchKernelGGL (mgReadsetKernel<false>, dim3 (rgrid), dim3 (256), 0, 0, dIx, dPos, dStart, nReads, dHitStart, dMiss,
                        (U32 *) 0, (unsigned short *) 0, (U32 *) 0);
    hipLaunchKernelGGL (mgChainScanKernel, dim3 (1), dim3 (1024), 0, 0, dHitStart, nReads);
    U64 totHit = 0;
    if (hipMemcpy (&totHit, dHitStart + nReads, 8, hipMemcpyDeviceToHost)) break;
    if (hipMalloc ((void **) &dHit, (totHit + 1) * 4) || hipMalloc ((void **) &dDx, (totHit + 1) * 2)) break;
    hipLaunchKernelGGL (mgReadsetKernel<true>, dim3 (rgrid), dim3 (256), 0, 0, dIx, dPos, dStart, nReads, dHitStart, dMiss, dHit, dDx, dDepth);
    if (hipGetLastError () != hipSuccess) break;
    *hHit = (U32 *) malloc ((size_t) (totHit + 1) * 4); *hDx = (unsigned short *) malloc ((size_t) (totHit + 1) * 2);
    if (hipMemcpy (hHitStart, dHitStart, ((size_t) nReads + 1) * 8, hipMemcpyDeviceToHost) || hipMemcpy (hNMiss, dMiss, (size_t) nReads * 4, hipMemcpyDeviceToHost)
        || (totHit && (hipMemcpy (*hHit, dHit, totHit * 4, hipMemcpyDeviceToHost) || hipMemcpy (*hDx, dDx, totHit * 2, hipMemcpyDeviceToHost)))
        || hipMemcpy (hDepthCount, dDepth, m * 4, hipMemcpyDeviceToHost)) break;
    rc = 0;
  } while (0);
  (void) hipFree (dIx); (void) hipFree (dPos); (void) hipFree (dRid); (void) hipFree (dStart); (void) hipFree (dHitStart);
  (void) hipFree (dMiss); (void) hipFree (dHit); (void) hipFree (dDx); (void) hipFree (dDepth);
  if (rc < 0) { free (*hHit); free (*hDx); *hHit = 0; *hDx = 0; if (!mgLastError ()[0]) mgSetError ("readset seeds on the device failed"); }
  return rc;
}
