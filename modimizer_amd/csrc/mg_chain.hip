/* mg_chain.hip — modmap's queryProcess (modmap.c:188-281) for a batch of reads, on the device:
 * the seed list of every read (scan + lookup, mgQueryReadsDevice) and, new here, the tallies of the
 * "Q" line and the chaining of seeds into "M" blocks (modmap.c:213-276).  The chaining is serial
 * inside a read — every step looks at the block built so far — but reads are independent: one lane
 * per read walks its seeds, after a lane per SEED has fetched what the walk needs from the reference
 * (round 5: see mgChainResolveKernel).  What comes back to the host is a few integers per read and per
 * block; the host only formats the lines.
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

/* Chaining in two steps.  Walking a read's seeds is serial (every step looks at the block built so far), but what a step needs
 * from the reference -- the seed's copy class, its first occurrence, its second one, the sequences they lie on -- depends on the seed
 * alone.  A lane per read that fetched them on the way (info[x] -> loc[x] -> rev[..] -> id[..]: four dependent random loads a seed)
 * spent 120 ms per 10 Gbp of ONT-like reads, the whole of it latency: the 3000 seeds of a 200 kb read one after the other.  So:
 *   resolve   a lane per SEED: copy class and CSR offset from one 8-byte word (li[x] = loc | info << 32), the first two occurrences
 *             with their sequence ids from one or two adjacent 8-byte words (revid[j] = rev[j] | id[rev[j]] << 32) -- two dependent
 *             random loads, all seeds of the batch in flight at once -- left as ONE 16-byte record per seed:
 *             {loc1, id1 | flags << 29, loc2, id2}, flags = copy class | hit << 2;
 *   chain     a lane per READ walks its records, which lie one after the other: no load of a step depends on the step before
 *             (the sequence id of the block's first occurrence travels in a register), so sixteen records are fetched at a time --
 *             one memory round trip per sixteen seeds instead of four per seed. */
#define MG_SEED_HIT 4u       /* flags of a seed: bits 0-1 copy class, bit 2 = the k-mer is in the modset */
#define MG_SEED_FLAG_SHIFT 29
__global__ __launch_bounds__ (256)
void mgChainResolveKernel (const U32 *__restrict__ seedIx, U64 nSeeds, const U64 *__restrict__ li, const U64 *__restrict__ revid, U32 refMax,
                           uint4 *__restrict__ rec)
{
  const U64 stride = (U64) gridDim.x * blockDim.x;
  for (U64 i = (U64) blockIdx.x * blockDim.x + threadIdx.x ; i < nSeeds ; i += stride)
    { const U32 x = seedIx[i];
      uint4 r = make_uint4 (0, 0, 0, 0);
      if (x)
        { const U64 v = li[x];
          const U32 c = (U32) (v >> 32) & 3u, l = (U32) v;
          r.y = (MG_SEED_HIT | c) << MG_SEED_FLAG_SHIFT;
          if (c != 3)                                                /* (copy M: counted, never chained: modmap.c:216) */
            { const U64 a = revid[l];
              r.x = (U32) a; r.y |= (U32) (a >> 32);
              if (c != 1) { const U64 b = revid[l + 1 < refMax ? l + 1 : refMax]; r.z = (U32) b; r.w = (U32) (b >> 32); }      /* the second copy, for the retry of modmap.c:242-254 (slot refMax is defined slack) */
            }
        }
      rec[i] = r;
    }
}

/* one end-of-block test, modmap.c:232-241 (repeated at :245-254 without the "no block" clause); idL / id0: the sequences of loc / loc0 */
__device__ __forceinline__ bool mgBlockEnds (U32 idL, U32 id0, U32 loc, U32 loc0, U32 locN, U32 i0, U32 iN, bool withUnset)
{
  if (withUnset && !loc0) return true;
  if (idL != id0) return true;
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

/* the walk's state and one step of it (modmap.c:213-258 for seed i of the read, its record in x y z w); both kernels below run this */
struct MgChainState { MgChainQ qq; U32 loc0, locN, i0, iN, id0; int n1, n2; };
struct MgChainOut { const U32 *seedPos; const U32 *refOffset; MgChainM *mine; U32 maxM; U32 *overflow; U64 s0; bool writer; };
__device__ __forceinline__ void mgChainEmit (MgChainState &st, const MgChainOut &o)
{
  if (o.writer)
    { if (st.qq.nM < o.maxM)
        { MgChainM e; e.pos0 = o.seedPos[o.s0 + st.i0] & MG_POS_MASK; e.posN = o.seedPos[o.s0 + st.iN] & MG_POS_MASK;
          e.id0 = st.id0; e.off0 = o.refOffset[st.loc0]; e.offN = o.refOffset[st.locN]; e.n1 = st.n1; e.n2 = st.n2;
          e.span = st.locN > st.loc0 ? st.locN - st.loc0 : st.loc0 - st.locN; o.mine[st.qq.nM] = e;
        }
      else *o.overflow = 1;
    }
  ++st.qq.nM;
}
__device__ __forceinline__ void mgChainStep (MgChainState &st, const MgChainOut &o, U32 i, U32 x, U32 y, U32 z, U32 w)
{
  const U32 f = y >> MG_SEED_FLAG_SHIFT;
  if (!(f & MG_SEED_HIT)) { ++st.qq.missed; return; }
  const U32 c = f & 3u;
  if (c == 1) ++st.qq.copy1; else if (c == 2) ++st.qq.copy2; else if (c == 3) ++st.qq.copyM;
  if (c == 3) return;
  U32 loc = x, idL = y & ((1u << MG_SEED_FLAG_SHIFT) - 1);
  const bool is1 = c == 1;
  bool end = mgBlockEnds (idL, st.id0, loc, st.loc0, st.locN, st.i0, st.iN, true);
  if (end && st.loc0 && !is1)
    { loc = z; idL = w;
      end = mgBlockEnds (idL, st.id0, loc, st.loc0, st.locN, st.i0, st.iN, false);
    }
  if (end)
    { if (st.n1 > 2) mgChainEmit (st, o);                           /* a block is reported when it ENDS only with more than two copy-1 seeds (modmap.c:256) */
      st.n1 = st.n2 = 0; st.loc0 = loc; st.id0 = idL; st.i0 = i;
    }
  if (is1) ++st.n1; else ++st.n2;
  st.locN = loc; st.iN = i;
}
__device__ __forceinline__ void mgChainBegin (MgChainState &st, U32 ns)
{ st.qq.nSeeds = ns; st.qq.missed = 0; st.qq.copy1 = st.qq.copy2 = st.qq.copyM = 0; st.qq.nM = 0; st.loc0 = st.locN = st.i0 = st.iN = st.id0 = 0; st.n1 = st.n2 = 0; }

/* a LANE per read.  (A wave per long read -- 64 records a fetch, the walk on wave-uniform values -- was built and measured: no
   faster.  What a batch's chaining costs is the serial walk of its LONGEST read, 3000 dependent steps of ~0.15 us for 200 kb,
   however the steps are issued; the file entry point therefore makes few, large batches of long reads: mg_seqio.c.) */
#define MG_CHAIN_AHEAD 16
__global__ __launch_bounds__ (256)
void mgChainKernel (const uint4 *__restrict__ rec, const U32 *__restrict__ seedPos,
                    const U64 *__restrict__ seedStart, U32 nReads, const U32 *__restrict__ refOffset, MgChainQ *__restrict__ q, MgChainM *__restrict__ mRec, U32 maxM,
                    U32 *__restrict__ overflow)
{
  const U32 r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nReads) return;
  const U64 s0 = seedStart[r], s1 = seedStart[r + 1];
  const U32 ns = (U32) (s1 - s0);
  MgChainState st; mgChainBegin (st, ns);
  MgChainOut o; o.seedPos = seedPos; o.refOffset = refOffset; o.mine = mRec + (U64) r * maxM; o.maxM = maxM; o.overflow = overflow; o.s0 = s0; o.writer = true;
  const uint4 *mineRec = rec + s0;
  for (U32 base = 0 ; base < ns ; base += MG_CHAIN_AHEAD)
    { uint4 v[MG_CHAIN_AHEAD];
#pragma unroll
      for (int j = 0 ; j < MG_CHAIN_AHEAD ; ++j) v[j] = base + j < ns ? mineRec[base + j] : make_uint4 (0, 0, 0, 0);
#pragma unroll
      for (int j = 0 ; j < MG_CHAIN_AHEAD ; ++j) if (base + j < ns) mgChainStep (st, o, base + j, v[j].x, v[j].y, v[j].z, v[j].w);
    }
  if (st.n2 > 2) mgChainEmit (st, o);                               /* the block open at the end of the read: only with more than two copy-2 seeds (modmap.c:269) */
  q[r] = st.qq;
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
enum { CS_IX, CS_POS, CS_RID, CS_START, CS_Q, CS_M, CS_OV, CS_MC, CS_REC, CS_N };
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
    uint4 *dRec = (uint4 *) csGet (CS_REC, ((size_t) n + 1) * sizeof (uint4));
    if (!dStart || !dQ || !dM || !dOv || !dRec) break;
    if (hipMemset (dOv, 0, 4)) break;
    unsigned grid = (unsigned) ((n + 1 + 255) / 256); if (grid > 16384) grid = 16384;
    mgProfBegin (MG_K_CHAIN_RESOLVE, 0);                   /* (bench.py: chain_ms_per_10Gbp = resolve + chain) */
    hipLaunchKernelGGL (mgSeedStartKernel, dim3 (grid), dim3 (256), 0, 0, dRid, n, nReads, dStart);
    if (n) hipLaunchKernelGGL (mgChainResolveKernel, dim3 (grid), dim3 (256), 0, 0, dIx, n, d.li, d.revid, d.refMax, dRec);
    mgProfEnd (MG_K_CHAIN_RESOLVE, 0);
    mgProfBegin (MG_K_CHAIN, 0);
    hipLaunchKernelGGL (mgChainKernel, dim3 ((nReads + 255) / 256), dim3 (256), 0, 0, dRec, dPos, dStart, nReads, d.offset, dQ, dM, maxM, dOv);
    mgProfEnd (MG_K_CHAIN, 0);
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
/* modasm's readsetFileRead (modasm.c:161-188) for a batch of reads: per read the hits (modset index, top bit = forward), the 16-bit
 * distance of each hit to the previous hit of the read, hit / miss counts, and per mod the number of hits.
 *
 * A lane per SEED (rounds 1-4: a lane per read walked its seeds twice, the batch waiting for its longest read: 6.5 ms per 4 Gbp).  The
 * seeds are in read order, so the hits of the batch in seed order ARE the reads' hit lists one after the other: a hit's place is the
 * number of hits before it (tile counts, their scan, ballots inside a tile); its distance needs the position of the hit before it IN ITS
 * READ -- the last hit before it in the batch (a running maximum of hit indices, carried through the same tile scan) if that one is not
 * before the read's first seed, else 0 (modasm.c:172: lastPos starts at 0).  A read's list starts at the hits counted before its first
 * seed; a read without seeds takes the start of the next read that has some. */
#define MG_RS_TILE 4096           /* 256 threads x 16 rows; a wave takes 64 consecutive seeds of a row */
__global__ __launch_bounds__ (256)
void mgRsTileCountKernel (const U32 *__restrict__ seedIx, U64 n, U32 *__restrict__ tileCnt, U32 *__restrict__ tileLast)
{
  __shared__ U32 sC[4], sL[4];
  const U64 base = (U64) blockIdx.x * MG_RS_TILE;
  U32 c = 0, last = 0;                                              /* last: 1 + index (inside the batch) of the tile's last hit, 0: none */
  for (int j = 0 ; j < 16 ; ++j)
    { const U64 i = base + (U64) j * 256 + threadIdx.x;
      if (i < n && seedIx[i]) { ++c; last = (U32) i + 1; }         /* (rows go up: the last assignment is the largest) */
    }
  for (int o = 32 ; o ; o >>= 1) { c += __shfl_down (c, o); const U32 l2 = __shfl_down (last, o); last = l2 > last ? l2 : last; }
  if ((threadIdx.x & 63) == 0) { sC[threadIdx.x >> 6] = c; sL[threadIdx.x >> 6] = last; }
  __syncthreads ();
  if (!threadIdx.x)
    { tileCnt[blockIdx.x] = sC[0] + sC[1] + sC[2] + sC[3];
      U32 m = sL[0]; for (int q = 1 ; q < 4 ; ++q) m = sL[q] > m ? sL[q] : m;
      tileLast[blockIdx.x] = m;
    }
}
/* one workgroup: cnt[0 .. nTiles) -> exclusive sums, cnt[nTiles] = total; last[0 .. nTiles) -> the running maximum BEFORE each tile */
__global__ __launch_bounds__ (1024)
void mgRsTileScanKernel (U32 *__restrict__ cnt, U32 *__restrict__ last, U32 nTiles)
{
  __shared__ U32 sSum[1024], sMax[1024];
  const int tid = threadIdx.x;
  const U32 per = (nTiles + 1023) / 1024;
  U32 sum = 0, mx = 0;
  for (U32 q = 0 ; q < per ; ++q) { const U32 j = tid * per + q; if (j < nTiles) { sum += cnt[j]; mx = last[j] > mx ? last[j] : mx; } }
  sSum[tid] = sum; sMax[tid] = mx;
  __syncthreads ();
  for (int off = 1 ; off < 1024 ; off <<= 1)
    { const U32 a = tid >= off ? sSum[tid - off] : 0, b = tid >= off ? sMax[tid - off] : 0;
      __syncthreads ();
      sSum[tid] += a; sMax[tid] = b > sMax[tid] ? b : sMax[tid];
      __syncthreads ();
    }
  U32 run = sSum[tid] - sum, rm = tid ? sMax[tid - 1] : 0;
  for (U32 q = 0 ; q < per ; ++q)
    { const U32 j = tid * per + q;
      if (j < nTiles) { const U32 c = cnt[j], l = last[j]; cnt[j] = run; last[j] = rm; run += c; rm = l > rm ? l : rm; }
    }
  if (tid == 1023) cnt[nTiles] = sSum[1023];
}
__global__ __launch_bounds__ (256)
void mgRsWriteKernel (const U32 *__restrict__ seedIx, const U32 *__restrict__ seedPosF, const U32 *__restrict__ seedRid, const U64 *__restrict__ seedStart, U64 n,
                      const U32 *__restrict__ tileBase, const U32 *__restrict__ tilePrev,
                      U32 *__restrict__ firstHit /* [read]: hits before the read's first seed, for reads that have seeds */,
                      U32 *__restrict__ hit, unsigned short *__restrict__ dx, U32 *__restrict__ depthCount)
{
  __shared__ U32 sCnt[64], sLast[64];                                /* per (row, wave), in seed order: hits; 1 + index of the last hit */
  const U64 base = (U64) blockIdx.x * MG_RS_TILE;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const U64 below = ((U64) 1 << lane) - 1;
  U32 x[16];
#pragma unroll
  for (int j = 0 ; j < 16 ; ++j)
    { const U64 i = base + (U64) j * 256 + threadIdx.x;
      x[j] = i < n ? seedIx[i] : 0u;
      const U64 b = __ballot (x[j] != 0);
      if (!lane) { sCnt[j * 4 + w] = (U32) __popcll (b); sLast[j * 4 + w] = b ? (U32) (base + (U64) j * 256 + w * 64) + (63u - (U32) __clzll ((long long) b)) + 1u : 0u; }
    }
  __syncthreads ();
  if (threadIdx.x == 0)
    { U32 run = tileBase[blockIdx.x], rm = tilePrev[blockIdx.x];
      for (int q = 0 ; q < 64 ; ++q) { const U32 c = sCnt[q], l = sLast[q]; sCnt[q] = run; sLast[q] = rm; run += c; rm = l > rm ? l : rm; }
    }
  __syncthreads ();
#pragma unroll
  for (int j = 0 ; j < 16 ; ++j)
    { const U64 i = base + (U64) j * 256 + threadIdx.x;
      const U64 b = __ballot (x[j] != 0);
      if (i >= n) continue;
      const int g = j * 4 + w;
      const U32 before = sCnt[g] + (U32) __popcll (b & below);          /* hits of the batch before seed i */
      const U32 r = seedRid[i];
      if (i == 0 || seedRid[i - 1] != r) firstHit[r] = before;          /* the read's first seed */
      if (x[j])
        { const U64 lower = b & below;
          const U32 prev = lower ? (U32) (i - lane) + (63u - (U32) __clzll ((long long) lower)) + 1u : sLast[g];      /* 1 + index of the hit before this one in the batch, 0: none */
          const U32 pf = seedPosF[i];
          const int pos = (int) (pf & MG_POS_MASK);
          const int lastPos = (U64) prev > seedStart[r] ? (int) (seedPosF[prev - 1] & MG_POS_MASK) : 0;               /* (prev - 1 >= the read's first seed: a hit of this read) */
          hit[before] = (pf & MG_FWD_BIT) ? (x[j] | 0x80000000u) : x[j];                                              /* modasm.c:171 */
          dx[before] = (unsigned short) (pos - lastPos);                                                              /* modasm.c:172 */
          atomicAdd (&depthCount[x[j]], 1u);                                                                          /* modasm.c:174, saturated by the caller */
        }
    }
}
/* per read: where its list starts and how many of its seeds missed */
__global__ void mgRsPerReadKernel (const U64 *__restrict__ seedStart, const U32 *__restrict__ seedRid, const U32 *__restrict__ firstHit, U64 n, U32 totHit, U32 nReads,
                                   U64 *__restrict__ hitStart, U32 *__restrict__ nMiss)
{
  const U32 r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > nReads) return;
  const U64 s0 = seedStart[r];
  const U32 h0 = s0 < n ? firstHit[seedRid[s0]] : totHit;             /* (seed s0 is the first seed of the read that owns it: this read, or the next one that has seeds) */
  hitStart[r] = h0;
  if (r == nReads) return;
  const U64 s1 = seedStart[r + 1];
  const U32 h1 = s1 < n ? firstHit[seedRid[s1]] : totHit;
  nMiss[r] = (U32) (s1 - s0) - (h1 - h0);
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

/* hHitStart[nReads+1], hNMiss[nReads]: host, filled here.  *dHitOut / *dDxOut: DEVICE arrays of hHitStart[nReads] entries, hipMalloc ()ed here (the
 * caller copies them where they go and frees them with mgDeviceFree).  dDepthAccum: device U32[ms->max + 1], the hits per mod of the file so
 * far (mg_refpack.hip keeps it across the batches): this batch's are added.  Returns 0, -1 on error. */
extern "C" int mgReadsetSeedsDevice (Modset *ms, const U32 *dPacked, U64 totalBases, const U64 *dReadOffsets, U32 nReads,
                                     U64 *hHitStart, U32 *hNMiss, U32 **dHitOut, unsigned short **dDxOut, U32 *dDepthAccum)
{
  *dHitOut = 0; *dDxOut = 0;
  const bool lapOn = mgKnobs ()->seedTiming == 1;          /* dev */
  struct timespec lq0; clock_gettime (CLOCK_MONOTONIC, &lq0);
#define RS_LAP(what) do { if (lapOn) { (void) hipDeviceSynchronize (); struct timespec q_; clock_gettime (CLOCK_MONOTONIC, &q_); fprintf (stderr, "mgReadsetSeedsDevice: %s at %.2f ms\n", what, (q_.tv_sec - lq0.tv_sec) * 1e3 + (q_.tv_nsec - lq0.tv_nsec) * 1e-6); } } while (0)
  U64 guess = totalBases / (U64) ms->hasher->w; guess += guess / 2 + 65536; if (guess > totalBases + 1) guess = totalBases + 1;
  U32 *dIx = 0, *dPos = 0, *dRid = 0, *dMiss = 0, *dHit = 0, *dFirst = 0, *dTileCnt = 0, *dTileLast = 0; unsigned short *dDx = 0; U64 *dStart = 0, *dHitStart = 0;
  U64 n = 0;
  int rc = -1;
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
    RS_LAP ("seeds");
    if (n >= ((U64) 1 << 32) - 1) { mgSetError ("too many seeds in one batch"); break; }
    const U32 nTiles = (U32) ((n + MG_RS_TILE - 1) / MG_RS_TILE);
    if (hipMalloc ((void **) &dStart, ((size_t) nReads + 2) * 8) || hipMalloc ((void **) &dHitStart, ((size_t) nReads + 2) * 8)
        || hipMalloc ((void **) &dMiss, ((size_t) nReads + 1) * 4) || hipMalloc ((void **) &dFirst, ((size_t) nReads + 1) * 4)
        || hipMalloc ((void **) &dTileCnt, ((size_t) nTiles + 2) * 4) || hipMalloc ((void **) &dTileLast, ((size_t) nTiles + 2) * 4)) break;
    unsigned grid = (unsigned) ((n + 1 + 255) / 256); if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL (mgSeedStartKernel, dim3 (grid), dim3 (256), 0, 0, dRid, n, nReads, dStart);
    U32 totHit32 = 0;
    if (nTiles)
      { hipLaunchKernelGGL (mgRsTileCountKernel, dim3 (nTiles), dim3 (256), 0, 0, dIx, n, dTileCnt, dTileLast);
        hipLaunchKernelGGL (mgRsTileScanKernel, dim3 (1), dim3 (1024), 0, 0, dTileCnt, dTileLast, nTiles);
        if (hipMemcpy (&totHit32, dTileCnt + nTiles, 4, hipMemcpyDeviceToHost)) break;
      }
    const U64 totHit = totHit32;
    RS_LAP ("counted");
    if (hipMalloc ((void **) &dHit, (totHit + 1) * 4) || hipMalloc ((void **) &dDx, (totHit + 1) * 2)) break;
    if (nTiles) hipLaunchKernelGGL (mgRsWriteKernel, dim3 (nTiles), dim3 (256), 0, 0, dIx, dPos, dRid, dStart, n, dTileCnt, dTileLast, dFirst, dHit, dDx, dDepthAccum);
    hipLaunchKernelGGL (mgRsPerReadKernel, dim3 (nReads / 256 + 1), dim3 (256), 0, 0, dStart, dRid, dFirst, n, totHit32, nReads, dHitStart, dMiss);
    if (hipGetLastError () != hipSuccess) break;
    if (hipMemcpy (hHitStart, dHitStart, ((size_t) nReads + 1) * 8, hipMemcpyDeviceToHost) || hipMemcpy (hNMiss, dMiss, (size_t) nReads * 4, hipMemcpyDeviceToHost)) break;
    *dHitOut = dHit; *dDxOut = dDx; dHit = 0; dDx = 0;
    RS_LAP ("written");
    rc = 0;
  } while (0);
  (void) hipFree (dIx); (void) hipFree (dPos); (void) hipFree (dRid); (void) hipFree (dStart); (void) hipFree (dHitStart);
  (void) hipFree (dMiss); (void) hipFree (dHit); (void) hipFree (dDx); (void) hipFree (dFirst); (void) hipFree (dTileCnt); (void) hipFree (dTileLast);
  RS_LAP ("freed");
#undef RS_LAP
  if (rc < 0 && !mgLastError ()[0]) mgSetError ("readset seeds on the device failed");
  return rc;
}
