/* mg_refpack.hip — modmap's Reference built on the device: the bookkeeping of referenceFastaRead
 * (modmap.c:106-118: per occurrence index / offset / id, ++depth[index]), the copy classes (modmap.c:125-129) and
 * referencePack (modmap.c:74-91: loc = exclusive sums of depth, rev = the occurrences grouped by modset index, in
 * occurrence order inside an index).
 *
 * The seeds of a batch -- (index, pos, sequence) per modimizer, in (sequence, pos) order -- are dense on the device when
 * mgInsertReadsDevice / mgQueryReadsDevice return.  They stay there:
 *   append    ordered compaction of the seeds with index != 0 (modmap.c:110) behind what the reference holds already,
 *             pos masked, id = first id of the batch + sequence ordinal, depth counted with atomics (a count: order free);
 *   finish    info[] classified from depth (one pass, the three tallies reduced per workgroup); loc[] by a device-wide
 *             exclusive scan; rev[] by a STABLE least-significant-digit radix sort of (index, occurrence ordinal), 8 bits a pass:
 *             stability is what keeps the occurrences of one index in occurrence order, which modmap.c:86-90 produces by
 *             walking the occurrences in order and queryProcess relies on (rev[loc[x]] is the FIRST occurrence, rev[loc[x] + 1]
 *             the second: modmap.c:219-221,242-254).  Ranks inside a tile come from wave-level matching (8 ballots give every
 *             lane the set of lanes that hold its digit), so a reference that is one k-mer a million times over sorts at the
 *             speed of any other;
 *   mirror    the six arrays go back to the host's Reference in one piece each (mg_xfer.hip); what stays resident is what
 *             mg_chain.hip's chaining reads, derived from them: li[x] = loc[x] | info[x] << 32 and revid[j] = rev[j] | id[rev[j]] << 32
 *             (one 8-byte load gives a seed its copy class and CSR offset, one or two adjacent ones its first two occurrences with
 *             the sequences they lie on) and offset[] -- no re-upload.
 * A Reference whose arrays came from a file (mgReferenceLoad) or were written by the host is uploaded and the same words derived.
 * The second half of the file is modasm's read ingest (N3): the same scan and sort make its inverse lists.
 */
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <mutex>
#include <unordered_map>
#include "mg_common.h"
#include "mg_internal.h"
#include "mg_xfer.h"
#include "mg_ref.h"

/* the device side of every Reference, by its address.  The lock guards the MAP (finding, making and dropping an entry); an entry itself belongs
   to the Reference, which -- like the reference's own struct -- one host thread works on at a time: builds of two References on two GPUs run
   side by side.  (An unordered_map keeps its elements where they are when it grows: the entry found under the lock stays valid after it.) */
static std::mutex gRefLock;
static std::unordered_map<const MgReference *, MgRefDev> gRefDev;
static std::unordered_map<const MgReference *, std::mutex> gRefOwn;      /* one lock per Reference: two threads that query ONE Reference do not both build its device side */
static MgRefDev &mgRefEntry (const MgReference *ref) { std::lock_guard<std::mutex> g (gRefLock); return gRefDev[ref]; }
static std::mutex &mgRefOwnLock (const MgReference *ref) { std::lock_guard<std::mutex> g (gRefLock); return gRefOwn[ref]; }

static void mgRefDevFree (MgRefDev &d)
{ (void) hipFree (d.info); (void) hipFree (d.loc); (void) hipFree (d.rev); (void) hipFree (d.id); (void) hipFree (d.offset);
  (void) hipFree (d.index); (void) hipFree (d.depth); (void) hipFree (d.li); (void) hipFree (d.revid); d = MgRefDev ();
}

/* li[] and revid[] (mg_ref.h) from info / loc / rev / id; those four are freed: the chaining reads the derived words only */
__global__ void mgRefDeriveLiKernel (const U8 *__restrict__ info, const U32 *__restrict__ loc, U64 m, U64 *__restrict__ li)
{ for (U64 i = (U64) blockIdx.x * blockDim.x + threadIdx.x ; i < m ; i += (U64) gridDim.x * blockDim.x) li[i] = (U64) loc[i] | ((U64) info[i] << 32); }
__global__ void mgRefDeriveRevidKernel (const U32 *__restrict__ rev, const U32 *__restrict__ id, U64 n, U64 *__restrict__ revid)      /* n = ref->max; slot n: occurrence 0 */
{ for (U64 j = (U64) blockIdx.x * blockDim.x + threadIdx.x ; j <= n ; j += (U64) gridDim.x * blockDim.x) { const U32 r = j < n ? rev[j] : 0u; revid[j] = (U64) r | ((U64) (n ? id[r] : 0u) << 32); } }
static MgStatus mgRefDerive (MgRefDev &d, U32 msMax, U32 refMax, hipStream_t st)
{
  const U64 m = (U64) msMax + 1, n = refMax;
  if (d.nSeq >= (1 << 29)) { mgSetError ("more than 2^29 reference sequences"); return MG_ERR_ARG; }      /* (the chaining's seed records keep three flag bits above the sequence id) */
  (void) hipFree (d.li); d.li = 0; (void) hipFree (d.revid); d.revid = 0;
  MG_HIP (hipMalloc ((void **) &d.li, m * 8));
  MG_HIP (hipMalloc ((void **) &d.revid, (n + 1) * 8));
  hipLaunchKernelGGL (mgRefDeriveLiKernel, dim3 (2048), dim3 (256), 0, st, d.info, d.loc, m, d.li);
  hipLaunchKernelGGL (mgRefDeriveRevidKernel, dim3 (2048), dim3 (256), 0, st, d.rev, d.id, n, d.revid);
  MG_HIP (hipGetLastError ());
  MG_HIP (hipStreamSynchronize (st));
  (void) hipFree (d.info); d.info = 0; (void) hipFree (d.loc); d.loc = 0; (void) hipFree (d.rev); d.rev = 0; (void) hipFree (d.id); d.id = 0;
  return MG_OK;
}

extern "C" void mgChainForget (const MgReference *ref)
{
  std::lock_guard<std::mutex> g (gRefLock);
  auto it = gRefDev.find (ref);
  if (it != gRefDev.end ()) { mgRefDevFree (it->second); gRefDev.erase (it); }
  gRefOwn.erase (ref);                                 /* (the Reference is going: nobody holds its lock) */
}

/* device copies of what the chaining reads: the ones the builder left, or uploaded from the host arrays (a Reference read from a
   file, or one whose modset has grown since) */
MgStatus mgRefDevGet (const MgReference *ref, MgRefDev *out)
{
  std::lock_guard<std::mutex> own (mgRefOwnLock (ref));
  MgRefDev &d = mgRefEntry (ref);
  const U32 msMax = ref->ms->max, refMax = ref->max;
  int cur = 0; MG_HIP (hipGetDevice (&cur));
  if (d.li && d.packed && d.msMax == msMax && d.refMax == refMax && d.device == cur) { *out = d; return MG_OK; }
  mgRefDevFree (d);                                    /* (also: made on another GPU than the one the calling thread is on now -- uploaded again, here) */
  d.device = cur;
  const size_t m = (size_t) msMax + 1, n = refMax ? refMax : 1;
  MG_HIP (hipMalloc ((void **) &d.info, m));
  MG_HIP (hipMalloc ((void **) &d.loc, m * 4));
  MG_HIP (hipMalloc ((void **) &d.rev, (n + 1) * 4));
  MG_HIP (hipMalloc ((void **) &d.id, n * 4));
  MG_HIP (hipMalloc ((void **) &d.offset, n * 4));
  MG_HIP (hipMemset (d.rev, 0, (n + 1) * 4));
  MG_HIP (hipDeviceSynchronize ());
  MgStatus s;
  if ((s = mgXferH2D (d.info, ref->ms->info, m)) || (s = mgXferH2D (d.loc, ref->loc, m * 4))) return s;
  if (refMax && ((s = mgXferH2D (d.rev, ref->rev, (size_t) refMax * 4)) || (s = mgXferH2D (d.id, ref->id, (size_t) refMax * 4))
                 || (s = mgXferH2D (d.offset, ref->offset, (size_t) refMax * 4)))) return s;
  d.nSeq = ref->nSeq;
  if ((s = mgRefDerive (d, msMax, refMax, 0))) return s;
  d.msMax = msMax; d.refMax = refMax; d.packed = true;
  *out = d;
  return MG_OK;
}

/* ---------------------------------------------------------------------------------------- */
/* device-wide exclusive scan of U32 (sums stay below 2^32: they count occurrences)           */

#define MG_SCAN_TILE 4096          /* 256 threads x 16 */
__global__ __launch_bounds__ (256)
void mgRefTileSumKernel (const U32 *__restrict__ in, U64 n, U32 *__restrict__ tileSum)
{
  __shared__ U32 sW[4];
  const U64 base = (U64) blockIdx.x * MG_SCAN_TILE;
  U32 s = 0;
  for (int j = 0 ; j < 16 ; ++j) { const U64 i = base + (U64) j * 256 + threadIdx.x; if (i < n) s += in[i]; }
  for (int o = 32 ; o ; o >>= 1) s += __shfl_down (s, o);
  if ((threadIdx.x & 63) == 0) sW[threadIdx.x >> 6] = s;
  __syncthreads ();
  if (!threadIdx.x) tileSum[blockIdx.x] = sW[0] + sW[1] + sW[2] + sW[3];
}
/* one workgroup: a[0 .. n) exclusive in place, a[n] = the total */
__global__ __launch_bounds__ (1024)
void mgRefScanSmallKernel (U32 *__restrict__ a, U32 n)
{
  __shared__ U32 sPart[1024];
  const int tid = threadIdx.x;
  const U32 per = (n + 1023) / 1024;
  U32 sum = 0;
  for (U32 i = 0 ; i < per ; ++i) { const U32 j = tid * per + i; if (j < n) sum += a[j]; }
  sPart[tid] = sum;
  __syncthreads ();
  for (int off = 1 ; off < 1024 ; off <<= 1)
    { const U32 v = tid >= off ? sPart[tid - off] : 0;
      __syncthreads ();
      sPart[tid] += v;
      __syncthreads ();
    }
  U32 run = sPart[tid] - sum;
  for (U32 i = 0 ; i < per ; ++i) { const U32 j = tid * per + i; if (j < n) { const U32 c = a[j]; a[j] = run; run += c; } }
  if (tid == 1023) a[n] = sPart[1023];
}
__global__ __launch_bounds__ (256)
void mgRefTileScanKernel (const U32 *__restrict__ in, U64 n, const U32 *__restrict__ tileBase, U32 *__restrict__ out)
{
  __shared__ U32 sT[256];
  const U64 base = (U64) blockIdx.x * MG_SCAN_TILE + (U64) threadIdx.x * 16;      /* a thread's 16 items are consecutive */
  U32 v[16]; U32 s = 0;
#pragma unroll
  for (int j = 0 ; j < 16 ; ++j) { v[j] = base + j < n ? in[base + j] : 0u; s += v[j]; }
  sT[threadIdx.x] = s;
  __syncthreads ();
  for (int off = 1 ; off < 256 ; off <<= 1)
    { const U32 x = (int) threadIdx.x >= off ? sT[threadIdx.x - off] : 0;
      __syncthreads ();
      sT[threadIdx.x] += x;
      __syncthreads ();
    }
  U32 run = tileBase[blockIdx.x] + sT[threadIdx.x] - s;
#pragma unroll
  for (int j = 0 ; j < 16 ; ++j) { if (base + j < n) out[base + j] = run; run += v[j]; }
}
/* out[i] = sum of in[0 .. i) for i < n (in == out allowed); tiles: (n / 4096 + 2) words of scratch; the total is left in tiles[nTiles] */
static MgStatus mgRefExclusiveScan (const U32 *in, U32 *out, U64 n, U32 *tiles, hipStream_t st)
{
  if (!n) return MG_OK;
  const U32 nTiles = (U32) ((n + MG_SCAN_TILE - 1) / MG_SCAN_TILE);
  hipLaunchKernelGGL (mgRefTileSumKernel, dim3 (nTiles), dim3 (256), 0, st, in, n, tiles);
  hipLaunchKernelGGL (mgRefScanSmallKernel, dim3 (1), dim3 (1024), 0, st, tiles, nTiles);
  hipLaunchKernelGGL (mgRefTileScanKernel, dim3 (nTiles), dim3 (256), 0, st, in, n, tiles, out);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}

/* ---------------------------------------------------------------------------------------- */
/* append: modmap.c:110-117 for a batch of seeds                                              */

__global__ __launch_bounds__ (256)
void mgRefCountHitsKernel (const U32 *__restrict__ ix, U64 n, U32 *__restrict__ tileCount)
{
  __shared__ U32 sW[4];
  const U64 base = (U64) blockIdx.x * MG_SCAN_TILE;
  U32 s = 0;
  for (int j = 0 ; j < 16 ; ++j) { const U64 i = base + (U64) j * 256 + threadIdx.x; if (i < n && ix[i]) ++s; }
  for (int o = 32 ; o ; o >>= 1) s += __shfl_down (s, o);
  if ((threadIdx.x & 63) == 0) sW[threadIdx.x >> 6] = s;
  __syncthreads ();
  if (!threadIdx.x) tileCount[blockIdx.x] = sW[0] + sW[1] + sW[2] + sW[3];
}
/* tile b's hits go to at0 + tileBase[b] .. in order: inside the tile a wave takes 64 consecutive seeds at a time (ballot + popcount),
   the four waves one after the other over the tile's 16 rows of 256 */
__global__ __launch_bounds__ (256)
void mgRefAppendKernel (const U32 *__restrict__ ix, const U32 *__restrict__ posF, const U32 *__restrict__ rid, U64 n, const U32 *__restrict__ tileBase,
                        U32 at0, U32 idBase, U32 *__restrict__ index, U32 *__restrict__ offset, U32 *__restrict__ id, U32 *__restrict__ depth)
{
  __shared__ U32 sRow[64];                                         /* hits per (row of 256, wave): 16 x 4 */
  const U64 base = (U64) blockIdx.x * MG_SCAN_TILE;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  U32 x[16];
#pragma unroll
  for (int j = 0 ; j < 16 ; ++j)
    { const U64 i = base + (U64) j * 256 + threadIdx.x;
      x[j] = i < n ? ix[i] : 0u;
      const U64 b = __ballot (x[j] != 0);
      if (!lane) sRow[j * 4 + w] = (U32) __popcll (b);
    }
  __syncthreads ();
  if (threadIdx.x == 0) { U32 run = 0; for (int q = 0 ; q < 64 ; ++q) { const U32 c = sRow[q]; sRow[q] = run; run += c; } }
  __syncthreads ();
  const U32 tb = at0 + tileBase[blockIdx.x];
#pragma unroll
  for (int j = 0 ; j < 16 ; ++j)
    { const U64 i = base + (U64) j * 256 + threadIdx.x;
      const U64 b = __ballot (x[j] != 0);
      if (x[j])
        { const U32 at = tb + sRow[j * 4 + w] + (U32) __popcll (b & (((U64) 1 << lane) - 1));
          index[at] = x[j]; offset[at] = posF[i] & MG_POS_MASK; id[at] = idBase + rid[i];
          atomicAdd (&depth[x[j]], 1u);
        }
    }
}

/* ---------------------------------------------------------------------------------------- */
/* finish: copy classes, loc, rev                                                             */

/* modmap.c:125-129: depth 1 -> copy 1, 2 -> copy 2, anything else (0 included) -> copy M; entries 1 .. max; tallies[3] */
__global__ __launch_bounds__ (256)
void mgRefClassifyKernel (const U32 *__restrict__ depth, U32 max, U8 *__restrict__ info, U32 *__restrict__ tallies)
{
  __shared__ U32 sT[3];
  if (threadIdx.x < 3) sT[threadIdx.x] = 0;
  __syncthreads ();
  U32 c1 = 0, c2 = 0, cM = 0;
  for (U64 i = 1 + (U64) blockIdx.x * blockDim.x + threadIdx.x ; i <= max ; i += (U64) gridDim.x * blockDim.x)
    { const U32 dp = depth[i]; const U8 f = info[i];
      if (dp == 1) { info[i] = (U8) ((f & 0xfc) | 1); ++c1; }
      else if (dp == 2) { info[i] = (U8) ((f & 0xfc) | 2); ++c2; }
      else { info[i] = (U8) (f | 3); ++cM; }
    }
  for (int o = 32 ; o ; o >>= 1) { c1 += __shfl_down (c1, o); c2 += __shfl_down (c2, o); cM += __shfl_down (cM, o); }
  if ((threadIdx.x & 63) == 0) { atomicAdd (&sT[0], c1); atomicAdd (&sT[1], c2); atomicAdd (&sT[2], cM); }
  __syncthreads ();
  if (threadIdx.x < 3 && sT[threadIdx.x]) atomicAdd (&tallies[threadIdx.x], sT[threadIdx.x]);
}

/* one pass of the stable radix sort: a workgroup's tile is 8192 consecutive elements, wave w's part of it elements
   [2048 w, 2048 (w + 1)), taken 64 at a time in order.  hist[digit * nTiles + tile]: digit-major, so that ONE exclusive scan over
   the whole array gives every (digit, tile) its place in the output. */
#define MG_RSORT_TILE 8192
#define MG_RSORT_WAVE (MG_RSORT_TILE / 4)
__global__ __launch_bounds__ (256)
void mgRefSortHistKernel (const U32 *__restrict__ keys, U64 n, int shift, U32 *__restrict__ hist, U32 nTiles)
{
  __shared__ U32 sC[256];
  sC[threadIdx.x] = 0;
  __syncthreads ();
  const U64 base = (U64) blockIdx.x * MG_RSORT_TILE;
  for (int j = 0 ; j < MG_RSORT_TILE / 256 ; ++j)
    { const U64 i = base + (U64) j * 256 + threadIdx.x;
      if (i < n) atomicAdd (&sC[(keys[i] >> shift) & 255u], 1u);
    }
  __syncthreads ();
  hist[(U64) threadIdx.x * nTiles + blockIdx.x] = sC[threadIdx.x];
}
template <bool FIRST>        /* FIRST: the values are the elements' own positions (the occurrence ordinals) */
__global__ __launch_bounds__ (256)
void mgRefSortScatterKernel (const U32 *__restrict__ keys, const U32 *__restrict__ vals, U64 n, int shift, const U32 *__restrict__ place, U32 nTiles,
                             U32 *__restrict__ keysOut, U32 *__restrict__ valsOut)
{
  __shared__ U32 sC[4][256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int q = 0 ; q < 4 ; ++q) sC[q][threadIdx.x] = 0;
  __syncthreads ();
  const U64 wbase = (U64) blockIdx.x * MG_RSORT_TILE + (U64) w * MG_RSORT_WAVE;
  for (int r = 0 ; r < MG_RSORT_WAVE / 64 ; ++r)
    { const U64 i = wbase + (U64) r * 64 + lane;
      if (i < n) atomicAdd (&sC[w][(keys[i] >> shift) & 255u], 1u);
    }
  __syncthreads ();
  { U32 run = place[(U64) threadIdx.x * nTiles + blockIdx.x];        /* digit threadIdx.x: where the tile's first such element goes; then wave by wave */
    for (int q = 0 ; q < 4 ; ++q) { const U32 c = sC[q][threadIdx.x]; sC[q][threadIdx.x] = run; run += c; }
  }
  __syncthreads ();
  for (int r = 0 ; r < MG_RSORT_WAVE / 64 ; ++r)
    { const U64 i = wbase + (U64) r * 64 + lane;
      const bool live = i < n;
      const U32 key = live ? keys[i] : 0u;
      const U32 dg = (key >> shift) & 255u;
      U64 peers = __ballot (live);                                   /* the lanes that hold my digit */
#pragma unroll
      for (int b = 0 ; b < 8 ; ++b)
        { const U64 m = __ballot ((dg >> b) & 1u);
          peers &= ((dg >> b) & 1u) ? m : ~m;
        }
      const U32 before = (U32) __popcll (peers & (((U64) 1 << lane) - 1));
      const U32 old = sC[w][dg];                                      /* every peer reads the same counter ... */
      __builtin_amdgcn_wave_barrier ();
      if (live && before == 0) sC[w][dg] = old + (U32) __popcll (peers);      /* ... and the first of them moves it on (one wave, LDS in order) */
      __builtin_amdgcn_wave_barrier ();
      if (live)
        { const U32 at = old + before;
          if (keysOut) keysOut[at] = key;
          valsOut[at] = FIRST ? (U32) i : vals[i];
        }
    }
}

/* ---------------------------------------------------------------------------------------- */

/* values (vals, or the positions 0 .. n-1 when vals == 0) in the order of their keys, equal keys in their original order: LSD passes of
   8 bits over keyBits bits.  *out: a fresh device array of n + 1 words (the caller frees it); tiles: scratch for the scans, at least
   256 * ceil (n / 8192) / 4096 + 4 words. */
static MgStatus mgRefStableSort (const U32 *keys, const U32 *vals, U32 n, int keyBits, U32 **out, U32 *tiles, hipStream_t st)
{
  *out = 0;
  const int passes = (keyBits + 7) / 8 > 0 ? (keyBits + 7) / 8 : 1;
  const U32 nSortTiles = (U32) (((U64) n + MG_RSORT_TILE - 1) / MG_RSORT_TILE);
  const size_t histWords = (size_t) 256 * (nSortTiles ? nSortTiles : 1);
  U32 *k1 = 0, *k2 = 0, *v1 = 0, *v2 = 0, *hist = 0;
  MgStatus s = MG_ERR_HIP;
  do {
    if (hipMalloc ((void **) &hist, (histWords + 2) * 4) || hipMalloc ((void **) &v1, ((size_t) n + 1) * 4) || hipMalloc ((void **) &v2, ((size_t) n + 1) * 4)) break;
    if (passes > 1 && (hipMalloc ((void **) &k1, (size_t) n * 4) || (passes > 2 && hipMalloc ((void **) &k2, (size_t) n * 4)))) break;
    const U32 *kin = keys; const U32 *vin = vals;
    U32 *kout = k1, *vout = v1;
    bool bad = false;
    for (int p = 0 ; p < passes ; ++p)
      { const bool last = p + 1 == passes;
        hipLaunchKernelGGL (mgRefSortHistKernel, dim3 (nSortTiles), dim3 (256), 0, st, kin, (U64) n, 8 * p, hist, nSortTiles);
        if (mgRefExclusiveScan (hist, hist, histWords, tiles, st)) { bad = true; break; }
        if (!vin) hipLaunchKernelGGL (mgRefSortScatterKernel<true>, dim3 (nSortTiles), dim3 (256), 0, st, kin, vin, (U64) n, 8 * p, hist, nSortTiles, last ? (U32 *) 0 : kout, vout);
        else hipLaunchKernelGGL (mgRefSortScatterKernel<false>, dim3 (nSortTiles), dim3 (256), 0, st, kin, vin, (U64) n, 8 * p, hist, nSortTiles, last ? (U32 *) 0 : kout, vout);
        kin = kout; vin = vout;
        kout = kout == k1 ? k2 : k1; vout = vout == v1 ? v2 : v1;
      }
    if (bad || hipGetLastError () != hipSuccess || hipStreamSynchronize (st) != hipSuccess) break;
    *out = (U32 *) vin;                                             /* the last pass's output */
    if (*out == v1) v1 = 0; else v2 = 0;
    s = MG_OK;
  } while (0);
  (void) hipFree (k1); (void) hipFree (k2); (void) hipFree (v1); (void) hipFree (v2); (void) hipFree (hist);
  if (s == MG_ERR_HIP && !mgLastError ()[0]) mgHipFail (hipGetLastError (), "stable sort on the device");
  return s;
}

static MgStatus mgRefGrow (U32 **p, size_t have, size_t keep, size_t want)      /* device array of `want` words holding the first `keep` of the old one */
{
  (void) have;
  U32 *q = 0;
  MG_HIP (hipMalloc ((void **) &q, (want ? want : 1) * 4));
  if (*p && keep) MG_HIP (hipMemcpy (q, *p, keep * 4, hipMemcpyDeviceToDevice));
  if (*p) MG_HIP (hipFree (*p));
  *p = q;
  return MG_OK;
}

/* the seeds of a batch (device arrays, n of them, in order) behind the ref->max occurrences held so far; *appended = how many had an index */
extern "C" MgStatus mgRefBuildAppend (MgReference *ref, const U32 *dIx, const U32 *dPosF, const U32 *dRid, U64 n, U32 idBase, U32 *appended)
{
  *appended = 0;
  std::lock_guard<std::mutex> own (mgRefOwnLock (ref));
  MgRefDev &d = mgRefEntry (ref);
  hipStream_t st = 0;
  if (d.packed)
    { /* a further file read into a packed Reference: referencePack left size == max (modmap.c:80), so the reference dies on the FIRST seed
         that has an index (modmap.c:111) and otherwise goes on, adds nothing and reports again */
      if (!n) return MG_OK;
      const U32 nT = (U32) ((n + MG_SCAN_TILE - 1) / MG_SCAN_TILE);
      U32 *tl = 0, cnt = 0;
      MG_HIP (hipMalloc ((void **) &tl, ((size_t) nT + 2) * 4));
      hipLaunchKernelGGL (mgRefCountHitsKernel, dim3 (nT), dim3 (256), 0, st, dIx, n, tl);
      hipLaunchKernelGGL (mgRefScanSmallKernel, dim3 (1), dim3 (1024), 0, st, tl, nT);
      const bool bad = hipMemcpyAsync (&cnt, tl + nT, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize (st) != hipSuccess;
      (void) hipFree (tl);
      if (bad) return mgHipFail (hipGetLastError (), "reference append");
      if (cnt) { mgSetError ("reference size overflow"); return MG_ERR_CAPACITY; }
      return MG_OK;
    }
  { int cur = 0; MG_HIP (hipGetDevice (&cur));
    if (d.device >= 0 && d.device != cur && d.depth) { mgSetError ("this Reference is being built on GPU %d, the calling thread is on GPU %d", d.device, cur); return MG_ERR_ARG; }
    d.device = cur;
  }
  const size_t msCap = ref->ms->size;
  if (!d.depth)
    { MG_HIP (hipMalloc ((void **) &d.depth, (msCap + 1) * 4));
      MG_HIP (hipMemsetAsync (d.depth, 0, (msCap + 1) * 4, st));
      d.capMs = msCap;
      if (ref->max)                                                  /* occurrences the host holds already (a caller that filled the arrays itself) */
        { mgSetError ("mgRefBuildAppend: a reference with host-built occurrences cannot be extended on the device"); return MG_ERR_ARG; }
    }
  if (!n) return MG_OK;
  if (n >= ((U64) 1 << 32)) { mgSetError ("too many seeds in one batch"); return MG_ERR_ARG; }
  const U32 nTiles = (U32) ((n + MG_SCAN_TILE - 1) / MG_SCAN_TILE);
  U32 *tiles = 0;
  MG_HIP (hipMalloc ((void **) &tiles, ((size_t) nTiles + 2) * 4));
  MgStatus s = MG_OK;
  do {
    hipLaunchKernelGGL (mgRefCountHitsKernel, dim3 (nTiles), dim3 (256), 0, st, dIx, n, tiles);
    hipLaunchKernelGGL (mgRefScanSmallKernel, dim3 (1), dim3 (1024), 0, st, tiles, nTiles);
    U32 cnt = 0;
    if (hipMemcpyAsync (&cnt, tiles + nTiles, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize (st) != hipSuccess) { s = mgHipFail (hipGetLastError (), "reference append"); break; }
    /* modmap.c:111: an append is refused once max + 1 >= size */
    if (cnt && (U64) ref->max + cnt > (U64) ref->size - 1) { mgSetError ("reference size overflow"); s = MG_ERR_CAPACITY; break; }
    const size_t want = (size_t) ref->max + cnt;
    if (want > d.capOcc)
      { size_t cap = want + want / 2 + 65536; if (cap > ref->size) cap = ref->size; if (cap < want) cap = want;
        if ((s = mgRefGrow (&d.index, d.capOcc, ref->max, cap)) || (s = mgRefGrow (&d.offset, d.capOcc, ref->max, cap)) || (s = mgRefGrow (&d.id, d.capOcc, ref->max, cap))) break;
        d.capOcc = cap;
      }
    if (cnt)
      { hipLaunchKernelGGL (mgRefAppendKernel, dim3 (nTiles), dim3 (256), 0, st, dIx, dPosF, dRid, n, tiles, ref->max, idBase, d.index, d.offset, d.id, d.depth);
        if (hipGetLastError () != hipSuccess || hipStreamSynchronize (st) != hipSuccess) { s = mgHipFail (hipGetLastError (), "reference append"); break; }
      }
    *appended = cnt;
  } while (0);
  (void) hipFree (tiles);
  return s;
}

/* 1 (and the tallies of its copy classes) if ref is packed on the device and nothing came since: what a further mgReferenceFinish reports again */
extern "C" int mgRefPackedTallies (MgReference *ref, U32 tallies[3])
{
  std::lock_guard<std::mutex> own (mgRefOwnLock (ref));
  MgRefDev &d = mgRefEntry (ref);
  if (!d.packed || d.refMax != ref->max || d.msMax != ref->ms->max) return 0;
  tallies[0] = d.tallies[0]; tallies[1] = d.tallies[1]; tallies[2] = d.tallies[2];
  return 1;
}

/* modmap.c:125-129 + 74-91.  The host arrays are the caller's, sized as referencePack sizes them: index / offset / id / rev
   [ref->max], depth / loc [ms->max + 1], info = ms->info; tallies[3] = copy 1, copy 2, multiple. */
extern "C" MgStatus mgRefBuildFinish (MgReference *ref, U32 *hIndex, U32 *hOffset, U32 *hId, U32 *hDepth, U32 *hRev, U32 *hLoc, U8 *hInfo, U32 tallies[3])
{
  std::lock_guard<std::mutex> own (mgRefOwnLock (ref));
  MgRefDev &d = mgRefEntry (ref);
  hipStream_t st = 0;
  const U32 msMax = ref->ms->max, n = ref->max;
  const size_t m = (size_t) msMax + 1;
  tallies[0] = tallies[1] = tallies[2] = 0;
  struct timespec tq0; clock_gettime (CLOCK_MONOTONIC, &tq0);
  const bool lapOn = mgKnobs ()->seedTiming == 1;
#define MG_LAP(what) do { if (lapOn) { (void) hipStreamSynchronize (st); struct timespec q_; clock_gettime (CLOCK_MONOTONIC, &q_); fprintf (stderr, "mgRefBuildFinish: %s at %.1f ms\n", what, (q_.tv_sec - tq0.tv_sec) * 1e3 + (q_.tv_nsec - tq0.tv_nsec) * 1e-6); } } while (0)
  if (d.packed) { mgSetError ("the reference is packed already"); return MG_ERR_ARG; }
  { int cur = 0; MG_HIP (hipGetDevice (&cur));
    if (d.device >= 0 && d.device != cur && d.depth) { mgSetError ("this Reference is being built on GPU %d, the calling thread is on GPU %d", d.device, cur); return MG_ERR_ARG; }
    d.device = cur;
  }
  if (!d.depth)                                                      /* no batch ever came: an empty reference */
    { MG_HIP (hipMalloc ((void **) &d.depth, (m + 1) * 4)); MG_HIP (hipMemsetAsync (d.depth, 0, (m + 1) * 4, st)); d.capMs = m; }
  if (m > d.capMs + 1) { mgSetError ("modset grew beyond its size"); return MG_ERR_CAPACITY; }
  MgStatus s;
  U32 *tiles = 0, *dTal = 0;
  do {
    s = MG_ERR_HIP;
    const U32 nSortTiles = (U32) (((U64) n + MG_RSORT_TILE - 1) / MG_RSORT_TILE);
    const size_t histWords = (size_t) 256 * (nSortTiles ? nSortTiles : 1);
    const size_t scanTiles = (m > histWords ? m : histWords) / MG_SCAN_TILE + 4;
    if (hipMalloc ((void **) &tiles, scanTiles * 4) || hipMalloc ((void **) &dTal, 16) || hipMemsetAsync (dTal, 0, 16, st)) break;
    /* info: the host's flag bytes up, the copy classes set, back down (and kept for the chaining) */
    (void) hipFree (d.info); d.info = 0; (void) hipFree (d.loc); d.loc = 0;
    if (hipMalloc ((void **) &d.info, m) || hipMalloc ((void **) &d.loc, (m + 1) * 4) || hipStreamSynchronize (st)) break;
    MG_LAP ("allocations");
    if ((s = mgXferH2DSparse (d.info, hInfo, m))) break;
    MG_LAP ("info up");      /* (a new Modset's info[] has never been written: nothing is read, and the mirror below lands on fresh pages) */
    s = MG_ERR_HIP;
    if (msMax) hipLaunchKernelGGL (mgRefClassifyKernel, dim3 (2048), dim3 (256), 0, st, d.depth, msMax, d.info, dTal);
    if ((s = mgRefExclusiveScan (d.depth, d.loc, m, tiles, st))) break;
    MG_LAP ("classes + loc");
    s = MG_ERR_HIP;
    if (hipMemcpyAsync (tallies, dTal, 12, hipMemcpyDeviceToHost, st)) break;
    /* rev: the occurrence ordinals sorted by index, stably */
    (void) hipFree (d.rev); d.rev = 0;
    if (n)
      { int keyBits = 1; while (keyBits < 32 && ((U64) 1 << keyBits) <= msMax) ++keyBits;
        MG_LAP ("sort allocations");
        if ((s = mgRefStableSort (d.index, 0, n, keyBits, &d.rev, tiles, st))) break;
        s = MG_ERR_HIP;
        MG_LAP ("sort");
        if (hipMemsetAsync (d.rev + n, 0, 4, st)) break;            /* (the chaining reads rev[loc[x] + 1] of a copy-2 seed: inside the array, but keep the slack defined) */
      }
    else { if (hipMalloc ((void **) &d.rev, 8) || hipMemsetAsync (d.rev, 0, 8, st)) break; }
    if (hipStreamSynchronize (st)) break;
    if (mgKnobs ()->seedTiming == 1) { struct timespec q; clock_gettime (CLOCK_MONOTONIC, &q); fprintf (stderr, "mgRefBuildFinish: kernels done at %.1f ms\n", (q.tv_sec - tq0.tv_sec) * 1e3 + (q.tv_nsec - tq0.tv_nsec) * 1e-6); }
    /* mirror */
    if ((s = mgXferD2H (hInfo, d.info, m, MG_XFER_COPY)) || (s = mgXferD2H (hDepth, d.depth, m * 4, MG_XFER_COPY)) || (s = mgXferD2H (hLoc, d.loc, m * 4, MG_XFER_COPY))) break;
    if (n && ((s = mgXferD2H (hIndex, d.index, (size_t) n * 4, MG_XFER_COPY)) || (s = mgXferD2H (hOffset, d.offset, (size_t) n * 4, MG_XFER_COPY))
              || (s = mgXferD2H (hId, d.id, (size_t) n * 4, MG_XFER_COPY)) || (s = mgXferD2H (hRev, d.rev, (size_t) n * 4, MG_XFER_COPY)))) break;
    /* what the chaining does not read goes; the rest is the resident copy */
    (void) hipFree (d.index); d.index = 0; (void) hipFree (d.depth); d.depth = 0; d.capOcc = 0; d.capMs = 0;
    if (!d.id) { if (hipMalloc ((void **) &d.id, 8) || hipMalloc ((void **) &d.offset, 8) || hipMemset (d.id, 0, 8)) { s = MG_ERR_HIP; break; } }
    d.nSeq = ref->nSeq;
    MG_LAP ("mirror");
    if ((s = mgRefDerive (d, msMax, n, st))) break;
    MG_LAP ("derive");
    d.msMax = msMax; d.refMax = n; d.packed = true;
    d.tallies[0] = tallies[0]; d.tallies[1] = tallies[1]; d.tallies[2] = tallies[2];
    s = MG_OK;
  } while (0);
  (void) hipFree (tiles); (void) hipFree (dTal);
  if (s == MG_ERR_HIP && !mgLastError ()[0]) mgHipFail (hipGetLastError (), "reference pack on the device");
  return s;
}


/* ---------------------------------------------------------------------------------------- */
/* modasm's read ingest: what readsetFileRead + invBuild (modasm.c:151-191,258-287) leave per MOD, on the device.
 *
 * Per batch the hit lists are made by mg_chain.hip (mgReadsetSeedsDevice); the hits per mod are counted THERE into an array that lives
 * here across the batches of a file (rounds 1-4: copied back and folded into ms->depth by a host loop over every mod, per batch).
 * At the end of the file: depth[] = the counts saturated at 65 535 (modasm.c:174); the inverse lists -- for every mod that was hit and
 * did not saturate, the reads that hit it, in read order (modasm.c:266,278) -- are a stable sort of the hits' read numbers by mod
 * (mgRefStableSort: the sort of referencePack), hits on saturated mods keyed past the last mod so that they fall off the end;
 * invStart[] is the exclusive scan of the lists' lengths; a read's copy-class tallies (modasm.c:276-277) are a lane per read. */
#define MG_RS_TOPMASK 0x7fffffffu
struct MgReadsetDev { U32 *depth = 0; size_t cap = 0; U32 *hitAll = 0; U64 hitLen = 0, hitCap = 0; bool hitsKept = true; };      /* hitAll: the file's hit lists so far, kept on the device for the inverse lists (given up, and uploaded at the end instead, if the device has no room) */
static std::mutex gRsLock;
static std::unordered_map<const void *, MgReadsetDev> gRsDev;

extern "C" void mgReadsetDevForget (const void *rs)
{ std::lock_guard<std::mutex> g (gRsLock); auto it = gRsDev.find (rs); if (it != gRsDev.end ()) { (void) hipFree (it->second.depth); (void) hipFree (it->second.hitAll); gRsDev.erase (it); } }

/* the per-mod hit counts of the file that follows: device U32[msMax + 2], zero (modasm.c:158) */
extern "C" MgStatus mgReadsetDevBegin (const void *rs, U32 msMax, U32 **dDepth)
{
  std::lock_guard<std::mutex> g (gRsLock);
  MgReadsetDev &d = gRsDev[rs];
  const size_t want = (size_t) msMax + 2;
  if (d.cap < want) { (void) hipFree (d.depth); d.depth = 0; d.cap = 0; MG_HIP (hipMalloc ((void **) &d.depth, want * 4)); d.cap = want; }
  MG_HIP (hipMemset (d.depth, 0, want * 4));
  d.hitLen = 0; d.hitsKept = true;
  *dDepth = d.depth;
  return MG_OK;
}

/* a batch's hit list (device, n words) behind the file's so far */
extern "C" void mgReadsetDevAppendHits (const void *rs, const U32 *dHit, U64 n)
{
  std::lock_guard<std::mutex> g (gRsLock);
  auto it = gRsDev.find (rs); if (it == gRsDev.end ()) return;
  MgReadsetDev &d = it->second;
  if (!d.hitsKept || !n) return;
  if (d.hitLen + n > d.hitCap)
    { const U64 cap = (d.hitLen + n) + (d.hitLen ? (d.hitLen + n) / 2 : 0) + 1024;      /* (a file that is one batch gets what it needs; one of many batches grows by halves) */
      U32 *q = 0;
      if (hipMalloc ((void **) &q, cap * 4) != hipSuccess || (d.hitLen && hipMemcpy (q, d.hitAll, d.hitLen * 4, hipMemcpyDeviceToDevice) != hipSuccess))
        { (void) hipGetLastError (); (void) hipFree (q); (void) hipFree (d.hitAll); d.hitAll = 0; d.hitCap = d.hitLen = 0; d.hitsKept = false; return; }
      (void) hipFree (d.hitAll); d.hitAll = q; d.hitCap = cap;
    }
  if (hipMemcpy (d.hitAll + d.hitLen, dHit, n * 4, hipMemcpyDeviceToDevice) != hipSuccess) { (void) hipGetLastError (); d.hitsKept = false; return; }
  d.hitLen += n;
}

__global__ void mgRsCountKernel (const U32 *__restrict__ depth32, U32 msMax, U32 *__restrict__ cnt, unsigned short *__restrict__ depth16)
{
  for (U64 i = (U64) blockIdx.x * blockDim.x + threadIdx.x ; i <= (U64) msMax + 1 ; i += (U64) gridDim.x * blockDim.x)
    { const U32 dp = (i >= 1 && i <= msMax) ? depth32[i] : 0u;
      cnt[i] = (dp && dp < 0xffffu) ? dp : 0u;                       /* a list only for a mod that was hit and did not saturate (modasm.c:266) */
      if (i <= msMax) depth16[i] = (unsigned short) (dp > 0xffffu ? 0xffffu : dp);
    }
}
__global__ void mgRsWidenKernel (const U32 *__restrict__ a, U64 n, U64 *__restrict__ out)
{ for (U64 i = (U64) blockIdx.x * blockDim.x + threadIdx.x ; i < n ; i += (U64) gridDim.x * blockDim.x) out[i] = a[i]; }
/* per hit: key = its mod (past the last mod if that one saturated), value = its read: the one whose hitStart range holds it (reads from 1) */
__global__ void mgRsKeyValKernel (const U32 *__restrict__ hit, U64 nHit, const U64 *__restrict__ hitStart, U32 nReads, const U32 *__restrict__ depth32, U32 msMax,
                                  U32 *__restrict__ key, U32 *__restrict__ val)
{
  for (U64 h = (U64) blockIdx.x * blockDim.x + threadIdx.x ; h < nHit ; h += (U64) gridDim.x * blockDim.x)
    { const U32 y = hit[h] & MG_RS_TOPMASK;
      key[h] = depth32[y] < 0xffffu ? y : msMax + 1;
      U32 lo = 1, hi = nReads;                                       /* the last r with hitStart[r] <= h */
      while (lo < hi) { const U32 mid = lo + (hi - lo + 1) / 2; if (hitStart[mid] <= h) lo = mid; else hi = mid - 1; }
      val[h] = lo;
    }
}
__global__ __launch_bounds__ (256)
void mgRsCopyTallyKernel (const U32 *__restrict__ hit, const U64 *__restrict__ hitStart, U32 nReads, const U8 *__restrict__ info, int4 *__restrict__ nCopy)
{
  const U32 r = 1 + blockIdx.x * blockDim.x + threadIdx.x;
  if (r > nReads) return;
  int c[4] = { 0, 0, 0, 0 };
  const U64 h0 = hitStart[r], h1 = hitStart[r + 1];
  for (U64 h = h0 ; h < h1 ; h += 8)
    { U32 cl[8];
#pragma unroll
      for (int j = 0 ; j < 8 ; ++j) cl[j] = h + j < h1 ? (U32) info[hit[h + j] & MG_RS_TOPMASK] & 3u : 4u;      /* eight gathers in flight */
#pragma unroll
      for (int j = 0 ; j < 8 ; ++j) { c[0] += cl[j] == 0; c[1] += cl[j] == 1; c[2] += cl[j] == 2; c[3] += cl[j] == 3; }
    }
  nCopy[r] = make_int4 (c[0], c[1], c[2], c[3]);
}

/* hHit[totHit], hHitStart[nReads + 2] (reads from 1; [nReads + 1] = totHit), hInfo[msMax + 1]: in.  hDepth16[msMax + 1], hInvStart[msMax + 2],
   *hInvSpace (malloc ()ed here, the lists' total length words), hNCopy[(nReads + 1) * 4]: out.  totHit < 2^32 - 1. */
extern "C" MgStatus mgReadsetFinishDevice (const void *rs, Modset *ms, U32 msMax, const U32 *hHit, U64 totHit, const U64 *hHitStart, U32 nReads, const U8 *hInfo,
                                           U16 *hDepth16, U64 *hInvStart, U32 **hInvSpace, int *hNCopy)
{
  *hInvSpace = 0;
  MgReadsetDev d;
  bool kept = false;
  { std::lock_guard<std::mutex> g (gRsLock); auto it = gRsDev.find (rs); if (it == gRsDev.end ()) { mgSetError ("mgReadsetFinishDevice: no read set in progress"); return MG_ERR_ARG; }
    d = it->second;
    kept = d.hitsKept && d.hitAll && d.hitLen == totHit && totHit;      /* the lists are on the device already: no upload */
    if (kept) { it->second.hitAll = 0; it->second.hitCap = it->second.hitLen = 0; }      /* (this call owns them now, and frees them) */
    else { (void) hipFree (it->second.hitAll); it->second.hitAll = 0; it->second.hitCap = it->second.hitLen = 0; }
  }
  hipStream_t st = 0;
  const size_t m = (size_t) msMax + 1;
  U32 *dHit = kept ? d.hitAll : 0, *dCnt = 0, *dKey = 0, *dVal = 0, *dSorted = 0, *tiles = 0; U64 *dStart = 0, *dInv64 = 0; U8 *dInfo = 0; unsigned short *dD16 = 0; int4 *dNc = 0;
  MgStatus s = MG_ERR_HIP;
  do {
    const size_t histWords = (size_t) 256 * ((totHit + MG_RSORT_TILE - 1) / MG_RSORT_TILE + 1);
    const size_t scanTiles = ((m + 2) > histWords ? (m + 2) : histWords) / MG_SCAN_TILE + 4;
    if (hipMalloc ((void **) &tiles, scanTiles * 4) || hipMalloc ((void **) &dCnt, (m + 2) * 4) || hipMalloc ((void **) &dD16, (m + 1) * 2) || hipMalloc ((void **) &dInv64, (m + 2) * 8)
        || hipMalloc ((void **) &dInfo, m) || hipMalloc ((void **) &dStart, ((size_t) nReads + 3) * 8) || hipMalloc ((void **) &dNc, ((size_t) nReads + 2) * sizeof (int4))
        || (!kept && hipMalloc ((void **) &dHit, (totHit + 1) * 4)) || hipMalloc ((void **) &dKey, (totHit + 1) * 4) || hipMalloc ((void **) &dVal, (totHit + 1) * 4) || hipDeviceSynchronize ()) break;
    if ((s = mgXferH2DSparse (dInfo, hInfo, m)) || (s = mgXferH2D (dStart, hHitStart, ((size_t) nReads + 2) * 8)) || (totHit && !kept && (s = mgXferH2D (dHit, hHit, totHit * 4)))) break;
    s = MG_ERR_HIP;
    hipLaunchKernelGGL (mgRsCountKernel, dim3 (2048), dim3 (256), 0, st, d.depth, msMax, dCnt, dD16);
    if ((s = mgRefExclusiveScan (dCnt, dCnt, m + 1, tiles, st))) break;      /* dCnt[i] = first place of mod i's list; [msMax + 1] = the lists' total */
    s = MG_ERR_HIP;
    hipLaunchKernelGGL (mgRsWidenKernel, dim3 (2048), dim3 (256), 0, st, dCnt, (U64) m + 1, dInv64);
    if (nReads) hipLaunchKernelGGL (mgRsCopyTallyKernel, dim3 ((nReads + 255) / 256), dim3 (256), 0, st, dHit, dStart, nReads, dInfo, dNc);
    U32 listed = 0;
    if (hipMemcpyAsync (&listed, dCnt + m, 4, hipMemcpyDeviceToHost, st) || hipStreamSynchronize (st)) break;
    if (totHit)
      { hipLaunchKernelGGL (mgRsKeyValKernel, dim3 (4096), dim3 (256), 0, st, dHit, totHit, dStart, nReads, d.depth, msMax, dKey, dVal);
        int keyBits = 1; while (keyBits < 32 && ((U64) 1 << keyBits) <= (U64) msMax + 1) ++keyBits;
        if ((s = mgRefStableSort (dKey, dVal, (U32) totHit, keyBits, &dSorted, tiles, st))) break;
        s = MG_ERR_HIP;
      }
    if (hipGetLastError () != hipSuccess || hipStreamSynchronize (st)) break;
    U32 *inv = (U32 *) mgAllocBig (((size_t) listed ? listed : 1) * 4);
    if (!inv) { s = MG_ERR_NOMEM; break; }
    *hInvSpace = inv;
    if ((s = mgXferD2H (hDepth16, dD16, m * 2, MG_XFER_COPY)) || (s = mgXferD2H (hInvStart, dInv64, (m + 1) * 8, MG_XFER_COPY))
        || (listed && (s = mgXferD2H (inv, dSorted, (size_t) listed * 4, MG_XFER_COPY)))
        || (nReads && (s = mgXferD2H (hNCopy + 4, dNc + 1, (size_t) nReads * sizeof (int4), MG_XFER_COPY)))) break;
    if ((s = mgModsetAdoptDepthDevice (ms, (const U16 *) dD16))) break;      /* the device table keeps up with the depth[] just mirrored (no rebuild on its next use) */
    s = MG_OK;
  } while (0);
  (void) hipFree (dHit); (void) hipFree (dCnt); (void) hipFree (dKey); (void) hipFree (dVal); (void) hipFree (dSorted); (void) hipFree (tiles);
  (void) hipFree (dStart); (void) hipFree (dInv64); (void) hipFree (dInfo); (void) hipFree (dD16); (void) hipFree (dNc);
  if (s == MG_ERR_HIP && !mgLastError ()[0]) mgHipFail (hipGetLastError (), "read set on the device");
  if (s && *hInvSpace) { free (*hInvSpace); *hInvSpace = 0; }
  return s;
}
