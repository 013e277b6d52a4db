/* mg_table.hip — K3/K4/K5: the device-resident modset table for gfx950.
 *
 * What it must reproduce (reference modset.c:45-62 and its callers): looking a k-mer up returns
 * its index or 0; inserting a new k-mer gives it index ++max, i.e. indices are handed out in
 * order of FIRST OCCURRENCE in the (read,pos)-ordered modimizer stream, and every occurrence bumps
 * a saturating 16-bit depth (modutils.c:26).
 *
 * The device table is not the reference's index[] array: with a power-of-two modulus d every
 * primary slot of the reference table has its low log2(d) bits zero (the hash that picks the slot
 * is the hash that was just tested to be 0 mod d), so that layout is only materialised on request
 * (mgReplayIndexKernel, for .mod files and host-side scalar lookups).  Here slots are 16 bytes
 * {kmer+1, ordIdx, cnt} in an open-addressed, linearly probed array addressed by a remix of the
 * k-mer, zero-initialised (key 0 = empty).
 *
 * Deterministic first-occurrence indices without a sort: every occurrence o of a not-yet-indexed
 * k-mer posts a token that is larger the smaller o is (atomicMax).  A second, streaming pass asks
 * each occurrence "is the slot's token mine?" - exactly the first occurrences say yes - and an
 * ordered prefix sum over those flags (decoupled look-back, as in the scan) turns them into
 * max+1, max+2, ... in stream order.  Assigned indices have bit 31 clear, tokens have it set, so
 * the two value spaces cannot be confused while the pass is rewriting slots.
 */
#include "mg_common.h"

#define MG_TOKEN_BIT 0x80000000u
__device__ __forceinline__ U32 mgToken (U64 o) { return MG_TOKEN_BIT | (0x7fffffffu - (U32) o); }
__device__ __forceinline__ bool mgAssigned (U32 v) { return v != 0 && !(v & MG_TOKEN_BIT); }

__device__ __forceinline__ U64 mgMix (U64 x)
{
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}

/* counters[]: 0 = new entries this call, 1 = probe overflow flag */

__global__ void mgTableInsertKernel (MgSlot *__restrict__ slots, U64 mask, const U64 *__restrict__ kmer, U64 n,
                                     U32 *__restrict__ slotId, int withDepth, U64 *counters)
{
  U64 o = (U64) blockIdx.x * blockDim.x + threadIdx.x;
  const U64 stride = (U64) gridDim.x * blockDim.x;
  for ( ; o < n ; o += stride)
    { const U64 km = kmer[o], key = km + 1;
      U64 s = mgMix (km) & mask;
      U64 probes = 0;
      bool ok = true;
      for (;;)
        { U64 cur = slots[s].key;                 /* plain load: a stale "empty" is repaired by the CAS */
          if (cur == 0)
            { cur = atomicCAS ((unsigned long long *) &slots[s].key, 0ull, (unsigned long long) key);
              if (cur == 0) cur = key;
            }
          if (cur == key) break;
          s = (s + 1) & mask;
          if (++probes > mask) { ok = false; break; }
        }
      if (!ok) { counters[1] = 1; slotId[o] = 0xffffffffu; continue; }
      U32 v = slots[s].ordIdx;                    /* tokens only grow, so a stale read is only ever too small */
      U32 tok = mgToken (o);
      if (!mgAssigned (v) && v < tok) atomicMax (&slots[s].ordIdx, tok);
      if (withDepth) atomicAdd (&slots[s].cnt, 1u);
      slotId[o] = (U32) s;
    }
}

#define MG_ASSIGN_PER_THREAD 8
#define MG_ASSIGN_TILE (256 * MG_ASSIGN_PER_THREAD)

__global__ __launch_bounds__ (256)
void mgTableAssignKernel (MgSlot *__restrict__ slots, const U64 *__restrict__ kmer, const U32 *__restrict__ slotId,
                          U64 n, U64 nTiles, U64 *desc, U32 *ticket,
                          U64 *__restrict__ value, U32 *__restrict__ slotOfIndex, U32 baseMax, U32 size,
                          U64 *counters)
{
  __shared__ U32 sWaveTot[4];
  __shared__ U64 sBase;
  __shared__ U32 sTile;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (;;)
    { if (tid == 0) sTile = atomicAdd (ticket, 1u);
      __syncthreads ();
      const U64 tile = sTile;
      if (tile >= nTiles) break;
      const U64 o0 = tile * MG_ASSIGN_TILE + (U64) tid * MG_ASSIGN_PER_THREAD;
      U32 sid[MG_ASSIGN_PER_THREAD];
      U32 flags = 0;
#pragma unroll
      for (int j = 0 ; j < MG_ASSIGN_PER_THREAD ; ++j)
        { U64 o = o0 + j;
          sid[j] = 0xffffffffu;
          if (o < n)
            { sid[j] = slotId[o];
              if (sid[j] != 0xffffffffu && slots[sid[j]].ordIdx == mgToken (o)) flags |= 1u << j;
            }
        }
      U32 cnt = __popc (flags), incl = cnt;
#pragma unroll
      for (int off = 1 ; off < 64 ; off <<= 1)
        { U32 v = __shfl_up (incl, off); if (lane >= off) incl += v; }
      if (lane == 63) sWaveTot[wave] = incl;
      __syncthreads ();
      U32 waveBase = 0, total = 0;
#pragma unroll
      for (int i = 0 ; i < 4 ; ++i) { U32 v = sWaveTot[i]; if (i < wave) waveBase += v; total += v; }
      if (wave == 0)
        { U64 b = mgLookback (desc, tile, total);
          if (lane == 0) { sBase = b; if (tile == nTiles - 1) counters[0] = b + total; }
        }
      __syncthreads ();
      U64 rank = sBase + waveBase + (incl - cnt);
#pragma unroll
      for (int j = 0 ; j < MG_ASSIGN_PER_THREAD ; ++j)
        if (flags & (1u << j))
          { U64 idx = (U64) baseMax + 1 + rank++;
            if (idx < size)
              { slots[sid[j]].ordIdx = (U32) idx;
                value[idx] = kmer[o0 + j];
                slotOfIndex[idx] = sid[j];
              }
          }
      __syncthreads ();
    }
}

__global__ void mgTableGatherKernel (const MgSlot *__restrict__ slots, const U32 *__restrict__ slotId, U64 n,
                                     U32 *__restrict__ out)
{
  U64 o = (U64) blockIdx.x * blockDim.x + threadIdx.x;
  const U64 stride = (U64) gridDim.x * blockDim.x;
  for ( ; o < n ; o += stride)
    { U32 s = slotId[o];
      U32 v = s == 0xffffffffu ? 0 : slots[s].ordIdx;
      out[o] = mgAssigned (v) ? v : 0;
    }
}

__global__ void mgTableFindKernel (const MgSlot *__restrict__ slots, U64 mask, const U64 *__restrict__ kmer, U64 n,
                                   U32 *__restrict__ out)
{
  U64 o = (U64) blockIdx.x * blockDim.x + threadIdx.x;
  const U64 stride = (U64) gridDim.x * blockDim.x;
  for ( ; o < n ; o += stride)
    { const U64 km = kmer[o], key = km + 1;
      U64 s = mgMix (km) & mask;
      U32 res = 0;
      for (U64 probes = 0 ; probes <= mask ; ++probes)
        { U64 cur = slots[s].key;
          if (cur == key) { U32 v = slots[s].ordIdx; res = mgAssigned (v) ? v : 0; break; }
          if (cur == 0) break;
          s = (s + 1) & mask;
        }
      out[o] = res;
    }
}

/* entries first..last (with their existing indices) from a host modset into the device table */
__global__ void mgTableLoadKernel (MgSlot *__restrict__ slots, U64 mask, const U64 *__restrict__ value,
                                   U32 first, U32 last, U32 *__restrict__ slotOfIndex, U64 *counters)
{
  U64 i = (U64) first + (U64) blockIdx.x * blockDim.x + threadIdx.x;
  const U64 stride = (U64) gridDim.x * blockDim.x;
  for ( ; i <= last ; i += stride)
    { const U64 km = value[i], key = km + 1;
      U64 s = mgMix (km) & mask;
      U64 probes = 0;
      for (;;)
        { U64 cur = slots[s].key;
          if (cur == 0)
            { cur = atomicCAS ((unsigned long long *) &slots[s].key, 0ull, (unsigned long long) key);
              if (cur == 0) break;
            }
          if (cur == key) break;        /* duplicate value in the host arrays: keep the first */
          s = (s + 1) & mask;
          if (++probes > mask) { counters[1] = 1; break; }
        }
      slots[s].ordIdx = (U32) i;
      slotOfIndex[i] = (U32) s;
    }
}

/* pending depth counts of entries first..last -> delta16[], folded into baseDepth, cnt zeroed */
__global__ void mgTableExportDepthKernel (MgSlot *__restrict__ slots, const U32 *__restrict__ slotOfIndex,
                                          U16 *__restrict__ baseDepth, U16 *__restrict__ delta, U32 first, U32 last)
{
  U64 i = (U64) first + (U64) blockIdx.x * blockDim.x + threadIdx.x;
  const U64 stride = (U64) gridDim.x * blockDim.x;
  for ( ; i <= last ; i += stride)
    { U32 s = slotOfIndex[i];
      U32 c = slots[s].cnt;
      slots[s].cnt = 0;
      U32 cl = c > 0xffffu ? 0xffffu : c;
      delta[i - first] = (U16) cl;
      U32 b = (U32) baseDepth[i] + cl;
      baseDepth[i] = (U16) (b > 0xffffu ? 0xffffu : b);
    }
}

/* K5: histogram of min(65535, baseDepth + pending) over entries 1..max (modutils.c:53-63) */
#define MG_HIST_LDS_BINS 8192
__global__ __launch_bounds__ (256)
void mgTableHistKernel (const MgSlot *__restrict__ slots, const U32 *__restrict__ slotOfIndex,
                        const U16 *__restrict__ baseDepth, U32 max, unsigned long long *__restrict__ hist)
{
  __shared__ U32 sBins[MG_HIST_LDS_BINS];
  for (int b = threadIdx.x ; b < MG_HIST_LDS_BINS ; b += blockDim.x) sBins[b] = 0;
  __syncthreads ();
  U64 i = 1 + (U64) blockIdx.x * blockDim.x + threadIdx.x;
  const U64 stride = (U64) gridDim.x * blockDim.x;
  for ( ; i <= max ; i += stride)
    { U32 d = (U32) baseDepth[i] + slots[slotOfIndex[i]].cnt;
      if (d > 0xffffu || d < baseDepth[i]) d = 0xffffu;
      if (d < MG_HIST_LDS_BINS) atomicAdd (&sBins[d], 1u);
      else atomicAdd (&hist[d], 1ull);
    }
  __syncthreads ();
  for (int b = threadIdx.x ; b < MG_HIST_LDS_BINS ; b += blockDim.x)
    if (sBins[b]) atomicAdd (&hist[b], (unsigned long long) sBins[b]);
}

/* The reference's index[] layout (modset.c:45-62) rebuilt in parallel.  Sequential insertion in
 * index order puts entry i in the first slot of its probe sequence not held by an entry < i.
 * That layout is the unique one in which every slot an entry skipped holds a smaller index, so
 * it is reached by letting entries race with atomicMin: a smaller index evicts a larger one,
 * which then resumes its own probe sequence from where it sat.  Empty is 0xffffffff during the
 * race; mgIndexFinishKernel turns it into the reference's 0. */
__global__ void mgReplayIndexKernel (const U64 *__restrict__ value, U32 max, U64 factor1, int shift1,
                                     int tableBits, U32 *__restrict__ index)
{
  const U64 tmask = ((U64) 1 << tableBits) - 1;
  U64 i = 1 + (U64) blockIdx.x * blockDim.x + threadIdx.x;
  const U64 stride = (U64) gridDim.x * blockDim.x;
  for ( ; i <= max ; i += stride)
    { U32 cur = (U32) i;
      U64 hash = (value[cur] * factor1) >> shift1;
      U64 off = hash & tmask;
      U64 diff = ((hash >> tableBits) & tmask) | 1;
      for (;;)
        { U32 prev = atomicMin (&index[off], cur);
          if (prev == 0xffffffffu) break;
          if (prev > cur)
            { cur = prev;                               /* evicted entry continues from this slot */
              hash = (value[cur] * factor1) >> shift1;
              diff = ((hash >> tableBits) & tmask) | 1;
            }
          off = (off + diff) & tmask;
        }
    }
}

__global__ void mgIndexFinishKernel (U32 *__restrict__ index, U64 n)
{
  U64 i = (U64) blockIdx.x * blockDim.x + threadIdx.x;
  const U64 stride = (U64) gridDim.x * blockDim.x;
  for ( ; i < n ; i += stride) if (index[i] == 0xffffffffu) index[i] = 0;
}

/* ---------------------------------------------------------------------------------------- */

static inline unsigned mgGrid (U64 n, unsigned per = 256, unsigned cap = 16384)
{ U64 b = (n + per - 1) / per; if (b > cap) b = cap; if (b < 1) b = 1; return (unsigned) b; }

size_t mgAssignDescBytes (U64 n)
{ U64 nTiles = (n + MG_ASSIGN_TILE - 1) / MG_ASSIGN_TILE; return (size_t) (256 + nTiles * 8 + 255) & ~(size_t) 255; }

MgStatus mgTableInsert (MgTable *t, const U64 *dKmer, U64 n, U32 *dSlotId, int withDepth, hipStream_t st)
{
  if (!n) return MG_OK;
  MG_LAUNCH (MG_K_TABLE_INSERT, st, mgTableInsertKernel, dim3 (mgGrid (n)), dim3 (256), 0, st,
                      t->slots, t->slotMask, dKmer, n, dSlotId, withDepth, t->counters);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}

MgStatus mgTableAssign (MgTable *t, const U64 *dKmer, U64 n, const U32 *dSlotId, void *dDesc, hipStream_t st)
{
  if (!n) return MG_OK;
  U64 nTiles = (n + MG_ASSIGN_TILE - 1) / MG_ASSIGN_TILE;
  MG_HIP (hipMemsetAsync (dDesc, 0, 256 + nTiles * 8, st));
  U32 *ticket = (U32 *) dDesc;
  U64 *desc = (U64 *) ((char *) dDesc + 256);
  unsigned grid = (unsigned) (nTiles < 2048 ? nTiles : 2048);
  MG_LAUNCH (MG_K_TABLE_ASSIGN, st, mgTableAssignKernel, dim3 (grid), dim3 (256), 0, st,
                      t->slots, dKmer, dSlotId, n, nTiles, desc, ticket,
                      t->value, t->slotOfIndex, t->max, t->size, t->counters);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}

MgStatus mgTableGather (MgTable *t, const U32 *dSlotId, U64 n, U32 *dIndexOut, hipStream_t st)
{
  if (!n) return MG_OK;
  MG_LAUNCH (MG_K_TABLE_GATHER, st, mgTableGatherKernel, dim3 (mgGrid (n)), dim3 (256), 0, st, t->slots, dSlotId, n, dIndexOut);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}

MgStatus mgTableFind (MgTable *t, const U64 *dKmer, U64 n, U32 *dIndexOut, hipStream_t st)
{
  if (!n) return MG_OK;
  MG_LAUNCH (MG_K_TABLE_FIND, st, mgTableFindKernel, dim3 (mgGrid (n)), dim3 (256), 0, st, t->slots, t->slotMask, dKmer, n, dIndexOut);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}

MgStatus mgTableLoadHost (MgTable *t, const U64 *dValue, U32 first, U32 last, hipStream_t st)
{
  if (last < first) return MG_OK;
  MG_LAUNCH (MG_K_TABLE_LOAD, st, mgTableLoadKernel, dim3 (mgGrid ((U64) last - first + 1)), dim3 (256), 0, st,
                      t->slots, t->slotMask, dValue, first, last, t->slotOfIndex, t->counters);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}

MgStatus mgTableExportDepth (MgTable *t, U16 *dDelta, U32 first, U32 last, hipStream_t st)
{
  if (last < first) return MG_OK;
  MG_LAUNCH (MG_K_TABLE_EXPORT, st, mgTableExportDepthKernel, dim3 (mgGrid ((U64) last - first + 1)), dim3 (256), 0, st,
                      t->slots, t->slotOfIndex, t->baseDepth, dDelta, first, last);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}

MgStatus mgTableHistogram (MgTable *t, U64 *dHist, hipStream_t st)
{
  if (!t->max) return MG_OK;
  MG_LAUNCH (MG_K_TABLE_HIST, st, mgTableHistKernel, dim3 (mgGrid (t->max, 256, 1024)), dim3 (256), 0, st,
                      t->slots, t->slotOfIndex, t->baseDepth, t->max, (unsigned long long *) dHist);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}

MgStatus mgTableReplayIndex (MgTable *t, const MgHashParams &p, int tableBits, U32 *dIndex, hipStream_t st)
{
  U64 n = (U64) 1 << tableBits;
  MG_HIP (hipMemsetAsync (dIndex, 0xff, n * sizeof (U32), st));
  if (t->max)
    { MG_LAUNCH (MG_K_INDEX_REPLAY, st, mgReplayIndexKernel, dim3 (mgGrid (t->max)), dim3 (256), 0, st,
                          t->value, t->max, p.factor1, p.shift1, tableBits, dIndex);
      MG_HIP (hipGetLastError ());
    }
  MG_LAUNCH (MG_K_INDEX_FINISH, st, mgIndexFinishKernel, dim3 (mgGrid (n, 256, 8192)), dim3 (256), 0, st, dIndex, n);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}
