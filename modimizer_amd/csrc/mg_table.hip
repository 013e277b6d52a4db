/* mg_table.hip — K3/K4/K5: the device-resident modset table for gfx950.
 *
 * What it must reproduce (reference modset.c:45-62 and its callers): looking a k-mer up returns
 * its index or 0; inserting a new k-mer gives it index ++max, i.e. indices are handed out in
 * order of FIRST OCCURRENCE in the (read,pos)-ordered modimizer stream, and every occurrence bumps
 * a saturating 16-bit depth (modutils.c:26).
 *
 * Layout.  The device table is not the reference's index[] array (with a power-of-two modulus d
 * every primary slot of that table has its low log2(d) bits zero, because the hash that picks the
 * slot is the hash that was just tested to be 0 mod d; that layout is only materialised on
 * request by mgReplayIndexKernel).  Here: NB buckets of R slots of 16 bytes {mix(kmer)+1, ord, cnt},
 * where mix is a bijection of the 2k-bit k-mers (mgMixK); a k-mer's bucket is the top bits of its mixed
 * value, its home slot the low bits, and linear probing wraps INSIDE the bucket, so a bucket is a
 * self-contained little table that fits LDS.
 * ord: 0 = none, bit 31 set = assigned index, otherwise a transient first-occurrence token.
 *
 * Two build paths, same table, same results:
 *
 *  direct   (small batches)  global atomics: CAS the key in, post a token that is larger the
 *           earlier the occurrence (atomicMax: assigned indices have bit 31 set, so they are never
 *           lowered), atomicAdd the count; then a streaming pass asks every occurrence "is the
 *           slot's token mine?" - exactly the first occurrences say yes - and an ordered count
 *           over those flags turns them into max+1, max+2, ...
 *
 *  bucketed (large batches)  no global atomics on the data path.  Measured on MI355X: a random
 *           global atomic costs ~1/17 G/s (three on one slot serialise), a random load ~1/45 G/s,
 *           streaming ~5 TB/s.  So: (1) radix-partition the batch by bucket (two streaming passes; the
 *           first reads the scan's per-worker segments as they are, MgSegSrc, and its digit counts come
 *           from the scan kernel), (2) one workgroup per bucket dedups its occurrences in LDS (LDS
 *           atomics), flags the first occurrence of every k-mer that is new to the table (or clears the
 *           flags of the others: markDup) and writes the bucket's unique k-mers back grouped by the slice
 *           of the ordinal range their first occurrence lies in, (3) an ordered count over the flags gives
 *           every new k-mer its index, writes value[] in order and leaves a rank record per 64 ordinals,
 *           (3b) the rank lookups turn the uniques' ordinals into indices slice by slice, an XCD reading
 *           one slice's rank records from its own L2, (4) one workgroup per bucket merges the bucket's
 *           unique k-mers into its LDS copy of the table bucket and streams it back.
 */
#include <stdlib.h>
#include <string.h>
#include "mg_common.h"
#include <type_traits>

#define MG_ASSIGNED 0x80000000u
#define MG_R_QUANTUM 64u              /* slots per bucket come in multiples of this (a bucket starts on a 1 KiB boundary) */
/* -DMG_BUILD_PRIO: the build's kernels raise their waves' issue priority (s_setprio 3).  An experiment of round 3 for running
 * them beside the instruction-bound scan of the next batch on a second stream ("scan || build"): with it the
 * co-run gains 9 % over no priority, but it still takes 93 % of the sum of the two (the build's kernels need the wave slots the
 * scan holds), and alone the partition scatter loses 9 % and the merge 5 % to it -- so it is off. */
#ifdef MG_BUILD_PRIO
#define MG_BUILD_PRIO() __builtin_amdgcn_s_setprio (3)
#else
#define MG_BUILD_PRIO() do { } while (0)
#endif
#ifdef MG_ABLATE
#define MG_ABLATE_AND(x) && (x)
#else
#define MG_ABLATE_AND(x)
#endif
__device__ __forceinline__ U32 mgToken (U64 o) { return 0x7fffffffu - (U32) o; }     /* 1..0x7fffffff */
__device__ __forceinline__ bool mgIsAssigned (U32 v) { return (v & MG_ASSIGNED) != 0; }

/* bucket geometry: MgGeom, mgMixK, mgBucketOfM, mgHomeOfM in mg_common.h */

/* counters[]: 0 = new entries this call, 1 = bucket overflow flag */

/* ======================================================================================== */
/* direct path                                                                                */

__global__ void mgTableInsertKernel (MgSlot *__restrict__ slots, MgGeom g, const U64 *__restrict__ kmer, U64 n,
                                     U32 *__restrict__ slotId, int withDepth, U64 *counters)
{
  U64 o = (U64) blockIdx.x * blockDim.x + threadIdx.x;
  const U64 stride = (U64) gridDim.x * blockDim.x;
  for ( ; o < n ; o += stride)
    { const U64 m = mgMixK (kmer[o], g.kbits), key = m + 1;
      const U64 base = (U64) mgBucketOfM (m, g) * g.R;
      U32 at = mgHomeOfM (m, g);
      bool ok = false;
      for (U32 probes = 0 ; probes < g.R ; ++probes)
        { U64 cur = slots[base + at].key;          /* plain load: a stale "empty" is repaired by the CAS */
          if (cur == 0)
            { cur = atomicCAS ((unsigned long long *) &slots[base + at].key, 0ull, (unsigned long long) key);
              if (cur == 0) cur = key;
            }
          if (cur == key) { ok = true; break; }
          at = mgNextSlot (at, g.R);
        }
      if (!ok) { counters[1] = 1; slotId[o] = 0xffffffffu; continue; }
      const U64 s = base + at;
      U32 v = slots[s].ord;                        /* tokens only grow: a stale read is only ever too small */
      U32 tok = mgToken (o);
      if (v < tok) atomicMax (&slots[s].ord, tok); /* an assigned index (bit 31) is never below a token */
      if (withDepth)
        { /* the lanes of the wave that sit in the same slot as its first one add their number once: a run of one k-mer (64 consecutive
             modimizers of a poly-A stretch are one) would otherwise queue 64 atomics on one word -- 12 ns per occurrence */
          const U64 s0 = ((U64) (U32) __builtin_amdgcn_readfirstlane ((int) (U32) (s >> 32)) << 32) | (U32) __builtin_amdgcn_readfirstlane ((int) (U32) s);
          const bool mine = s == s0;
          const U64 grp = __ballot (mine);
          if (!mine) atomicAdd (&slots[s].cnt, 1u);
          else if ((threadIdx.x & 63) == (U32) __builtin_ctzll (grp)) atomicAdd (&slots[s].cnt, (U32) __popcll (grp));
        }
      slotId[o] = (U32) s;
    }
}

/* Ordered count of first-occurrence flags.  The ordinal range is cut into contiguous pieces, one
 * per wave ("rank unit"); a wave walks its piece in rows of 64 ordinals, so flags, k-mers and the
 * value[] it writes are all lane-contiguous.  Pass A counts per unit, a one-block scan turns the
 * counts into bases, pass B assigns. */
struct __attribute__ ((aligned (16))) MgRankGrp { U64 bits; U32 rank; U32 pad; };   /* per 64 ordinals */

__global__ __launch_bounds__ (256)
void mgDirectFlagKernel (const MgSlot *__restrict__ slots, const U32 *__restrict__ slotId, U64 n,
                         unsigned char *__restrict__ flags)
{
  U64 o = (U64) blockIdx.x * blockDim.x + threadIdx.x;
  const U64 stride = (U64) gridDim.x * blockDim.x;
  for ( ; o < n ; o += stride)
    { U32 s = slotId[o];
      flags[o] = (s != 0xffffffffu && slots[s].ord == mgToken (o)) ? 1 : 0;
    }
}


__global__ __launch_bounds__ (256)
void mgRankCountKernel (const unsigned char *__restrict__ flags, U64 n, U64 rowsPerUnit, U64 *__restrict__ unitCount)
{
  MG_BUILD_PRIO ();
  const int lane = threadIdx.x & 63;
  const U64 unit = (U64) blockIdx.x * 4 + (threadIdx.x >> 6);
  const U64 nRows = (n + 63) / 64;
  U64 row = unit * rowsPerUnit, rEnd = row + rowsPerUnit;
  if (rEnd > nRows) rEnd = nRows;
  U32 c = 0;
  /* a lane reads 16 consecutive flags at once (one 16-byte load: a wave covers 16 rows), four such loads in flight */
  if ((reinterpret_cast<uintptr_t> (flags) & 15) == 0)
    for ( ; row + 64 <= rEnd && (row + 64) * 64 <= n ; row += 64)
      { uint4 q[4];
#pragma unroll
        for (int j = 0 ; j < 4 ; ++j) q[j] = *reinterpret_cast<const uint4 *> (flags + (row + 16 * j) * 64 + (U64) lane * 16);
#pragma unroll
        for (int j = 0 ; j < 4 ; ++j)
          c += (U32) __popc (q[j].x & 0x01010101u) + (U32) __popc (q[j].y & 0x01010101u) + (U32) __popc (q[j].z & 0x01010101u) + (U32) __popc (q[j].w & 0x01010101u);
      }
  for ( ; row + 4 <= rEnd ; row += 4)
    { U32 v[4];
#pragma unroll
      for (int j = 0 ; j < 4 ; ++j) { U64 o = (row + j) * 64 + lane; v[j] = (o < n) ? (flags[o] & 1) : 0; }
      c += v[0] + v[1] + v[2] + v[3];
    }
  for ( ; row < rEnd ; ++row) { U64 o = row * 64 + lane; c += (o < n) ? (flags[o] & 1) : 0; }
  for (int off = 32 ; off ; off >>= 1) c += __shfl_xor (c, off);
  if (lane == 0) unitCount[unit] = c;
}

/* exclusive scan of up to a few thousand block counts (one workgroup); counters[0] = total */
__global__ __launch_bounds__ (1024)
void mgRankScanKernel (const U64 *__restrict__ blockCount, U32 nBlocks, U64 *__restrict__ blockBase, U64 *__restrict__ counters)
{
  __shared__ U64 sPart[1024];
  const int tid = threadIdx.x;
  const U32 per = (nBlocks + 1023) / 1024;
  U64 sum = 0;
  for (U32 i = 0 ; i < per ; ++i) { U32 b = tid * per + i; if (b < nBlocks) sum += blockCount[b]; }
  sPart[tid] = sum;
  __syncthreads ();
  for (int off = 1 ; off < 1024 ; off <<= 1)
    { U64 v = tid >= off ? sPart[tid - off] : 0;
      __syncthreads ();
      sPart[tid] += v;
      __syncthreads ();
    }
  U64 run = sPart[tid] - sum;
  for (U32 i = 0 ; i < per ; ++i) { U32 b = tid * per + i; if (b < nBlocks) { blockBase[b] = run; run += blockCount[b]; } }
  if (tid == 1023) counters[0] = sPart[1023];
}

/* ---- k-mers read from the scan's segments (MgSegSrc) instead of a dense array ---------------------------------
 * A wave reads 64 consecutive ordinals at a time; they lie in one segment, rarely in two or more.  The wave keeps the
 * segment of its first ordinal in scalar registers (w, its first ordinal s0 and the next segment's s1) and walks it
 * forward as its ordinals grow; a lane behind s1 walks on by itself. */
#ifndef MG_PART_SUB
#define MG_PART_SUB 8192
#endif                            /* elements of a sub-chunk of the partition passes (LDS counting sort) */
struct MgSegCursor { U32 w; U64 s0, s1; };
__device__ __forceinline__ U64 mgUniform64 (U64 x)
{ return ((U64) (U32) __builtin_amdgcn_readfirstlane ((int) (U32) (x >> 32)) << 32) | (U32) __builtin_amdgcn_readfirstlane ((int) (U32) x); }
/* the segment holding ordinal o (o < total): first w with segStart[w + 1] > o */
__device__ __forceinline__ U32 mgSegOfOrdinal (const MgSegSrc &src, U64 o)
{
  U32 a = 0, b = src.nSegs - 1;
  while (a < b) { const U32 m = (a + b) / 2; if (src.segStart[m + 1] > o) b = m; else a = m + 1; }
  return a;
}
/* (w and o0 are the same in every lane: saying so keeps the cursor and its loads on the scalar unit) */
__device__ __forceinline__ void mgSegCursorAt (const MgSegSrc &src, U32 w, MgSegCursor *c)
{ w = (U32) __builtin_amdgcn_readfirstlane ((int) w); c->w = w; c->s0 = mgUniform64 (src.segStart[w]); c->s1 = mgUniform64 (src.segStart[w + 1]); }
/* advance to the segment of o0 (uniform, o0 < total, not before the cursor) */
__device__ __forceinline__ void mgSegCursorSeek (const MgSegSrc &src, MgSegCursor *c, U64 o0)
{ o0 = mgUniform64 (o0); while (c->s1 <= o0) { ++c->w; c->s0 = c->s1; c->s1 = mgUniform64 (src.segStart[c->w + 1]); } }
/* address of this lane's ordinal i = o0 + lane, for a cursor at o0 (uniform) and `last` = the last ordinal any lane asks
 * for (uniform, < total).  The walk over the segments the 64 ordinals touch is the same in every lane and loads through
 * the scalar unit only: a vector load inside it would make the compiler drain the k-mer loads already in flight. */
__device__ __forceinline__ const U64 *mgSegAddr (const MgSegSrc &src, const MgSegCursor &c, U64 i, U64 last)
{
  U32 w = c.w; U64 b0 = c.s0, b1 = c.s1;
  const U64 *addr = src.segKmer + (U64) w * src.segCap + (i - b0);
  last = mgUniform64 (last);
  while (b1 <= last)
    { ++w; b0 = b1; b1 = mgUniform64 (src.segStart[w + 1]);
      if (i >= b0) addr = src.segKmer + (U64) w * src.segCap + (i - b0);
    }
  return addr;
}

/* subSeg[q] = the cursor at ordinal q * MG_PART_SUB (w, s0, s1): where the first partition pass starts a sub-chunk's
 * walk and the index assignment a wave's (one load instead of a binary search of dependent ones) */
struct __attribute__ ((aligned (8))) MgSubSeg { U64 w, s0, s1; };
__global__ void mgSubSegKernel (const MgSegSrc src, U64 n, U32 sub, MgSubSeg *__restrict__ subSeg)
{
  const U64 q = (U64) blockIdx.x * blockDim.x + threadIdx.x;
  if (q * sub >= n) return;
  const U32 w = mgSegOfOrdinal (src, q * sub);
  MgSubSeg e; e.w = w; e.s0 = src.segStart[w]; e.s1 = src.segStart[w + 1];
  subSeg[q] = e;
}
__device__ __forceinline__ void mgSegCursorFrom (const MgSubSeg *__restrict__ subSeg, U64 q, MgSegCursor *c)
{ const MgSubSeg e = subSeg[q]; c->w = (U32) __builtin_amdgcn_readfirstlane ((int) (U32) e.w); c->s0 = mgUniform64 (e.s0); c->s1 = mgUniform64 (e.s1); }

/* pass B: every flagged ordinal o gets index baseMax+1+rank(o); value[index] = kmer[o].
 * DIRECT: also store the index in the slot.  BUCKETED: record, per row of 64 ordinals, the flag
 * bitmap and the number of flags before the row, for mgRankLookupKernel's rank(o) lookups. */
#define MG_RANK_ROWS 8
template <bool DIRECT, bool SEG>
__global__ __launch_bounds__ (256)
void mgRankAssignKernel (const unsigned char *__restrict__ flags, const U64 *__restrict__ kmer, const MgSegSrc src, const MgSubSeg *__restrict__ subSeg,
                         U64 n, U64 rowsPerUnit,
                         const U64 *__restrict__ unitBase, U32 baseMax, U32 size,
                         U64 *__restrict__ value, MgSlot *__restrict__ slots, const U32 *__restrict__ slotId,
                         MgRankGrp *__restrict__ grp)
{
  MG_BUILD_PRIO ();
  const int lane = threadIdx.x & 63;
  const U64 unit = (U64) blockIdx.x * 4 + (U32) __builtin_amdgcn_readfirstlane ((int) (threadIdx.x >> 6));
  const U64 nRows = (n + 63) / 64;
  U64 row = unit * rowsPerUnit, rEnd = row + rowsPerUnit;
  if (rEnd > nRows) rEnd = nRows;
  if (row >= rEnd) return;
  U64 run = unitBase[unit];
  const U64 below = ((U64) 1 << lane) - 1;
  MgSegCursor cur; cur.w = 0; cur.s0 = 0; cur.s1 = 0;
  if (SEG) { mgSegCursorFrom (subSeg, row * 64 / MG_PART_SUB, &cur); mgSegCursorSeek (src, &cur, row * 64); }
  /* MG_RANK_ROWS rows per step: the loads of all four are in flight before the first ballot */
  for ( ; row < rEnd ; row += MG_RANK_ROWS)
    { bool f[MG_RANK_ROWS]; U64 km[MG_RANK_ROWS]; U32 fl[MG_RANK_ROWS];
      /* all the flag loads first, then all the k-mer loads, then the first use: with the segment walk's (uniform)
         branches between them the compiler otherwise waits for each row's flags -- and everything older -- in turn */
#pragma unroll
      for (int j = 0 ; j < MG_RANK_ROWS ; ++j)
        { const U64 o = (row + j) * 64 + lane;
          const bool in = (row + j < rEnd) && (o < n);
          fl[j] = flags[in ? o : 0];
        }
#pragma unroll
      for (int j = 0 ; j < MG_RANK_ROWS ; ++j)
        { const U64 o = (row + j) * 64 + lane;
          const bool in = (row + j < rEnd) && (o < n);
          if (SEG)
            { km[j] = 0;
              if ((row + j) * 64 < n && row + j < rEnd)                      /* uniform */
                { const U64 o0 = (row + j) * 64, last = o0 + 63 < n ? o0 + 63 : n - 1;
                  mgSegCursorSeek (src, &cur, o0);
                  const U64 *at = mgSegAddr (src, cur, o, last);
                  if (in) km[j] = __builtin_nontemporal_load (at);
                }
            }
          else km[j] = in ? __builtin_nontemporal_load (&kmer[o]) : 0;
        }
#pragma unroll
      for (int j = 0 ; j < MG_RANK_ROWS ; ++j)
        { const U64 o = (row + j) * 64 + lane;
          f[j] = (row + j < rEnd) && (o < n) && (fl[j] & 1);
          /* every load has landed before the first store goes out: with loads and stores both in flight the counter
             cannot tell them apart, and each row's store would wait for the stores of the rows before it */
          asm volatile ("" : "+v" (km[j]));
        }
#pragma unroll
      for (int j = 0 ; j < MG_RANK_ROWS ; ++j)
        { if (row + j >= rEnd) break;
          const U64 o = (row + j) * 64 + lane;
          const U64 bits = __ballot (f[j]);
          if (!DIRECT && lane == 0) { MgRankGrp g; g.bits = bits; g.rank = (U32) run; g.pad = 0; grp[row + j] = g; }
          if (f[j])
            { U64 idx = (U64) baseMax + 1 + run + (U32) __popcll (bits & below);
              if (idx < size)
                { __builtin_nontemporal_store (km[j], &value[idx]);
                  if (DIRECT) slots[slotId[o]].ord = (U32) idx | MG_ASSIGNED;
                }
            }
          run += (U32) __popcll (bits);
        }
    }
}

/* The query's lookups.  MG_FIND_PER k-mers per thread and step: all the k-mer loads, then all the first probes (one
 * 16-byte load each: key and index together), land before anything is stored -- a loop of "load, probe, store" made
 * every store wait for the one before it; only the probes that hit another k-mer's slot walk on.  CHECK_OCC = false
 * when every bucket's bytes are defined (mgTableClean has zeroed the never-written ones): no dependent load in front. */
#ifndef MG_FIND_PER
#define MG_FIND_PER 1
#endif
template <bool CHECK_OCC>
__global__ __launch_bounds__ (256)
void mgTableFindKernel (const MgSlot *__restrict__ slots, const U32 *__restrict__ occ, MgGeom g,
                        const U64 *__restrict__ kmer, U64 n, U32 *__restrict__ out)
{
  MG_BUILD_PRIO ();
  const U64 stride = (U64) gridDim.x * blockDim.x;
  for (U64 o0 = (U64) blockIdx.x * blockDim.x + threadIdx.x ; o0 < n ; o0 += stride * MG_FIND_PER)
    { U64 km[MG_FIND_PER], key[MG_FIND_PER], base[MG_FIND_PER]; U32 at[MG_FIND_PER], res[MG_FIND_PER]; bool live[MG_FIND_PER];
      uint4 v[MG_FIND_PER];
#pragma unroll
      for (int j = 0 ; j < MG_FIND_PER ; ++j) { const U64 o = o0 + (U64) j * stride; km[j] = o < n ? __builtin_nontemporal_load (&kmer[o]) : 0; }
#pragma unroll
      for (int j = 0 ; j < MG_FIND_PER ; ++j)
        { const U64 m = mgMixK (km[j], g.kbits);
          const U32 bkt = mgBucketOfM (m, g);
          key[j] = m + 1; base[j] = (U64) bkt * g.R; at[j] = mgHomeOfM (m, g);
          live[j] = o0 + (U64) j * stride < n;
          if (CHECK_OCC && live[j] && !occ[bkt]) live[j] = false;      /* never written: its bytes are undefined */
          v[j] = make_uint4 (0, 0, 0, 0);
          if (live[j]) v[j] = *reinterpret_cast<const uint4 *> (&slots[base[j] + at[j]]);
        }
#pragma unroll
      for (int j = 0 ; j < MG_FIND_PER ; ++j) asm volatile ("" : "+v" (v[j].x), "+v" (v[j].y), "+v" (v[j].z));
#pragma unroll
      for (int j = 0 ; j < MG_FIND_PER ; ++j)
        { res[j] = 0;
          U64 cur = ((U64) v[j].y << 32) | v[j].x; U32 ord = v[j].z;
          for (U32 probes = 1 ; live[j] && cur != 0 ; ++probes)
            { if (cur == key[j]) { res[j] = mgIsAssigned (ord) ? (ord & ~MG_ASSIGNED) : 0; break; }
              if (probes >= g.R) break;
              at[j] = mgNextSlot (at[j], g.R);
              const uint4 w = *reinterpret_cast<const uint4 *> (&slots[base[j] + at[j]]);
              cur = ((U64) w.y << 32) | w.x; ord = w.z;
            }
        }
#pragma unroll
      for (int j = 0 ; j < MG_FIND_PER ; ++j) { const U64 o = o0 + (U64) j * stride; if (o < n) __builtin_nontemporal_store (res[j], &out[o]); }
    }
}

/* the same lookups for k-mers that still sit in the scan's segments: a wave walks a contiguous range of rows of 64
 * ordinals with a segment cursor (see MgSegCursor) */
__global__ __launch_bounds__ (256)
void mgTableFindSegKernel (const MgSlot *__restrict__ slots, MgGeom g, const MgSegSrc src, U64 n, U64 rowsPerWave, U32 *__restrict__ out)
{
  MG_BUILD_PRIO ();
  const int lane = threadIdx.x & 63;
  const U64 wave = (U64) blockIdx.x * 4 + (U32) __builtin_amdgcn_readfirstlane ((int) (threadIdx.x >> 6));
  const U64 nRows = (n + 63) / 64;
  U64 row = wave * rowsPerWave, rEnd = row + rowsPerWave;
  if (rEnd > nRows) rEnd = nRows;
  if (row >= rEnd) return;
  MgSegCursor cur;
  mgSegCursorAt (src, mgSegOfOrdinal (src, row * 64), &cur);
  for ( ; row < rEnd ; ++row)
    { const U64 o0 = row * 64, o = o0 + (U64) lane, last = o0 + 63 < n ? o0 + 63 : n - 1;
      mgSegCursorSeek (src, &cur, o0);
      const U64 *at = mgSegAddr (src, cur, o < n ? o : last, last);
      if (o >= n) continue;
      const U64 m = mgMixK (__builtin_nontemporal_load (at), g.kbits), key = m + 1;
      const U64 base = (U64) mgBucketOfM (m, g) * g.R;
      U32 slot = mgHomeOfM (m, g), res = 0;
      for (U32 probes = 0 ; probes < g.R ; ++probes)
        { const uint4 w = *reinterpret_cast<const uint4 *> (&slots[base + slot]);
          const U64 cur64 = ((U64) w.y << 32) | w.x;
          if (cur64 == key) { res = mgIsAssigned (w.z) ? (w.z & ~MG_ASSIGNED) : 0; break; }
          if (cur64 == 0) break;
          slot = mgNextSlot (slot, g.R);
        }
      __builtin_nontemporal_store (res, &out[o]);
    }
}

/* entries first..last (with their existing indices) from a host modset into the device table */
__global__ void mgTableLoadKernel (MgSlot *__restrict__ slots, MgGeom g, const U64 *__restrict__ value,
                                   U32 first, U32 last, U32 *__restrict__ occ, U64 *counters)
{
  U64 i = (U64) first + (U64) blockIdx.x * blockDim.x + threadIdx.x;
  const U64 stride = (U64) gridDim.x * blockDim.x;
  for ( ; i <= last ; i += stride)
    { const U64 m = mgMixK (value[i], g.kbits), key = m + 1;
      const U32 b = mgBucketOfM (m, g);
      const U64 base = (U64) b * g.R;
      U32 at = mgHomeOfM (m, g);
      bool placed = false, dup = false;
      for (U32 probes = 0 ; probes < g.R ; ++probes)
        { U64 cur = slots[base + at].key;
          if (cur == 0)
            { cur = atomicCAS ((unsigned long long *) &slots[base + at].key, 0ull, (unsigned long long) key);
              if (cur == 0) { placed = true; break; }
            }
          if (cur == key) { dup = true; break; }     /* duplicate value in the host arrays: keep the first */
          at = mgNextSlot (at, g.R);
        }
      if (placed) { slots[base + at].ord = (U32) i | MG_ASSIGNED; atomicAdd (&occ[b], 1u); }
      else if (!dup) counters[1] = 1;
    }
}

/* ======================================================================================== */
/* whole-table streaming passes (export / histogram)                                          */

/* pending depth counts -> delta16[idx-1], folded into baseDepth, cnt zeroed.  A workgroup takes whole buckets (R is no power of two:
   one occupancy test per bucket instead of a division per slot) */
__global__ void mgTableExportDepthKernel (MgSlot *__restrict__ slots, U32 nBuckets, const U32 *__restrict__ occ, U32 R,
                                          U16 *__restrict__ baseDepth, U16 *__restrict__ delta, U32 max)
{
  for (U32 bk = blockIdx.x ; bk < nBuckets ; bk += gridDim.x)
    { if (!occ[bk]) continue;
      MgSlot *base = slots + (U64) bk * R;
      for (U32 i = threadIdx.x ; i < R ; i += blockDim.x)
        { uint4 v = *reinterpret_cast<const uint4 *> (&base[i]);
          if (!(v.x | v.y) || !mgIsAssigned (v.z)) continue;
          U32 idx = v.z & ~MG_ASSIGNED, c = v.w;
          if (idx > max) continue;
          U32 cl = c > 0xffffu ? 0xffffu : c;
          delta[idx - 1] = (U16) cl;
          U32 b = (U32) baseDepth[idx] + cl;
          baseDepth[idx] = (U16) (b > 0xffffu ? 0xffffu : b);
          if (c) base[i].cnt = 0;
        }
    }
}

/* K5: histogram of min(65535, baseDepth + pending) over all entries (modutils.c:53-63) */
#define MG_HIST_LDS_BINS 8192
__global__ __launch_bounds__ (256)
void mgTableHistKernel (const MgSlot *__restrict__ slots, U32 nBuckets, const U32 *__restrict__ occ, U32 R,
                        const U16 *__restrict__ baseDepth, unsigned long long *__restrict__ hist)
{
  __shared__ U32 sBins[MG_HIST_LDS_BINS];
  for (int b = threadIdx.x ; b < MG_HIST_LDS_BINS ; b += blockDim.x) sBins[b] = 0;
  __syncthreads ();
  /* a workgroup takes whole buckets (one occupancy test per bucket) and keeps four 16-byte loads per lane in
     flight; whole waves run every step, so the two commonest bins (depth 1 and 2: sequencing errors) are counted
     per wave with ballots instead of 64 colliding LDS atomics */
  const int lane = threadIdx.x & 63;
  U32 n1 = 0, n2 = 0;
  for (U32 bk = blockIdx.x ; bk < nBuckets ; bk += gridDim.x)
    { if (!occ[bk]) continue;
      const MgSlot *base = slots + (U64) bk * R;
      for (U32 i0 = 0 ; i0 < R ; i0 += 4 * blockDim.x)
        { uint4 v[4];
#pragma unroll
          for (int j = 0 ; j < 4 ; ++j)
            { const U32 i = i0 + j * blockDim.x + threadIdx.x;
              v[j] = i < R ? *reinterpret_cast<const uint4 *> (&base[i]) : make_uint4 (0, 0, 0, 0);
            }
#pragma unroll
          for (int j = 0 ; j < 4 ; ++j)
            { U32 d = 0; bool have = false;
              if ((v[j].x | v[j].y) && mgIsAssigned (v[j].z))
                { U32 idx = v[j].z & ~MG_ASSIGNED;
                  d = (baseDepth ? (U32) baseDepth[idx] : 0u) + v[j].w;        /* baseDepth == 0: known to be all zero */
                  if (d > 0xffffu || d < v[j].w) d = 0xffffu;
                  have = true;
                }
              n1 += (U32) __popcll (__ballot (have && d == 1));
              n2 += (U32) __popcll (__ballot (have && d == 2));
              if (have && d != 1 && d != 2)
                { if (d < MG_HIST_LDS_BINS) atomicAdd (&sBins[d], 1u);
                  else atomicAdd (&hist[d], 1ull);
                }
            }
        }
    }
  if (lane == 0) { if (n1) atomicAdd (&sBins[1], n1); if (n2) atomicAdd (&sBins[2], n2); }
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

/* ======================================================================================== */
/* bucketed path, step 1: radix partition of (kmer, ordinal) by bucket id                     */

#define MG_PART_CHUNK (2 * MG_PART_SUB)          /* elements per workgroup pass, at least (a pass with larger sub-chunks takes two of those) */
#define MG_PART_MAXBINS 512

/* What a partition pass reads and writes.  The first pass reads the dense k-mers (an element's ordinal is its
 * position) and mixes them; from then on everything is in mixed space.  Two element formats:
 *   wide    8 bytes of mixed k-mer + 4 bytes of ordinal (separate arrays): any k, any batch size;
 *   packed  one 8-byte word (rem << ordBits) | ordinal, where rem = the mixed k-mer WITHOUT its coarse digit (the
 *           bin the element sits in implies those hiB bits: the mix is a bijection and the bucket id is a prefix of
 *           it) and ordBits = bits of the batch's largest ordinal.  Fits when 2k - hiB + ordBits <= 64, e.g. k = 21
 *           with 2^28 modimizers: 34 + 28 bits.  A third less traffic in both passes and in the dedup kernel. */
#define MG_EL_DENSE  0      /* input of the first pass: k-mers, ordinal = index */
#define MG_EL_WIDE   1      /* (mixed k-mer, ordinal) */
#define MG_EL_PACKED 2      /* (rem << ordBits) | ordinal */
#define MG_EL_SEG    3      /* input of the first pass when the k-mers still sit in the scan's segments (MgSegSrc): ordinal = position in the concatenation */
struct MgPartFmt { int ordBits, remBits, loB; };      /* remBits = 2k - hiB; loB = bits of the fine digit */

/* digit of an element = bits [shift, shift+log2(nBins)) of its bucket id */
template <int MODE>
__device__ __forceinline__ U32 mgDigitOf (U64 x, const MgGeom &g, const MgPartFmt &f, int shift, U32 binMask)
{
  if (MODE == MG_EL_DENSE)  return (mgBucketOfM (mgMixK (x, g.kbits), g) >> shift) & binMask;
  if (MODE == MG_EL_WIDE)   return (mgBucketOfM (x, g) >> shift) & binMask;
  return (U32) (x >> (f.ordBits + f.remBits - f.loB)) & binMask;      /* packed: only ever asked for the fine digit, the top loB bits of rem */
}

/* chunk -> (segment, range): segments are [segStart[s], segStart[s+1]); chunkBase[s] = first chunk of s,
 * chunkBase[nSeg] = number of chunks, and behind that table (at MG_CHUNK_SEG_AT) the segment of every chunk, so
 * that a workgroup moving to its next chunk does two rounds of loads, not a binary search of dependent ones */
#define MG_CHUNK_SEG_AT (MG_PART_MAXBINS + 2)
__device__ __forceinline__ bool mgChunkRange (const U64 *__restrict__ segStart, const U32 *__restrict__ chunkBase, U32 nSeg, U32 chunkElems,
                                              U32 chunk, U32 *seg, U64 *lo, U64 *hi)
{
  if (chunk >= chunkBase[nSeg]) return false;
  const U32 a = chunkBase[MG_CHUNK_SEG_AT + chunk];
  *seg = a;
  *lo = segStart[a] + (U64) (chunk - chunkBase[a]) * chunkElems;
  U64 e = segStart[a + 1];
  *hi = *lo + chunkElems < e ? *lo + chunkElems : e;
  return true;
}

__global__ __launch_bounds__ (MG_PART_MAXBINS)
void mgPartChunksKernel (const U64 *__restrict__ segStart, U32 nSeg, U32 chunkElems, U32 *__restrict__ chunkBase)
{
  __shared__ U32 sS[MG_PART_MAXBINS];
  const U32 t = threadIdx.x;
  U32 c = t < nSeg ? (U32) ((segStart[t + 1] - segStart[t] + chunkElems - 1) / chunkElems) : 0;
  sS[t] = c;
  __syncthreads ();
  for (int off = 1 ; off < MG_PART_MAXBINS ; off <<= 1)
    { U32 v = t >= (U32) off ? sS[t - off] : 0;
      __syncthreads ();
      sS[t] += v;
      __syncthreads ();
    }
  if (t < nSeg) chunkBase[t] = sS[t] - c;
  if (t == nSeg - 1) chunkBase[nSeg] = sS[t];
  /* the segment of every chunk: all threads share the chunks; sS[] holds the inclusive prefix, so the segment
     of chunk q is the first s with sS[s] > q */
  const U32 total = sS[(nSeg ? nSeg : 1) - 1];
  for (U32 q = t ; q < total ; q += MG_PART_MAXBINS)
    { U32 a = 0, b = nSeg - 1;
      while (a < b) { U32 m = (a + b) / 2; if (sS[m] > q) b = m; else a = m + 1; }
      chunkBase[MG_CHUNK_SEG_AT + q] = a;
    }
}

template <int MODE>
__global__ __launch_bounds__ (256)
void mgPartHistKernel (const U64 *__restrict__ kIn, MgGeom g, MgPartFmt f, int shift, U32 nBins,
                       const U64 *__restrict__ segStart, const U32 *__restrict__ chunkBase, U32 nSeg, U32 chunkElems,
                       U32 *__restrict__ binCount)
{
  MG_BUILD_PRIO ();
  __shared__ U32 sH[MG_PART_MAXBINS];
  /* a workgroup takes a contiguous run of chunks and adds its LDS counts to the global ones only when the
     segment changes (one workgroup per chunk meant thousands of atomics on each of a few hundred addresses) */
  const U32 nChunks = chunkBase[nSeg];
  const U32 per = (nChunks + gridDim.x - 1) / gridDim.x;
  U32 c = blockIdx.x * per;
  const U32 cEnd = c + per < nChunks ? c + per : nChunks;
  if (c >= cEnd) return;
  for (U32 b = threadIdx.x ; b < nBins ; b += 256) sH[b] = 0;
  __syncthreads ();
  U32 curSeg = 0xffffffffu;
  for ( ; c < cEnd ; ++c)
    { U32 seg; U64 lo, hi;
      if (!mgChunkRange (segStart, chunkBase, nSeg, chunkElems, c, &seg, &lo, &hi)) break;
      if (seg != curSeg && curSeg != 0xffffffffu)
        { __syncthreads ();
          for (U32 b = threadIdx.x ; b < nBins ; b += 256) { U32 v = sH[b]; if (v) { atomicAdd (&binCount[(U64) curSeg * nBins + b], v); sH[b] = 0; } }
          __syncthreads ();
        }
      curSeg = seg;
      /* eight loads per lane in flight before the first LDS add */
      for (U64 i0 = lo ; i0 < hi ; i0 += 8 * 256)
        { U64 v[8];
#pragma unroll
          for (int j = 0 ; j < 8 ; ++j) { U64 i = i0 + (U64) j * 256 + threadIdx.x; v[j] = i < hi ? kIn[i] : 0; }
#pragma unroll
          for (int j = 0 ; j < 8 ; ++j)
            { U64 i = i0 + (U64) j * 256 + threadIdx.x;
              if (i < hi) atomicAdd (&sH[mgDigitOf<MODE> (v[j], g, f, shift, nBins - 1)], 1u);
            }
        }
    }
  __syncthreads ();
  if (curSeg != 0xffffffffu)
    for (U32 b = threadIdx.x ; b < nBins ; b += 256) if (sH[b]) atomicAdd (&binCount[(U64) curSeg * nBins + b], sH[b]);
}

/* the same counts from the digit bytes the pass before left beside its output (mgPartScatterKernel dOut) */
__global__ __launch_bounds__ (256)
void mgPartHistBytesKernel (const unsigned char *__restrict__ dIn, U32 nBins,
                            const U64 *__restrict__ segStart, const U32 *__restrict__ chunkBase, U32 nSeg, U32 chunkElems,
                            U32 *__restrict__ binCount)
{
  MG_BUILD_PRIO ();
  __shared__ U32 sH[MG_PART_MAXBINS];
  const U32 nChunks = chunkBase[nSeg];
  const U32 per = (nChunks + gridDim.x - 1) / gridDim.x;
  U32 c = blockIdx.x * per;
  const U32 cEnd = c + per < nChunks ? c + per : nChunks;
  if (c >= cEnd) return;
  for (U32 b = threadIdx.x ; b < nBins ; b += 256) sH[b] = 0;
  __syncthreads ();
  U32 curSeg = 0xffffffffu;
  for ( ; c < cEnd ; ++c)
    { U32 seg; U64 lo, hi;
      if (!mgChunkRange (segStart, chunkBase, nSeg, chunkElems, c, &seg, &lo, &hi)) break;
      if (seg != curSeg && curSeg != 0xffffffffu)
        { __syncthreads ();
          for (U32 b = threadIdx.x ; b < nBins ; b += 256) { U32 v = sH[b]; if (v) { atomicAdd (&binCount[(U64) curSeg * nBins + b], v); sH[b] = 0; } }
          __syncthreads ();
        }
      curSeg = seg;
      /* the unaligned head byte by byte, then four digits a load */
      U64 i0 = lo;
      const U64 head = ((lo + 3) & ~(U64) 3) < hi ? ((lo + 3) & ~(U64) 3) : hi;
      if (i0 + threadIdx.x < head) atomicAdd (&sH[dIn[i0 + threadIdx.x]], 1u);
      i0 = head;
      const U64 nWords = (hi - i0) >> 2;
      const U32 *w = reinterpret_cast<const U32 *> (dIn + i0);
      for (U64 q0 = 0 ; q0 < nWords ; q0 += 8 * 256)
        { U32 v[8];
#pragma unroll
          for (int j = 0 ; j < 8 ; ++j) { const U64 q = q0 + (U64) j * 256 + threadIdx.x; v[j] = q < nWords ? w[q] : 0; }
#pragma unroll
          for (int j = 0 ; j < 8 ; ++j)
            { const U64 q = q0 + (U64) j * 256 + threadIdx.x;
              if (q < nWords)
                { atomicAdd (&sH[v[j] & 255u], 1u); atomicAdd (&sH[(v[j] >> 8) & 255u], 1u); atomicAdd (&sH[(v[j] >> 16) & 255u], 1u); atomicAdd (&sH[v[j] >> 24], 1u); }
            }
        }
      const U64 tail = i0 + (nWords << 2);
      if (tail + threadIdx.x < hi) atomicAdd (&sH[dIn[tail + threadIdx.x]], 1u);
    }
  __syncthreads ();
  if (curSeg != 0xffffffffu)
    for (U32 b = threadIdx.x ; b < nBins ; b += 256) if (sH[b]) atomicAdd (&binCount[(U64) curSeg * nBins + b], sH[b]);
}

/* per segment: binStart = segStart + exclusive scan of its bin counts; cursor = binStart */
__global__ __launch_bounds__ (MG_PART_MAXBINS)
void mgPartScanKernel (const U32 *__restrict__ binCount, U32 nBins, const U64 *__restrict__ segStart,
                       U64 *__restrict__ binStart, unsigned long long *__restrict__ cursor, U32 cstride, U32 nSeg, U64 n)
{
  __shared__ U32 sS[MG_PART_MAXBINS];
  const U32 seg = blockIdx.x, t = threadIdx.x;
  U32 c = t < nBins ? binCount[(U64) seg * nBins + t] : 0;
  sS[t] = c;
  __syncthreads ();
  for (int off = 1 ; off < MG_PART_MAXBINS ; off <<= 1)
    { U32 v = t >= (U32) off ? sS[t - off] : 0;
      __syncthreads ();
      sS[t] += v;
      __syncthreads ();
    }
  if (t < nBins)
    { U64 st = segStart[seg] + (sS[t] - c);
      binStart[(U64) seg * nBins + t] = st;
      cursor[((U64) seg * nBins + t) * cstride] = st;
    }
  if (seg == nSeg - 1 && t == 0) binStart[(U64) nSeg * nBins] = n;
}

/* Scatter with LDS staging: a sub-chunk of 4096 elements is counting-sorted by bin in LDS, so the
 * elements of one bin leave as one contiguous run written by consecutive lanes (plain scattered
 * 8-byte stores ran at ~22 G/s: 9 ms per 1.5e8 elements for the two passes). */
#ifndef MG_PART_THREADS
#define MG_PART_THREADS 1024
#endif
/* Sub-chunks of MG_PART_SUB_BIG elements where they fit the LDS -- one packed 8-byte word per element, at most 256 bins, so that a staged
 * element's digit is one byte: 152 KB.  A bin's run is then 64 elements (512 bytes) instead of 32, and the barriers, the scan of the
 * counts and the reservations are paid once per 16384 elements: the two passes 1.335 -> 1.235 ms at config 2. */
#define MG_PART_SUB_BIG (2 * MG_PART_SUB)
#define MG_PART_BIG_BINS 256
/* the elements [sub, subHi) of a sub-chunk into registers, MG_PART_THREADS apart */
template <int INMODE, int SUB>
__device__ __forceinline__ void mgPartFetch (U64 *km, U32 *tk, const U64 *__restrict__ kIn, const U32 *__restrict__ tIn,
                                             const MgSegSrc &src, const MgSubSeg *__restrict__ subSeg, U64 sub, U64 subHi, int tid)
{
  if (INMODE == MG_EL_SEG)
    { const U32 wave = (U32) __builtin_amdgcn_readfirstlane (tid >> 6);
      const int lane = tid & 63;
      MgSegCursor cur;
      mgSegCursorFrom (subSeg, sub / MG_PART_SUB, &cur);            /* the first partition pass has one segment [0, n): sub is a multiple of MG_PART_SUB */
#pragma unroll
      for (int j = 0 ; j < SUB / MG_PART_THREADS ; ++j)
        { const U64 o0 = sub + (U64) j * MG_PART_THREADS + (U64) wave * 64;
          if (o0 >= subHi) break;                                     /* uniform */
          mgSegCursorSeek (src, &cur, o0);
          const U64 i = o0 + (U64) lane;
          const U64 *at = mgSegAddr (src, cur, i, o0 + 63 < subHi ? o0 + 63 : subHi - 1);
          if (i < subHi) km[j] = *at;
        }
      return;
    }
#pragma unroll
  for (int j = 0 ; j < SUB / MG_PART_THREADS ; ++j)
    { U64 i = sub + (U64) j * MG_PART_THREADS + tid;
      if (i < subHi) { km[j] = kIn[i]; if (INMODE == MG_EL_WIDE) tk[j] = tIn[i]; }
    }
}

template <int INMODE, bool PACKOUT, int SUB>     /* INMODE: what kIn holds; PACKOUT: one packed word out, otherwise (mixed k-mer, ordinal); SUB: elements of a sub-chunk */
__global__ __launch_bounds__ (MG_PART_THREADS)
void mgPartScatterKernel (const U64 *__restrict__ kIn, const U32 *__restrict__ tIn, const MgSegSrc src, const MgSubSeg *__restrict__ subSeg,
                          MgGeom g, MgPartFmt f, int shift, U32 nBins,
                          const U64 *__restrict__ segStart, const U32 *__restrict__ chunkBase, U32 nSeg, U32 chunkElems,
                          unsigned long long *__restrict__ cursor, U32 cstride, U64 *__restrict__ kOut, U32 *__restrict__ tOut,
                          unsigned long long *__restrict__ runTab, int runMode, unsigned char *__restrict__ dOut, int dShift, U32 dMask)
{
  /* dOut (the first of two passes): the NEXT pass's digit of every element, one byte each, at the element's place -- the next pass
     counts its digits from these bytes instead of reading the 8-byte elements a second time (config 2: 0.28 -> 0.06 ms) */
  /* runMode (with runTab, the partitioned lookup): 0 = the run table's rows are the sub-chunks of ONE segment (row = sub / SUB);
     1 = the second level: rows are (chunk, half) = 2 c + (sub - lo) / SUB, and every element's ordinal field is replaced by its
     position in kIn -- where its result has to go back to */
  MG_BUILD_PRIO ();
  constexpr bool WIDE = !PACKOUT;
  constexpr bool BIG = SUB > MG_PART_SUB;
  constexpr int PER = SUB / MG_PART_THREADS;
  constexpr int BINS = BIG ? MG_PART_BIG_BINS : MG_PART_MAXBINS;
  static_assert (!BIG || PACKOUT, "large sub-chunks: packed elements only");
  typedef typename std::conditional<BIG, unsigned char, unsigned short>::type Digit;
  __shared__ U64 stK[SUB];
  __shared__ U32 stT[WIDE ? SUB : 1];
  __shared__ Digit stB[SUB];
  __shared__ U32 sH[BINS], sOff[BINS];
  __shared__ unsigned long long sBase[BINS];
  __shared__ U32 sWave[MG_PART_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const U64 remMask = f.remBits >= 64 ? ~0ull : (((U64) 1 << f.remBits) - 1);
  /* A workgroup walks chunks blockIdx.x, blockIdx.x + gridDim.x, ... sub-chunk by sub-chunk, and the elements
     of the next sub-chunk are already on their way into registers while the current one is written out (one
     workgroup fills a CU's LDS, so nothing else would hide that latency). */
  U32 c = blockIdx.x, seg; U64 lo, hi;
  bool have = mgChunkRange (segStart, chunkBase, nSeg, chunkElems, c, &seg, &lo, &hi);
  U64 sub = have ? lo : 0;
  U64 km[PER]; U32 tk[PER];
#pragma unroll
  for (int j = 0 ; j < PER ; ++j) { km[j] = 0; tk[j] = 0; }
  if (have)
    { const U64 subHi = sub + SUB < hi ? sub + SUB : hi;
      mgPartFetch<INMODE, SUB> (km, tk, kIn, tIn, src, subSeg, sub, subHi, tid);
    }
  while (have)
    { const U64 subHi = sub + SUB < hi ? sub + SUB : hi;
      const U32 cnt = (U32) (subHi - sub);
      /* what comes after this sub-chunk */
      U32 nc = c, nseg = seg; U64 nlo = lo, nhi = hi, nsub = sub + SUB; bool nhave = true;
      if (nsub >= hi) { nc = c + gridDim.x; nhave = mgChunkRange (segStart, chunkBase, nSeg, chunkElems, nc, &nseg, &nlo, &nhi); nsub = nlo; }
      for (U32 b = tid ; b < nBins ; b += MG_PART_THREADS) sH[b] = 0;
      __syncthreads ();
      U32 dr[PER];
#pragma unroll
      for (int j = 0 ; j < PER ; ++j)
        { U64 i = sub + (U64) j * MG_PART_THREADS + tid;
          dr[j] = 0xffffffffu;
          if (i < subHi)
            { constexpr bool FIRST = INMODE == MG_EL_DENSE || INMODE == MG_EL_SEG;
              if (FIRST)                                         /* into mixed space, once */
                { km[j] = mgMixK (km[j], g.kbits); tk[j] = (U32) i; }
              U32 d = mgDigitOf<(FIRST ? MG_EL_WIDE : INMODE)> (km[j], g, f, shift, nBins - 1);
              dr[j] = (d << 16) | atomicAdd (&sH[d], 1u);        /* rank within (sub-chunk, bin) */
              if (FIRST && PACKOUT) km[j] = ((km[j] & remMask) << f.ordBits) | (U64) tk[j];
              if (INMODE == MG_EL_PACKED && runMode == 1) km[j] = ((km[j] >> f.ordBits) << f.ordBits) | i;
            }
        }
      __syncthreads ();
      /* exclusive scan of the bin counts (two bins per thread) + reservation of the output runs */
      unsigned long long base0 = 0, base1 = 0;
      { U32 c0 = (U32) (2 * tid) < nBins ? sH[2 * tid] : 0, c1 = (U32) (2 * tid + 1) < nBins ? sH[2 * tid + 1] : 0;
        const U32 pair = c0 + c1, incl = mgWaveInclusiveSum (pair);
        if (lane == 63) sWave[wave] = incl;
        __syncthreads ();
        U32 wb = 0;
#pragma unroll
        for (int w = 0 ; w < MG_PART_THREADS / 64 ; ++w) if (w < wave) wb += sWave[w];
        U32 ex = wb + incl - pair;
        /* the two reservations are global atomics that return a value (a microsecond or two): they are issued here
           and only waited for after the staging below, which does not need them */
        if ((U32) (2 * tid) < nBins)
          { sOff[2 * tid] = ex;
            if (c0) base0 = atomicAdd (&cursor[((U64) seg * nBins + 2 * tid) * cstride], (unsigned long long) c0);
          }
        if ((U32) (2 * tid + 1) < nBins)
          { sOff[2 * tid + 1] = ex + c0;
            if (c1) base1 = atomicAdd (&cursor[((U64) seg * nBins + 2 * tid + 1) * cstride], (unsigned long long) c1);
          }
      }
      __syncthreads ();
#pragma unroll
      for (int j = 0 ; j < PER ; ++j)
        if (dr[j] != 0xffffffffu)
          { U32 d = dr[j] >> 16, p = sOff[d] + (dr[j] & 0xffffu);
            stK[p] = km[j]; if (WIDE) stT[p] = tk[j]; stB[p] = (Digit) d;
          }
      if ((U32) (2 * tid) < nBins) sBase[2 * tid] = base0;
      if ((U32) (2 * tid + 1) < nBins) sBase[2 * tid + 1] = base1;
      if (runTab)                                          /* (uniform; the partitioned lookup) where this sub-chunk's run of every bin went, and its length */
        { const U64 row = (runMode == 1 ? 2 * (U64) c + (sub - lo) / SUB : sub / SUB) * nBins;
          if ((U32) (2 * tid) < nBins) runTab[row + 2 * tid] = base0 | ((unsigned long long) sH[2 * tid] << 40);
          if ((U32) (2 * tid + 1) < nBins) runTab[row + 2 * tid + 1] = base1 | ((unsigned long long) sH[2 * tid + 1] << 40);
        }
      /* the registers are free: fetch the next sub-chunk */
      if (nhave)
        { const U64 nsubHi = nsub + SUB < nhi ? nsub + SUB : nhi;
          mgPartFetch<INMODE, SUB> (km, tk, kIn, tIn, src, subSeg, nsub, nsubHi, tid);
        }
      __syncthreads ();
      for (U32 p = tid ; p < cnt ; p += MG_PART_THREADS)
        { U32 d = stB[p];
          U64 at = sBase[d] + (p - sOff[d]);
          kOut[at] = stK[p];                               /* plain stores: the runs of a bin are short, the L2 combines them (non-temporal: 1.5 -> 1.85 ms) */
          if (WIDE) tOut[at] = stT[p];
          if (dOut) dOut[at] = (unsigned char) mgDigitOf<(PACKOUT ? MG_EL_PACKED : MG_EL_WIDE)> (stK[p], g, f, dShift, dMask);   /* (uniform) */
        }
      __syncthreads ();
      c = nc; seg = nseg; lo = nlo; hi = nhi; sub = nsub; have = nhave;
    }
}

/* ======================================================================================== */
/* bucketed path, steps 2 and 4: one workgroup per bucket, the bucket lives in LDS             */


struct MgBucketArgs {
  MgSlot *slots; MgGeom g; U32 nBuckets;
  const U64 *bucketStart;          /* [NB+1] ranges into pK/pT/pC */
  U64 *pK; U32 *pT; U32 *pC;       /* in: occurrences, wide (mixed k-mer in pK, ordinal in pT) or packed (pK alone); out (in place over
                                      pK, and in pT / pC): uniques (mixed k-mer, ord field, count) */
  MgPartFmt f;                     /* packed format */
  U32 *uniqCount;                  /* [NB] */
  U32 *occ;                        /* [NB] entries per bucket */
  unsigned char *flags;            /* [n] first occurrence of a k-mer new to the table */
  const MgRankGrp *grp; U32 baseMax; U32 size;
  /* a bucket's uniques leave the dedup kernel grouped by the SLICE of the ordinal range their first occurrence lies in
     (groups 0..nSlices-1: new k-mers of slice tok >> sliceShift; group nSlices: k-mers the table already holds), so that
     the rank lookups can run slice by slice against a piece of the rank records that stays in the L2 */
  unsigned short *sliceOff;        /* [NB x (nSlices + 2)]: start of group g in bucket b's list; [nSlices + 1] = the list's length */
  U32 nSlices; int sliceShift;
  int slotShift;                   /* != 0 (2k <= 48): a list entry carries, above bit slotShift of its mixed k-mer, the slot the k-mer holds in
                                      the dedup kernel's LDS image of the bucket -- a valid place in the bucket (the image starts from the
                                      table's own), so the merge kernel puts it there without probing */
  int markDup;                     /* which way round the flags are written: 0 = cleared by a memset, the dedup kernel sets the first occurrence
                                      of every new k-mer (one store per unique); 1 = preset to 1, it clears every occurrence that is NOT one (one
                                      store per duplicate: fewer when most of a batch's modimizers are new k-mers) */
  int withDepth;
  U64 *counters;
  /* oversize buckets (a k-mer with very many copies: poly-A, satellite monomers): a bucket of more than hotSplit occurrences is
     cut into chunks of mgHotChunkLen () occurrences, mgHotReduceKernel -- a workgroup per chunk -- reduces every chunk in place
     to weighted entries (one per distinct k-mer: its earliest ordinal in the chunk, its count in pC; pC = 0 ends a chunk's
     list), and the bucket's own workgroup in the dedup kernel walks the chunks' lists instead of the occurrences */
  U32 hotSplit, hotChunk;
  U64 *hotItems; unsigned long long *hotCount;     /* the chunks to reduce: (bucket << 32 | chunk), and how many ([0]; [1]: oversize buckets) */
  U32 *hotBuckets;                                 /* the oversize buckets, for the dedup kernel's HOT instance */
#ifdef MG_ABLATE
  int debug;                       /* ablation builds only (MODGPU_BUCKET_DEBUG): dedup: 1 no flag stores, 2 plain stores for max/add, 4 no claim loop, 8 no list stores; merge: 32 no claim loop, 64 no image stores; lookup: 16 no rank gathers, 128 no index stores, 256 no ordinal loads; results are wrong */
#endif
  unsigned long long *liveHist;    /* != 0: the merge kernel also counts the final depths of the entries it writes */
};

extern __shared__ __attribute__ ((aligned (16))) unsigned char mgDynLds[];

/* find-or-claim the LDS slot of key; returns R on overflow.  Linear probing: the slot is the first one of home, home + 1, ... (wrapping
 * inside the bucket) that holds the key or is empty.  This loop is what both bucket kernels spend their time in -- a workgroup waits at
 * its barrier for its LONGEST chain (8 links at load 0.38, 30 at 0.6, 80 at 0.75), every link a dependent LDS round trip -- and its code
 * shape counts (DESIGN_EXPERIMENTS.md section K: four slots a step, lanes walking independently, first probes issued together all made the
 * kernels slower).  The compare-and-swap IS the probe: one round trip where the slot is empty, not a read and then the swap (round 6:
 * dedup 1.32 -> 1.25 ms, merge 1.18 -> 1.07). */
__device__ __forceinline__ U32 mgLdsClaim (unsigned long long *sKey, U32 R, U32 home, unsigned long long key)
{
  U32 at = home;
  for (U32 probes = 0 ; probes < R ; ++probes)
    { const unsigned long long cur = atomicCAS (&sKey[at], 0ull, key);
      if (cur == 0 || cur == key) return at;
      at = mgNextSlot (at, R);
    }
  return R;
}

/* Both bucket kernels run a workgroup over a contiguous range of buckets.  Per bucket there is
 * little arithmetic and several dependent global round trips (bounds -> elements -> rank record),
 * so the next bucket's bounds and first elements are fetched into registers while the current
 * bucket is processed, and the LDS image is kept all-zero between buckets by clearing exactly the
 * slots the closing sweep visits (no 64 KiB re-zeroing per bucket). */
#ifndef MG_BUCKET_PREFETCH
#define MG_BUCKET_PREFETCH 3         /* occurrences per thread of the dedup kernel fetched one bucket ahead: with 1024 threads a whole bucket of config 2 (1.42 -> 1.20 ms against 2) */
#endif
#ifndef MG_MERGE_PREFETCH
#define MG_MERGE_PREFETCH 2          /* uniques per thread of the merge kernel fetched one bucket ahead */
#endif
#ifndef MG_HOT_DEPTH
#define MG_HOT_DEPTH 4                /* occurrences per thread and turn of the dedup kernel's loop over what was not fetched ahead */
#endif
#define MG_RANK_GROUPS 64            /* most groups a bucket's list is cut into: slices of the ordinal range + 1 */
#define MG_SLOT_SHIFT 48             /* a list entry's slot sits above this bit of its mixed k-mer (when 2k <= 48) */
#define MG_DEDUP_PER 4               /* slots of the LDS image per thread of the dedup kernel: R <= 4 x threads ... */
#define MG_DEDUP_PER_BIG 8           /* ... or 8 (R = 8192: a table of 2^31 slots, i.e. more than 6.4e8 entries at table bits 32; one workgroup per CU, 128 registers) */
#define MG_LIVE_BINS 256              /* depths below this are counted in LDS by the merge kernel's live histogram */

/* step 2: dedup the bucket's occurrences; uniques written in place over the bucket's range */
/* one occurrence of bucket b -> (mixed k-mer, ordinal) */
template <bool PACKED>
__device__ __forceinline__ void mgOccurrence (const MgBucketArgs &a, U32 b, U64 x, U32 t, U64 *m, U32 *ord)
{
  if (PACKED)
    { *ord = (U32) (x & (((U64) 1 << a.f.ordBits) - 1));
      *m = ((U64) (b >> a.f.loB) << a.f.remBits) | (x >> a.f.ordBits);     /* the bin's coarse digit in front of rem */
    }
  else { *m = x; *ord = t; }
}

/* one more occurrence (ordinal ord) of the k-mer in LDS slot at: the earliest ordinal wins the slot's token (assigned
 * entries carry bit 31 and stay as they are).  markDup: whoever loses -- this occurrence, or the one that held the token --
 * is not a first occurrence of a new k-mer: its flag is cleared */
__device__ __forceinline__ void mgDedupCount (const MgBucketArgs &a, U32 *sOrd, U32 *sCnt, U32 at, U32 ord)
{
  const U32 tok = mgToken (ord);
  if (a.markDup)
    { const U32 old = atomicMax (&sOrd[at], tok);
      const U32 loser = old > tok ? tok : old;
      if (loser MG_ABLATE_AND (!(a.debug & 1))) a.flags[0x7fffffffu - loser] = 0;
    }
  else atomicMax (&sOrd[at], tok);
  atomicAdd (&sCnt[at], 1u);
}
/* the largest value over the 64 lanes of a wave (DPP, as mgWaveInclusiveSum; every lane must be active) */
__device__ __forceinline__ U32 mgWaveMax (U32 v)
{
#define MG_MAX_DPP(ctrl, rows) do { const U32 o_ = (U32) __builtin_amdgcn_update_dpp (0, (int) v, ctrl, rows, 0xf, false); v = o_ > v ? o_ : v; } while (0)
  MG_MAX_DPP (0x111, 0xf); MG_MAX_DPP (0x112, 0xf); MG_MAX_DPP (0x114, 0xf); MG_MAX_DPP (0x118, 0xf); MG_MAX_DPP (0x142, 0xa); MG_MAX_DPP (0x143, 0xc);
#undef MG_MAX_DPP
  return (U32) __builtin_amdgcn_readlane ((int) v, 63);
}
/* A bucket's occurrences beyond the ones fetched ahead -- only a bucket with a k-mer of very many copies has any (a poly-A
 * 21-mer is a modimizer at k = 21, d = 64, seed 17, and a human read set holds millions of them): a wave takes 64 of them at a
 * time, and the lanes that landed in the same slot as its first lane are counted as ONE occurrence with their number -- the
 * earliest of them stands for all at the slot's token, the others are not first occurrences (markDup: their flags are
 * cleared) -- instead of 64 atomics queueing on one LDS word (1.7 ns per occurrence: 1.7 s for a batch of nothing but poly-A).
 * Every lane of the wave calls this, `live` or not. */
__device__ __forceinline__ void mgDedupCountWave (const MgBucketArgs &a, U32 *sOrd, U32 *sCnt, U32 R, bool live, U32 at, U32 ord)
{
  live = live && at < R;
  const U32 at0 = (U32) __builtin_amdgcn_readfirstlane ((int) at);
  const bool mine = live && at == at0;
  const U32 n = (U32) __popcll (__ballot (mine));
  if (n > 1)                                                   /* (uniform) */
    { const U32 tok = mgToken (ord);
      const U32 best = mgWaveMax (mine ? tok : 0u);
      if (mine)
        { if (tok == best)
            { const U32 old = atomicMax (&sOrd[at], tok);
              if (a.markDup) { const U32 loser = old > tok ? tok : old; if (loser) a.flags[0x7fffffffu - loser] = 0; }
              atomicAdd (&sCnt[at], n);
            }
          else if (a.markDup) a.flags[ord] = 0;
        }
      else if (live) mgDedupCount (a, sOrd, sCnt, at, ord);
    }
  else if (live) mgDedupCount (a, sOrd, sCnt, at, ord);
}

/* A run [from, hi) of occurrences of bucket b into the LDS image, MG_HOT_DEPTH x T at a time; every lane stays in the
 * loop: the wave works together (see mgDedupCountWave).  Used by the dedup kernel for what it did not fetch ahead and by
 * mgHotReduceKernel for a chunk of an oversize bucket. */
template <bool PACKED>
__device__ __forceinline__ void mgDedupRun (const MgBucketArgs &a, U32 b, unsigned long long *sKey, U32 *sOrd, U32 *sCnt,
                                            U32 R, U32 T, U32 tid, U64 from, U64 hi)
{
  /* the occurrences beyond the ones fetched ahead (a bucket with a k-mer of very many copies), MG_HOT_DEPTH x T at a time; every
     lane stays in the loop: the wave works together */
  for (U64 i0 = from ; i0 < hi ; i0 += (U64) MG_HOT_DEPTH * T)
    { U64 x[MG_HOT_DEPTH]; U32 tx[MG_HOT_DEPTH]; bool live[MG_HOT_DEPTH];
#pragma unroll
      for (int j = 0 ; j < MG_HOT_DEPTH ; ++j)
        { const U64 i = i0 + (U64) j * T + tid;
          live[j] = i < hi;
          x[j] = live[j] ? a.pK[i] : 0; tx[j] = (!PACKED && live[j]) ? a.pT[i] : 0;
        }
      /* all of the wave's elements one k-mer (that of its first lane)?  Then one claim, one token, one count for all of them */
      const int kShift = PACKED ? a.f.ordBits : 0;
      const U64 x0 = mgUniform64 (x[0]); const U32 tx0 = (U32) __builtin_amdgcn_readfirstlane ((int) tx[0]);
      bool same = true;
#pragma unroll
      for (int j = 0 ; j < MG_HOT_DEPTH ; ++j) if (live[j] && (x[j] >> kShift) != (x0 >> kShift)) same = false;
      const bool first = __builtin_amdgcn_readfirstlane ((int) live[0]) != 0;
      bool done = false;
      if (first && __ballot (!same) == 0 MG_ABLATE_AND (!(a.debug & 6)))                     /* (uniform) */
        { U64 m; U32 o0; mgOccurrence<PACKED> (a, b, x0, tx0, &m, &o0);
          const U32 at = mgLdsClaim (sKey, R, mgHomeOfM (m, a.g), m + 1);          /* (every lane claims the same slot with the same key) */
          if (at < R)
            { U32 tok[MG_HOT_DEPTH], tmax = 0, cnt = 0;
#pragma unroll
              for (int j = 0 ; j < MG_HOT_DEPTH ; ++j)
                { const U32 ord = PACKED ? (U32) (x[j] & (((U64) 1 << a.f.ordBits) - 1)) : tx[j];
                  tok[j] = live[j] ? mgToken (ord) : 0u;
                  if (tok[j] > tmax) tmax = tok[j];
                  cnt += live[j] ? 1u : 0u;
                }
              const U32 best = mgWaveMax (tmax);
              const U32 total = (U32) __builtin_amdgcn_readlane ((int) mgWaveInclusiveSum (cnt), 63);
              if (cnt && tmax == best)                                                        /* (one lane: tokens are all different) */
                { const U32 old = atomicMax (&sOrd[at], best);
                  if (a.markDup) { const U32 loser = old > best ? best : old; if (loser) a.flags[0x7fffffffu - loser] = 0; }
                  atomicAdd (&sCnt[at], total);
                }
              if (a.markDup)
                {
#pragma unroll
                  for (int j = 0 ; j < MG_HOT_DEPTH ; ++j) if (live[j] && tok[j] != best) a.flags[0x7fffffffu - tok[j]] = 0;
                }
              done = true;
            }
        }
      if (!done)                                                                              /* (uniform) */
        {
#pragma unroll
          for (int j = 0 ; j < MG_HOT_DEPTH ; ++j)
            { U64 m = 0; U32 ord = 0, at = R;
              if (live[j])
                { mgOccurrence<PACKED> (a, b, x[j], tx[j], &m, &ord);
#ifdef MG_ABLATE
                  if (a.debug & 4) { at = mgHomeOfM (m, a.g); sKey[at] = m + 1; } else
#endif
                  at = mgLdsClaim (sKey, R, mgHomeOfM (m, a.g), m + 1);
                  if (at == R) a.counters[1] = 1;
                }
#ifdef MG_ABLATE
              if (a.debug & 2) { if (live[j] && at < R) { sOrd[at] = mgToken (ord); sCnt[at] = 1; } continue; }
#endif
              mgDedupCountWave (a, sOrd, sCnt, R, live[j], at, ord);
            }
        }
    }
}

#define MG_HOT_SPLIT_DEFAULT 16384u  /* occurrences above which a bucket is reduced chunk by chunk first (a bucket of config 2 holds 2400); measured on the repeat-genome probe: 8192 / 16384 / 32768 / 65536 -> 1.50 / 1.44 / 1.48 / 1.50 ms per Gbp */
#define MG_HOT_CHUNK_DEFAULT 4096u   /* occurrences of a chunk, at least */
#define MG_HOT_MAXCHUNKS 1024u       /* chunks of a bucket, at most: a bucket of 1e9 occurrences is 1024 chunks of 1e6 */
__host__ __device__ __forceinline__ U64 mgHotChunkLen (U64 cnt, U32 minChunk)
{ const U64 L = (cnt + MG_HOT_MAXCHUNKS - 1) / MG_HOT_MAXCHUNKS; return L < minChunk ? minChunk : L; }

/* the chunks of every oversize bucket, listed for mgHotReduceKernel */
__global__ __launch_bounds__ (256)
void mgHotPlanKernel (const MgBucketArgs a)
{
  const U32 b = blockIdx.x * 256 + threadIdx.x;
  if (b >= a.nBuckets) return;
  const U64 cnt = a.bucketStart[b + 1] - a.bucketStart[b];
  if (cnt <= a.hotSplit) return;
  const U64 L = mgHotChunkLen (cnt, a.hotChunk);
  const U32 nCh = (U32) ((cnt + L - 1) / L);
  const U64 at = atomicAdd (a.hotCount, (unsigned long long) nCh);
  for (U32 j = 0 ; j < nCh ; ++j) a.hotItems[at + j] = ((U64) b << 32) | j;
  a.hotBuckets[atomicAdd (a.hotCount + 1, 1ull)] = b;
}

/* one chunk of an oversize bucket -> its distinct k-mers as weighted entries, in place at the chunk's start.  The LDS image
 * is the dedup kernel's (keys of ONE bucket: at most R distinct, or the bucket overflows there too); it starts empty, so a
 * k-mer's token here is its earliest ordinal within the chunk, and with markDup every other occurrence of the chunk has its
 * flag cleared right here -- none of them can be a first occurrence. */
template <bool PACKED>
__global__ __launch_bounds__ (1024)
void mgHotReduceKernel (const MgBucketArgs a)
{
  const U32 R = a.g.R, T = blockDim.x, tid = threadIdx.x;
  unsigned long long *sKey = reinterpret_cast<unsigned long long *> (mgDynLds);
  U32 *sOrd = reinterpret_cast<U32 *> (mgDynLds + (size_t) R * 8);
  U32 *sCnt = sOrd + R;
  U32 *sN = sCnt + R;
  const U64 nItems = *a.hotCount;
  if (blockIdx.x >= nItems) return;
  for (U32 i = tid ; i < R ; i += T) { sKey[i] = 0; sOrd[i] = 0; sCnt[i] = 0; }
  if (tid == 0) sN[0] = 0;
  __syncthreads ();
  for (U64 w = blockIdx.x ; w < nItems ; w += gridDim.x)
    { const U64 item = a.hotItems[w];
      const U32 b = (U32) (item >> 32), j = (U32) item;
      const U64 lo = a.bucketStart[b], hi = a.bucketStart[b + 1];
      const U64 L = mgHotChunkLen (hi - lo, a.hotChunk);
      const U64 cs = lo + (U64) j * L, ce = cs + L < hi ? cs + L : hi;
      mgDedupRun<PACKED> (a, b, sKey, sOrd, sCnt, R, T, tid, cs, ce);
      __syncthreads ();
      for (U32 i = tid ; i < R ; i += T)
        { const unsigned long long key = sKey[i];
          if (key)
            { const U32 tok = sOrd[i], c = sCnt[i];
              sKey[i] = 0; sOrd[i] = 0; sCnt[i] = 0;
              const U32 at = atomicAdd (&sN[0], 1u);
              const U32 ord = 0x7fffffffu - tok;
              if (PACKED) a.pK[cs + at] = (((key - 1) & (((U64) 1 << a.f.remBits) - 1)) << a.f.ordBits) | ord;
              else { a.pK[cs + at] = key - 1; a.pT[cs + at] = ord; }
              a.pC[cs + at] = c;
            }
        }
      __syncthreads ();
      if (tid == 0) { const U32 nc = sN[0]; if (cs + nc < ce) a.pC[cs + nc] = 0; sN[0] = 0; }
      __syncthreads ();
    }
}

/* the dedup kernel's side of it: the weighted entries of an oversize bucket's chunks into the bucket's image; a wave takes
 * chunks w, w + T / 64, ... and reads a chunk's list 64 entries at a time up to its end (pC = 0, or the chunk's end) */
template <bool PACKED>
__device__ __forceinline__ void mgDedupHotBucket (const MgBucketArgs &a, U32 b, unsigned long long *sKey, U32 *sOrd, U32 *sCnt,
                                                  U32 R, U32 T, U32 tid, U64 lo, U64 hi)
{
  const U64 L = mgHotChunkLen (hi - lo, a.hotChunk);
  const U32 nCh = (U32) ((hi - lo + L - 1) / L);
  const U32 lane = tid & 63;
  for (U32 j = tid >> 6 ; j < nCh ; j += T >> 6)
    { const U64 cs = lo + (U64) j * L, ce = cs + L < hi ? cs + L : hi;
      for (U64 base = cs ; base < ce ; base += 64)
        { const U64 i = base + lane;
          const U32 c = i < ce ? a.pC[i] : 0u;
          const unsigned long long ended = __ballot (c == 0);
          const bool live = c != 0 && (ended == 0 || lane < (U32) __builtin_ctzll (ended));
          if (live)
            { U64 m; U32 ord;
              mgOccurrence<PACKED> (a, b, a.pK[i], PACKED ? 0u : a.pT[i], &m, &ord);
              const U32 at = mgLdsClaim (sKey, R, mgHomeOfM (m, a.g), m + 1);
              if (at == R) a.counters[1] = 1;
              else
                { const U32 tok = mgToken (ord);
                  const U32 old = atomicMax (&sOrd[at], tok);
                  if (a.markDup) { const U32 loser = old > tok ? tok : old; if (loser) a.flags[0x7fffffffu - loser] = 0; }
                  atomicAdd (&sCnt[at], c);
                }
            }
          if (ended) break;
        }
    }
}

/* HOT = false: every bucket but the oversize ones; HOT = true (its own launch, a few workgroups): the oversize buckets the plan
   listed, from their chunks' weighted entries -- a second instance so that the ordinary one keeps its registers (with the walk
   inlined as a branch the dedup of config 2 went from 1.22 to 1.27 ms, as a called function to 3.2) */
template <bool PACKED, bool SLOT, int PER, bool HOT>       /* SLOT: a.slotShift != 0; PER: slots of the image per thread, R <= PER x threads */
__global__ __launch_bounds__ (1024) __attribute__ ((amdgpu_waves_per_eu (PER == MG_DEDUP_PER ? 8 : 4)))      /* 64 registers: two workgroups per CU */
void mgBucketDedupKernel (const MgBucketArgs a, U32 bucketsPerBlock)
{
  MG_BUILD_PRIO ();
  const U32 R = a.g.R, T = blockDim.x, tid = threadIdx.x;
  unsigned long long *sKey = reinterpret_cast<unsigned long long *> (mgDynLds);
  U32 *sOrd = reinterpret_cast<U32 *> (mgDynLds + (size_t) R * 8);
  U32 *sCnt = sOrd + R;
  U32 *sGrp = sCnt + R;                            /* [MG_RANK_GROUPS] members of each group of the bucket's list */
  const U32 nGrp = a.nSlices + 1;
  U32 b = blockIdx.x * bucketsPerBlock, bEnd = b + bucketsPerBlock;
  U64 hotAt = blockIdx.x; const U64 nHot = HOT ? a.hotCount[1] : 0;
  if (HOT) { if (hotAt >= nHot) return; }
  else
    { if (bEnd > a.nBuckets) bEnd = a.nBuckets;
      if (b >= bEnd) return;
    }
  for (U32 i = tid ; i < R ; i += T) { sKey[i] = 0; sOrd[i] = 0; sCnt[i] = 0; }
  if (tid < MG_RANK_GROUPS) sGrp[tid] = 0;
  for ( ; ; )                                           /* (HOT: one listed bucket per turn; otherwise a single turn over the workgroup's range) */
  {
  if (HOT) { b = a.hotBuckets[hotAt]; bEnd = b + 1; }
  U64 lo = a.bucketStart[b], hi = a.bucketStart[b + 1];
  U32 occNow = a.occ[b];                          /* fetched one bucket ahead like the bounds: it gates a branch */
  U64 ck[MG_BUCKET_PREFETCH]; U32 ct[MG_BUCKET_PREFETCH];
#pragma unroll
  for (int j = 0 ; j < MG_BUCKET_PREFETCH ; ++j)
    { U64 i = lo + (U64) j * T + tid; ck[j] = 0; ct[j] = 0; if (!HOT && i < hi) { ck[j] = __builtin_nontemporal_load (&a.pK[i]); if (!PACKED) ct[j] = __builtin_nontemporal_load (&a.pT[i]); } }
  __syncthreads ();
  for ( ; b < bEnd ; ++b)
    { /* fetch the next bucket while this one is processed */
      U64 nlo = hi, nhi = hi;
      U64 nk[MG_BUCKET_PREFETCH]; U32 nt[MG_BUCKET_PREFETCH];
#pragma unroll
      for (int j = 0 ; j < MG_BUCKET_PREFETCH ; ++j) { nk[j] = 0; nt[j] = 0; }
      U32 occNext = 0;
      if (b + 1 < bEnd)
        { nhi = a.bucketStart[b + 2];
          occNext = a.occ[b + 1];
#pragma unroll
          for (int j = 0 ; j < MG_BUCKET_PREFETCH ; ++j)
            { U64 i = nlo + (U64) j * T + tid; if (i < nhi) { nk[j] = __builtin_nontemporal_load (&a.pK[i]); if (!PACKED) nt[j] = __builtin_nontemporal_load (&a.pT[i]); } }
        }
      if (hi == lo)
        { if (tid == 0) a.uniqCount[b] = 0;
          if (tid < nGrp + 1) a.sliceOff[(U64) b * (nGrp + 1) + tid] = 0;
        }
      else if (!HOT && hi - lo > a.hotSplit) { }          /* (uniform) an oversize bucket: the other instance's */
      else
        { if (occNow)
            { for (U32 i = tid ; i < R ; i += T)
                { uint4 v = *reinterpret_cast<const uint4 *> (&a.slots[(U64) b * R + i]);
                  sKey[i] = ((unsigned long long) v.y << 32) | v.x; sOrd[i] = v.z;
                }
              __syncthreads ();
            }
          if (HOT) mgDedupHotBucket<PACKED> (a, b, sKey, sOrd, sCnt, R, T, tid, lo, hi);   /* reduced to weighted entries by mgHotReduceKernel */
          else
          {
#pragma unroll
          for (int j = 0 ; j < MG_BUCKET_PREFETCH ; ++j)
            if (lo + (U64) j * T + tid < hi)
              { U64 m; U32 ord; mgOccurrence<PACKED> (a, b, ck[j], ct[j], &m, &ord);
                U32 at;
#ifdef MG_ABLATE
                if (a.debug & 4) { at = mgHomeOfM (m, a.g); sKey[at] = m + 1; } else
#endif
                at = mgLdsClaim (sKey, R, mgHomeOfM (m, a.g), m + 1);
                if (at == R) a.counters[1] = 1;
#ifdef MG_ABLATE
                else if (a.debug & 2) { sOrd[at] = mgToken (ord); sCnt[at] = 1; }
#endif
                else { mgDedupCount (a, sOrd, sCnt, at, ord); }
              }
          /* the occurrences beyond the ones fetched ahead (a bucket with a k-mer of very many copies) */
          mgDedupRun<PACKED> (a, b, sKey, sOrd, sCnt, R, T, tid, lo + (U64) MG_BUCKET_PREFETCH * T, hi);
          }
          __syncthreads ();
          /* the uniques leave grouped (see MgBucketArgs).  Every thread takes its slots of the image into registers
             (clearing them for the next bucket) and counts the groups' members -- a unique's place inside its group is
             what the add returns; the counts become places in the list; the entries go back into the (now free) LDS
             arrays in list order, and leave from there with coalesced stores */
          unsigned long long rk[PER]; U32 rc[PER], ro[PER], rp[PER];
#pragma unroll
          for (int j = 0 ; j < PER ; ++j)
            { const U32 i = (U32) j * T + tid;
              rk[j] = 0; rc[j] = 0; ro[j] = 0; rp[j] = 0;
              if (i < R) rk[j] = sKey[i];
              if (rk[j])
                { rc[j] = sCnt[i]; ro[j] = sOrd[i];
                  sKey[i] = 0; sOrd[i] = 0; sCnt[i] = 0;
                  rk[j] = (rk[j] - 1) | (SLOT ? (unsigned long long) i << MG_SLOT_SHIFT : 0ull);   /* the list holds the mixed k-mer (key - 1), under its slot where there is room */
                  if (rc[j]) rp[j] = atomicAdd (&sGrp[mgIsAssigned (ro[j]) ? a.nSlices : ((0x7fffffffu - ro[j]) >> a.sliceShift)], 1u);
                }
            }
          __syncthreads ();
          const int lane = (int) (tid & 63);
          const U32 gv = (U32) lane < nGrp ? sGrp[lane] : 0;            /* every wave scans the counts for itself */
          const U32 gIncl = mgWaveInclusiveSum (gv);
          const U32 gBase = gIncl - gv;
          const U32 total = (U32) __builtin_amdgcn_readlane ((int) gIncl, 63);
          if (tid < 64)
            { if ((U32) lane <= nGrp) a.sliceOff[(U64) b * (nGrp + 1) + lane] = (unsigned short) gBase;     /* [nGrp]: the list's length */
              if (lane == 0) a.uniqCount[b] = total;
            }
#pragma unroll
          for (int j = 0 ; j < PER ; ++j)
            { const U32 grp = mgIsAssigned (ro[j]) ? a.nSlices : ((0x7fffffffu - ro[j]) >> a.sliceShift);
              const U32 at = (U32) __shfl ((int) gBase, (int) (rc[j] ? grp : 0)) + rp[j];
              if (rc[j]) { sKey[at] = rk[j]; sOrd[at] = ro[j]; sCnt[at] = rc[j]; }
            }
          __syncthreads ();
          if (tid < MG_RANK_GROUPS) sGrp[tid] = 0;
          for (U32 i = tid ; i < total ; i += T)
            { const U64 k = sKey[i]; const U32 ord = sOrd[i], c = sCnt[i];
              sKey[i] = 0; sOrd[i] = 0; sCnt[i] = 0;
              if (true MG_ABLATE_AND (!(a.debug & 8)))
                { __builtin_nontemporal_store (k, &a.pK[lo + i]); __builtin_nontemporal_store (ord, &a.pT[lo + i]); __builtin_nontemporal_store (c, &a.pC[lo + i]); }
              if (!a.markDup && !mgIsAssigned (ord) MG_ABLATE_AND (!(a.debug & 1))) a.flags[0x7fffffffu - ord] = 1;
            }
          __syncthreads ();
        }
      lo = nlo; hi = nhi; occNow = occNext;
#pragma unroll
      for (int j = 0 ; j < MG_BUCKET_PREFETCH ; ++j) { ck[j] = nk[j]; ct[j] = nt[j]; }
    }
  if (!HOT) break;
  hotAt += gridDim.x;
  if (hotAt >= nHot) break;
  }
}

/* step 4: merge the bucket's uniques into the table bucket and stream it back */
__global__ __launch_bounds__ (1024)
void mgBucketMergeKernel (const MgBucketArgs a, U32 bucketsPerBlock)
{
  MG_BUILD_PRIO ();
  const U32 R = a.g.R, T = blockDim.x, tid = threadIdx.x;
  unsigned long long *sKey = reinterpret_cast<unsigned long long *> (mgDynLds);
  U32 *sOrd = reinterpret_cast<U32 *> (mgDynLds + (size_t) R * 8);
  U32 *sCnt = sOrd + R;
  U32 b = blockIdx.x * bucketsPerBlock, bEnd = b + bucketsPerBlock;
  if (bEnd > a.nBuckets) bEnd = a.nBuckets;
  if (b >= bEnd) return;
  for (U32 i = tid ; i < R ; i += T) { sKey[i] = 0; sOrd[i] = 0; sCnt[i] = 0; }
  U32 *sLive = sCnt + R + 4;                       /* MG_LIVE_BINS small-depth bins */
  for (U32 i = tid ; i < MG_LIVE_BINS ; i += T) sLive[i] = 0;
  U32 n1 = 0, n2 = 0;                              /* depth 1 and 2, the commonest, counted per wave */
  U32 nu = a.uniqCount[b];
  U32 nNew = a.sliceOff[(U64) b * (a.nSlices + 2) + a.nSlices];   /* the list's first nNew uniques are new to the table, the others are in it */
  U32 occNow = a.occ[b];                          /* one bucket ahead, like the counts */
  U64 lo = a.bucketStart[b];
  U64 ck[MG_MERGE_PREFETCH]; U32 co[MG_MERGE_PREFETCH], cc[MG_MERGE_PREFETCH];
#pragma unroll
  for (int j = 0 ; j < MG_MERGE_PREFETCH ; ++j)
    { const U32 i = (U32) j * T + tid; ck[j] = 0; co[j] = 0; cc[j] = 0;
      if (i < nu) { ck[j] = __builtin_nontemporal_load (&a.pK[lo + i]); co[j] = __builtin_nontemporal_load (&a.pT[lo + i]); cc[j] = __builtin_nontemporal_load (&a.pC[lo + i]); }
    }
  __syncthreads ();
  for ( ; b < bEnd ; ++b)
    { U32 nnu = 0; U64 nlo = 0; U64 nk[MG_MERGE_PREFETCH]; U32 no[MG_MERGE_PREFETCH], ncc[MG_MERGE_PREFETCH], occNext = 0, nNewNext = 0;
#pragma unroll
      for (int j = 0 ; j < MG_MERGE_PREFETCH ; ++j) { nk[j] = 0; no[j] = 0; ncc[j] = 0; }
      if (b + 1 < bEnd)
        { nnu = a.uniqCount[b + 1]; nlo = a.bucketStart[b + 1]; occNext = a.occ[b + 1];
          nNewNext = a.sliceOff[(U64) (b + 1) * (a.nSlices + 2) + a.nSlices];
#pragma unroll
          for (int j = 0 ; j < MG_MERGE_PREFETCH ; ++j)
            { const U32 i = (U32) j * T + tid;
              if (i < nnu) { nk[j] = __builtin_nontemporal_load (&a.pK[nlo + i]); no[j] = __builtin_nontemporal_load (&a.pT[nlo + i]); ncc[j] = __builtin_nontemporal_load (&a.pC[nlo + i]); }
            }
        }
      if (nu)
        {
          if (occNow)
            { for (U32 i = tid ; i < R ; i += T)
                { uint4 v = *reinterpret_cast<const uint4 *> (&a.slots[(U64) b * R + i]);
                  sKey[i] = ((unsigned long long) v.y << 32) | v.x; sOrd[i] = v.z; sCnt[i] = v.w;
                }
              __syncthreads ();
            }
          /* the rank lookup kernel has turned the new uniques' ordinals into indices: place and count */
          for (U32 i = tid, jj = 0 ; i < nu ; i += T, ++jj)
            { U64 km; U32 ord, c;
              if (jj < MG_MERGE_PREFETCH)
                { km = ck[0]; ord = co[0]; c = cc[0];
#pragma unroll
                  for (int j = 1 ; j < MG_MERGE_PREFETCH ; ++j) if (jj == (U32) j) { km = ck[j]; ord = co[j]; c = cc[j]; }
                }
              else { km = __builtin_nontemporal_load (&a.pK[lo + i]); ord = __builtin_nontemporal_load (&a.pT[lo + i]); c = __builtin_nontemporal_load (&a.pC[lo + i]); }
              if (!a.withDepth) c = 0;
              U32 at;
              if (a.slotShift)                                   /* the slot comes with the entry: every entry its own */
                { at = (U32) (km >> MG_SLOT_SHIFT); km &= ((U64) 1 << MG_SLOT_SHIFT) - 1;
                  if (i < nNew) { sKey[at] = km + 1; sOrd[at] = ord; sCnt[at] = c; }
                  else if (c) sCnt[at] += c;
                  continue;
                }
#ifdef MG_ABLATE
              if (a.debug & 32) { at = mgHomeOfM (km, a.g); sKey[at] = km + 1; } else
#endif
              at = mgLdsClaim (sKey, R, mgHomeOfM (km, a.g), km + 1);      /* km: the mixed k-mer the dedup kernel left */
              if (at == R) { a.counters[1] = 1; continue; }
              if (i < nNew) { sOrd[at] = ord; sCnt[at] = c; }
              else if (c) atomicAdd (&sCnt[at], c);
            }
          __syncthreads ();
          for (U32 i0 = 0 ; i0 < R ; i0 += T)
            { const U32 i = i0 + tid;
              unsigned long long k = 0; U32 dep = 0;
              if (i < R)
                { k = sKey[i];
                  uint4 v; v.x = (U32) k; v.y = (U32) (k >> 32); v.z = sOrd[i]; v.w = sCnt[i];
                  { typedef unsigned v4u __attribute__ ((ext_vector_type (4)));
                    v4u vv = { v.x, v.y, v.z, v.w };
                    if (true MG_ABLATE_AND (!(a.debug & 64)))
                    __builtin_nontemporal_store (vv, reinterpret_cast<v4u *> (&a.slots[(U64) b * R + i]));   /* one 16-byte nt store; the 4.3 GB image is not read again this step: keep it out of the caches the rank records live in (2.58 -> 2.53 ms) */
                  }
                  dep = v.w > 0xffffu ? 0xffffu : v.w;
                  if (k) { sKey[i] = 0; sOrd[i] = 0; sCnt[i] = 0; }
                }
              if (a.liveHist)                                  /* uniform: depths of a set built by this one add */
                { n1 += (U32) __popcll (__ballot (k && dep == 1));
                  n2 += (U32) __popcll (__ballot (k && dep == 2));
                  if (k && dep != 1 && dep != 2)
                    { if (dep < MG_LIVE_BINS) atomicAdd (&sLive[dep], 1u); else atomicAdd (&a.liveHist[dep], 1ull); }
                }
            }
          __syncthreads ();
          if (tid == 0 && nNew) a.occ[b] += nNew;
        }
      nu = nnu; lo = nlo; occNow = occNext; nNew = nNewNext;
#pragma unroll
      for (int j = 0 ; j < MG_MERGE_PREFETCH ; ++j) { ck[j] = nk[j]; co[j] = no[j]; cc[j] = ncc[j]; }
    }
  if (a.liveHist)
    { __syncthreads ();
      if ((tid & 63) == 0) { if (n1) atomicAdd (&sLive[1], n1); if (n2) atomicAdd (&sLive[2], n2); }
      __syncthreads ();
      for (U32 i = tid ; i < MG_LIVE_BINS ; i += T) if (sLive[i]) atomicAdd (&a.liveHist[i], (unsigned long long) sLive[i]);
    }
}

/* step 3b: ordinal of the first occurrence -> index, for every new unique of every bucket's list, slice by slice.
 * A wave takes 64 consecutive buckets' group s (one bucket per lane for the bounds, then list by list).  Workgroups
 * b, b + 8, b + 16 ... are placed on one XCD (observed, MI355X_MICROARCH.md; it only matters for speed), and they
 * take the slices x, x + 8, ... one after the other, so an XCD works on one slice at a time and reads its rank records
 * (2^sliceShift / 4 bytes) from its own L2 instead of once per k-mer from the fabric (tools/ubench_rand: 240 G/s
 * against 66 G/s at the whole structure's 39 MB). */
#define MG_LOOKUP_UNROLL 4
__global__ __launch_bounds__ (256)
void mgRankLookupKernel (const MgBucketArgs a, U32 groupsPerSlice)
{
  MG_BUILD_PRIO ();
  const int lane = threadIdx.x & 63;
  const U32 blocksPerSlice = (groupsPerSlice + 3) / 4;
  const U32 xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
  const U32 s = xcd + 8 * (k / blocksPerSlice), g = (k % blocksPerSlice) * 4 + (threadIdx.x >> 6);
  if (s >= a.nSlices || g >= groupsPerSlice) return;
  const U32 b = g * 64 + (U32) lane;
  U64 lo = 0; U32 n = 0;
  if (b < a.nBuckets)
    { const unsigned short *o = a.sliceOff + (U64) b * (a.nSlices + 2) + s;
      const U32 o0 = o[0], o1 = o[1];
      n = o1 - o0; lo = a.bucketStart[b] + o0;
    }
  auto indexOf = [&] (U32 tok, const uint4 &r) -> U32
    { const U64 bits = ((U64) r.y << 32) | r.x;
      const U64 idx = (U64) a.baseMax + 1 + r.z + (U32) __popcll (bits & (((U64) 1 << (tok & 63)) - 1));
      return idx < a.size ? ((U32) idx | MG_ASSIGNED) : 0;
    };
  /* MG_LOOKUP_UNROLL lists per round, all their loads in flight together.  (Fetching the next round's ordinals under
     this round's gathers, or 8 lists per round, changes nothing; nor does using every lane of every gather -- round 6: a wave's 64
     lists flattened, 64 items per instruction instead of a list's 42, 0.935 against 0.897 ms, DESIGN_EXPERIMENTS.md §K: what counts is
     the number of scattered LANES the memory path takes, 115 G/s here, not the number of instructions.) */
  for (int j = 0 ; j < 64 ; j += MG_LOOKUP_UNROLL)
    { U64 l[MG_LOOKUP_UNROLL]; U32 m[MG_LOOKUP_UNROLL], tok[MG_LOOKUP_UNROLL]; uint4 r[MG_LOOKUP_UNROLL];
#pragma unroll
      for (int u = 0 ; u < MG_LOOKUP_UNROLL ; ++u)
        { l[u] = ((U64) (U32) __shfl ((int) (U32) (lo >> 32), j + u) << 32) | (U32) __shfl ((int) (U32) lo, j + u);
          m[u] = (U32) __shfl ((int) n, j + u);
        }
#pragma unroll
      for (int u = 0 ; u < MG_LOOKUP_UNROLL ; ++u)
        { tok[u] = 0; if ((U32) lane < m[u] MG_ABLATE_AND (!(a.debug & 256))) tok[u] = 0x7fffffffu - __builtin_nontemporal_load (&a.pT[l[u] + lane]); }
#pragma unroll
      for (int u = 0 ; u < MG_LOOKUP_UNROLL ; ++u)
        { r[u] = make_uint4 (0, 0, 0, 0); if ((U32) lane < m[u] MG_ABLATE_AND (!(a.debug & 16))) r[u] = *reinterpret_cast<const uint4 *> (&a.grp[tok[u] >> 6]); }
#pragma unroll
      for (int u = 0 ; u < MG_LOOKUP_UNROLL ; ++u) asm volatile ("" : "+v" (r[u].x), "+v" (r[u].y), "+v" (r[u].z));   /* all records in before the first store (see mgRankAssignKernel) */
#pragma unroll
      for (int u = 0 ; u < MG_LOOKUP_UNROLL ; ++u)
        if ((U32) lane < m[u] MG_ABLATE_AND (!(a.debug & 128))) a.pT[l[u] + lane] = indexOf (tok[u], r[u]);
#pragma unroll
      for (int u = 0 ; u < MG_LOOKUP_UNROLL ; ++u)
        for (U32 i = 64 + (U32) lane ; i < m[u] ; i += 64)                /* lists longer than a wave: rare at the usual slice size */
          { const U32 t = 0x7fffffffu - __builtin_nontemporal_load (&a.pT[l[u] + i]);
            const uint4 rr = *reinterpret_cast<const uint4 *> (&a.grp[t >> 6]);
            a.pT[l[u] + i] = indexOf (t, rr);
          }
    }
}

/* ======================================================================================== */
/* Partitioned lookups (the modmap query path on a table far larger than the caches, round 4).  A lookup in ordinal order
 * is one random 64-byte line out of a 2 GB table: 35 G lookups/s, the chip's rate for that footprint.  Here the batch's
 * modimizers go through the FIRST partition pass of the build (by the top bits of the mixed k-mer, straight from the scan's
 * segments, the digit counts made by the scan); a bin's lookups then touch one contiguous piece of the table -- 1 / nBins
 * of it -- and the workgroups of one XCD take one bin at a time, so the piece is read from that XCD's L2; every element
 * becomes (ordinal << 32 | index) in place; and because the scatter kernel noted where each sub-chunk's run of every bin
 * went (runTab), a workgroup per sub-chunk pulls its 16384 ordinals' results back from the nBins runs into an LDS tile and
 * writes them out in order: no second sort, no random store.  Semantics: modsetIndexFind (ms, kmer, false), modset.c:45-62. */
__global__ __launch_bounds__ (256)
void mgBinFindKernel (const MgSlot *__restrict__ slots, MgGeom g, MgPartFmt f, U64 *__restrict__ el, const U64 *__restrict__ binStart,
                      U32 nBins, U32 wgPerXcd)
{
  const U32 xcd = blockIdx.x & 7, j = blockIdx.x >> 3;      /* workgroups b, b + 8, ... share an XCD (observed placement; speed only) */
  const U64 ordMask = ((U64) 1 << f.ordBits) - 1;
  for (U32 b = xcd ; b < nBins ; b += 8)
    { const U64 lo = binStart[b], hi = binStart[b + 1];
      for (U64 i = lo + (U64) j * 256 + threadIdx.x ; i < hi ; i += (U64) wgPerXcd * 256)      /* (four lookups in flight per thread: 3.4 against 2.8 ms) */
        { const U64 x = __builtin_nontemporal_load (&el[i]);
          const U64 m = ((U64) b << f.remBits) | (x >> f.ordBits), key = m + 1;
          const U64 base = (U64) mgBucketOfM (m, g) * g.R;
          U32 slot = mgHomeOfM (m, g), res = 0;
          for (U32 probes = 0 ; probes < g.R ; ++probes)
            { const uint4 w = *reinterpret_cast<const uint4 *> (&slots[base + slot]);
              const U64 cur64 = ((U64) w.y << 32) | w.x;
              if (cur64 == key) { res = mgIsAssigned (w.z) ? (w.z & ~MG_ASSIGNED) : 0; break; }
              if (cur64 == 0) break;
              slot = mgNextSlot (slot, g.R);
            }
          __builtin_nontemporal_store (((x & ordMask) << 32) | res, &el[i]);
        }
    }
}

template <int SUB>
__global__ __launch_bounds__ (1024)
void mgUnpartKernel (const U64 *__restrict__ el, const unsigned long long *__restrict__ runTab, U32 nBins, U64 n, U32 *__restrict__ out)
{
  __shared__ U32 sTile[SUB];
  __shared__ unsigned long long sRun[MG_PART_MAXBINS];
  const U32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const U64 nSub = (n + SUB - 1) / SUB;
  for (U64 s = blockIdx.x ; s < nSub ; s += gridDim.x)
    { const U64 o0 = s * SUB;
      const U32 cnt = (U32) (o0 + SUB < n ? SUB : n - o0);
      for (U32 b = tid ; b < nBins ; b += 1024) sRun[b] = runTab[s * nBins + b];
      __syncthreads ();
      for (U32 b = wave ; b < nBins ; b += 16)
        { const unsigned long long r = sRun[b];
          const U64 base = r & (((U64) 1 << 40) - 1); const U32 len = (U32) (r >> 40);
          for (U32 i = lane ; i < len ; i += 64)
            { const U64 v = __builtin_nontemporal_load (&el[base + i]);
              sTile[(U32) ((v >> 32) - o0)] = (U32) v;
            }
        }
      __syncthreads ();
      for (U32 i = tid ; i < cnt ; i += 1024) __builtin_nontemporal_store (sTile[i], &out[o0 + i]);
      __syncthreads ();
    }
}

/* ---- two levels (MODGPU_FIND_PATH=2): the second partition pass of the build as well, so that a bucket's lookups are contiguous
 * and the bucket (64 KiB) sits in LDS while they run.  The second pass replaces every element's ordinal by its position in the
 * first pass's output, where the element itself (with its ordinal) stays; results come back in two pulls: (chunk, half) tiles of
 * the first pass's output from the second pass's runs, then sub-chunks of ordinals from the first pass's runs. ---- */
__global__ __launch_bounds__ (1024)
void mgBucketFindKernel (const MgSlot *__restrict__ slots, const U32 *__restrict__ occ, MgGeom g, MgPartFmt f, U32 nBuckets,
                         const U64 *__restrict__ bucketStart, U64 *__restrict__ el, U32 bucketsPerBlock)
{
  const U32 R = g.R, T = blockDim.x, tid = threadIdx.x;
  unsigned long long *sKey = reinterpret_cast<unsigned long long *> (mgDynLds);
  U32 *sIdx = reinterpret_cast<U32 *> (mgDynLds + (size_t) R * 8);
  const U64 posMask = ((U64) 1 << f.ordBits) - 1;
  U32 b = blockIdx.x * bucketsPerBlock, bEnd = b + bucketsPerBlock;
  if (bEnd > nBuckets) bEnd = nBuckets;
  for ( ; b < bEnd ; ++b)
    { const U64 lo = bucketStart[b], hi = bucketStart[b + 1];
      if (hi == lo) continue;                                /* (uniform) */
      const bool any = occ[b] != 0;
      if (any)
        { for (U32 i = tid ; i < R ; i += T)
            { const uint4 v = *reinterpret_cast<const uint4 *> (&slots[(U64) b * R + i]);
              sKey[i] = ((unsigned long long) v.y << 32) | v.x; sIdx[i] = mgIsAssigned (v.z) ? (v.z & ~MG_ASSIGNED) : 0;
            }
          __syncthreads ();
        }
      for (U64 i = lo + tid ; i < hi ; i += T)
        { const U64 x = __builtin_nontemporal_load (&el[i]);
          U32 res = 0;
          if (any)
            { const U64 m = ((U64) (b >> f.loB) << f.remBits) | (x >> f.ordBits), key = m + 1;
              U32 slot = mgHomeOfM (m, g);
              for (U32 probes = 0 ; probes < R ; ++probes)
                { const unsigned long long cur = sKey[slot];
                  if (cur == key) { res = sIdx[slot]; break; }
                  if (cur == 0) break;
                  slot = mgNextSlot (slot, g.R);
                }
            }
          __builtin_nontemporal_store (((x & posMask) << 32) | res, &el[i]);
        }
      __syncthreads ();                                      /* the image is loaded again for the next bucket */
    }
}

/* The same out of an 8-byte-per-slot copy of the table (round 6).  The two-level lookups STREAM the table once per batch -- 16 bytes a slot of which
 * a lookup needs the key and the index; a bucket implies the key's leading bits, so where 2k - log2 NB <= 32 both fit one word:
 * (key's bits below the bucket id + 1) << 31 | index, 0 = empty.  The copy (MgTable.find8) is made by mgTablePack8Kernel when a lookup batch finds
 * it missing or older than the table (one streaming pass), and a batch then reads half the bytes: config 3's table 1.9 -> 0.95 GB per batch. */
__global__ __launch_bounds__ (256)
void mgTablePack8Kernel (const MgSlot *__restrict__ slots, const U32 *__restrict__ occ, U32 nBuckets, U32 R, int remB, U64 *__restrict__ out)
{
  const U64 remMask = ((U64) 1 << remB) - 1;
  for (U32 b = blockIdx.x ; b < nBuckets ; b += gridDim.x)
    { if (!occ[b]) continue;                                /* (never read: the lookups ask occ[] first) */
      for (U32 i = threadIdx.x ; i < R ; i += blockDim.x)
        { const uint4 v = *reinterpret_cast<const uint4 *> (&slots[(U64) b * R + i]);
          const U64 key = ((U64) v.y << 32) | v.x;
          U64 w = 0;
          if (key && mgIsAssigned (v.z)) w = ((((key - 1) & remMask) + 1) << 31) | (U64) (v.z & ~MG_ASSIGNED);
          __builtin_nontemporal_store (w, &out[(U64) b * R + i]);
        }
    }
}

__global__ __launch_bounds__ (1024)
void mgBucketFind8Kernel (const U64 *__restrict__ find8, const U32 *__restrict__ occ, MgGeom g, MgPartFmt f, U32 nBuckets, int remB,
                          const U64 *__restrict__ bucketStart, U64 *__restrict__ el, U32 bucketsPerBlock)
{
  const U32 R = g.R, T = blockDim.x, tid = threadIdx.x;
  unsigned long long *sW = reinterpret_cast<unsigned long long *> (mgDynLds);
  const U64 posMask = ((U64) 1 << f.ordBits) - 1, remMask = ((U64) 1 << remB) - 1;
  U32 b = blockIdx.x * bucketsPerBlock, bEnd = b + bucketsPerBlock;
  if (bEnd > nBuckets) bEnd = nBuckets;
  for ( ; b < bEnd ; ++b)
    { const U64 lo = bucketStart[b], hi = bucketStart[b + 1];
      if (hi == lo) continue;                                /* (uniform) */
      const bool any = occ[b] != 0;
      if (any)
        { for (U32 i = tid ; i < R ; i += T) sW[i] = __builtin_nontemporal_load (&find8[(U64) b * R + i]);
          __syncthreads ();
        }
      for (U64 i = lo + tid ; i < hi ; i += T)
        { const U64 x = __builtin_nontemporal_load (&el[i]);
          U32 res = 0;
          if (any)
            { const U64 m = ((U64) (b >> f.loB) << f.remBits) | (x >> f.ordBits), want = (m & remMask) + 1;
              U32 slot = mgHomeOfM (m, g);
              for (U32 probes = 0 ; probes < R ; ++probes)
                { const unsigned long long cur = sW[slot];
                  if ((cur >> 31) == want) { res = (U32) cur & 0x7fffffffu; break; }
                  if (cur == 0) break;
                  slot = mgNextSlot (slot, g.R);
                }
            }
          __builtin_nontemporal_store (((x & posMask) << 32) | res, &el[i]);
        }
      __syncthreads ();                                      /* the image is loaded again for the next bucket */
    }
}

/* the second level's results back into the order of the first pass's output: a workgroup per (chunk, half) of that output */
template <int SUB>
__global__ __launch_bounds__ (1024)
void mgUnpartPosKernel (const U64 *__restrict__ el, const unsigned long long *__restrict__ runTab, U32 nBins,
                        const U64 *__restrict__ segStart, const U32 *__restrict__ chunkBase, U32 nSeg, U32 chunkElems, U32 *__restrict__ idxOut)
{
  __shared__ U32 sTile[SUB];
  __shared__ unsigned long long sRun[MG_PART_MAXBINS];
  const U32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const U32 nChunks = chunkBase[nSeg];
  for (U32 w = blockIdx.x ; w < 2 * nChunks ; w += gridDim.x)
    { const U32 c = w >> 1, h = w & 1;
      U32 seg; U64 lo, hi;
      if (!mgChunkRange (segStart, chunkBase, nSeg, chunkElems, c, &seg, &lo, &hi)) continue;
      const U64 sub = lo + (U64) h * SUB;
      if (sub >= hi) continue;                               /* (uniform) */
      const U32 cnt = (U32) (sub + SUB < hi ? SUB : hi - sub);
      for (U32 b = tid ; b < nBins ; b += 1024) sRun[b] = runTab[(U64) w * nBins + b];
      __syncthreads ();
      for (U32 b = wave ; b < nBins ; b += 16)
        { const unsigned long long r = sRun[b];
          const U64 base = r & (((U64) 1 << 40) - 1); const U32 len = (U32) (r >> 40);
          for (U32 i = lane ; i < len ; i += 64)
            { const U64 v = __builtin_nontemporal_load (&el[base + i]);
              sTile[(U32) ((v >> 32) - sub)] = (U32) v;
            }
        }
      __syncthreads ();
      for (U32 i = tid ; i < cnt ; i += 1024) idxOut[sub + i] = sTile[i];
      __syncthreads ();
    }
}

/* the first level's pull when the results sit beside the elements (idx[]) and the elements still hold their ordinals */
template <int SUB>
__global__ __launch_bounds__ (1024)
void mgUnpartOrdKernel (const U64 *__restrict__ el, const U32 *__restrict__ idx, int ordBits, const unsigned long long *__restrict__ runTab, U32 nBins, U64 n,
                        U32 *__restrict__ out)
{
  __shared__ U32 sTile[SUB];
  __shared__ unsigned long long sRun[MG_PART_MAXBINS];
  const U32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const U64 nSub = (n + SUB - 1) / SUB, ordMask = ((U64) 1 << ordBits) - 1;
  for (U64 s = blockIdx.x ; s < nSub ; s += gridDim.x)
    { const U64 o0 = s * SUB;
      const U32 cnt = (U32) (o0 + SUB < n ? SUB : n - o0);
      for (U32 b = tid ; b < nBins ; b += 1024) sRun[b] = runTab[s * nBins + b];
      __syncthreads ();
      for (U32 b = wave ; b < nBins ; b += 16)
        { const unsigned long long r = sRun[b];
          const U64 base = r & (((U64) 1 << 40) - 1); const U32 len = (U32) (r >> 40);
          for (U32 i = lane ; i < len ; i += 64)
            sTile[(U32) ((__builtin_nontemporal_load (&el[base + i]) & ordMask) - o0)] = __builtin_nontemporal_load (&idx[base + i]);
        }
      __syncthreads ();
      for (U32 i = tid ; i < cnt ; i += 1024) __builtin_nontemporal_store (sTile[i], &out[o0 + i]);
      __syncthreads ();
    }
}

/* ======================================================================================== */
/* host side                                                                                  */

static inline unsigned mgGrid (U64 n, unsigned per = 256, unsigned cap = 16384)
{ U64 b = (n + per - 1) / per; if (b > cap) b = cap; if (b < 1) b = 1; return (unsigned) b; }
static inline size_t mgAl (size_t n) { return (n + 255) & ~(size_t) 255; }
static inline int mgLog2 (U64 x) { int l = 0; while (((U64) 1 << l) < x) ++l; return l; }
static inline MgGeom mgGeomOf (const MgTable *t) { MgGeom g; g.R = t->R; g.log2NB = t->log2NB; g.kbits = t->kbits; return g; }

#define MG_RANK_UNITS 8192           /* waves that share the ordered flag count */
static inline U64 mgRankRowsPerUnit (U64 n, U32 *nBlocks)
{
  U64 nRows = (n + 63) / 64;
  U64 per = (nRows + MG_RANK_UNITS - 1) / MG_RANK_UNITS; if (!per) per = 1;
  U64 units = (nRows + per - 1) / per; if (!units) units = 1;
  *nBlocks = (U32) ((units + 3) / 4);
  return per;
}

/* the split of oversize buckets: occurrences above which a bucket is split, occurrences of a chunk at least */
static void mgHotKnobs (U32 *split, U32 *chunk)
{
  const MgKnobs *k = mgKnobs ();                             /* test knob "split,chunk": small values send ordinary buckets through the split path */
  U32 a = MG_HOT_SPLIT_DEFAULT, b = MG_HOT_CHUNK_DEFAULT;
  if (k->hotSplit != MG_KNOB_UNSET && k->hotSplit > 0)
    { a = (U32) k->hotSplit; b = k->hotChunk != MG_KNOB_UNSET && k->hotChunk > 0 ? (U32) k->hotChunk : a / 4; }
  if (b < 64) b = 64;
  if (a < b) a = b;
  *split = a; *chunk = b;
}
static U64 mgHotItemsCap (U64 n) { U32 sp, ch; mgHotKnobs (&sp, &ch); return n / ch + n / sp + 16; }

/* scratch needed by mgTableAdd for a batch of n */
size_t mgTableAddScratchBytes (const MgTable *t, U64 n)
{
  (void) t;
  U64 NB = (U64) 1 << 18;               /* the largest bucket count: the table may grow between passes of one call */
  size_t rank = mgAl (n) /*flags*/ + 2 * mgAl ((MG_RANK_UNITS + 8) * 8) + mgAl ((n / 64 + 2) * sizeof (MgRankGrp));
  size_t direct = mgAl (n * 4);
  size_t part = 2 * (mgAl (n * 8) + mgAl (n * 4)) + mgAl (n * 4)
              + mgAl ((NB + 2) * 8) * 3 + mgAl ((NB + 2) * 4) * 2 + mgAl (((U64) MG_PART_MAXBINS + 2) * 8) * 3 + 2 * mgAl ((U64) MG_PART_MAXBINS * 16 * 8 + 4096)
              + mgAl ((MG_PART_MAXBINS + 2) * 4) + mgAl ((MG_PART_MAXBINS + 2 + n / MG_PART_CHUNK + MG_PART_MAXBINS + 2) * 4)
              + mgAl ((size_t) (MG_RANK_GROUPS + 2) * NB * sizeof (unsigned short)) + mgAl ((n / MG_PART_SUB + 2) * 24)
              + mgAl (mgHotItemsCap (n) * 8) + mgAl (mgHotItemsCap (n) * 4) + 256;
  return rank + (direct > part ? direct : part) + 4096;
}

static int mgPathOverride (void)
{
  const long c = mgKnobs ()->tablePath;                       /* test knob: the first letter of "direct" / "bucket" */
  return c == 'd' ? 1 : (c == 'b' ? 2 : 0);
}

bool mgTableUseBuckets (const MgTable *t, U64 n);
/* can mgTableAdd read this batch from the scan's segments?  Only the bucketed path does, and its first pass then
   needs the digit counts the scan made for exactly this table geometry */
bool mgTableAddTakesSegments (const MgTable *t, U64 n, const MgHistReq *counted)
{
  const int off = mgKnobs ()->noSegmentInput == 1;   /* test knob: always compact first */
  return !off && n && mgTableUseBuckets (t, n) && counted && counted->binCount && counted->log2NB == t->log2NB && counted->kbits == t->kbits && !counted->hiB;
}

bool mgTableUseBuckets (const MgTable *t, U64 n)
{
  int ov = mgPathOverride ();
  if (ov == 1) return false;
  if (ov == 2) return true;
  /* a batch of a million or so: the bucketed path's dozen launches and its per-bucket lists cost 0.43 ms whatever the size, the atomics
     0.1 ms + 0.19 ms per million (tools/size_sweep_probe.py: 0.78 M modimizers 0.30 against 0.43 ms, 3.1 M 0.68 against 0.45) */
  if (n < 1500000) return false;
  return n >= t->nSlots / 16;           /* streaming every touched bucket twice beats ~100 ps/modimizer of atomics */
}

/* one partition pass: nSeg segments of kIn -> nBins bins each.  inMode: what kIn holds (MG_EL_*); packed: the format
 * of the output (and, after the first pass, of the input) */
static MgStatus mgPartPass (const MgTable *t, int inMode, bool packed, const MgPartFmt &f, const U64 *kIn, const U32 *tIn, U64 n,
                            const U64 *segStart, U32 nSeg, int shift, U32 nBins,
                            U64 *kOut, U32 *tOut, U64 *binStart, unsigned long long *cursor, U32 *binCount, U32 *chunkBase,
                            hipStream_t st, const U32 *counted = 0, const MgSegSrc *segSrc = 0, MgSubSeg *subSeg = 0,
                            unsigned long long *runTab = 0, U32 *subElems = 0, int runMode = 0, U32 *maxChunksOut = 0,
                            unsigned char *digitOut = 0, int nextShift = 0, U32 nextBins = 0, const unsigned char *digitIn = 0)
{
  /* digitOut: this pass also writes the NEXT pass's digit (bits [nextShift, ..) of the bucket id, nextBins <= 256 of them) of every element
     beside it; digitIn: this pass counts its digits from such bytes */
  MgGeom g = mgGeomOf (t);
  MgSegSrc src; src.segKmer = 0; src.segCount = 0; src.segStart = 0; src.segCap = 0; src.nSegs = 0;
  if (inMode == MG_EL_SEG)
    { if (!segSrc || !subSeg || !counted) { mgSetError ("internal: segment input needs its counts"); return MG_ERR_ARG; }
      src = *segSrc;
      const U64 nSub = (n + MG_PART_SUB - 1) / MG_PART_SUB;
      MG_LAUNCH (MG_K_PART, st, mgSubSegKernel, dim3 ((unsigned) ((nSub + 255) / 256)), dim3 (256), 0, st, src, n, (U32) MG_PART_SUB, subSeg);
    }
  if (counted) MG_HIP (hipMemcpy2DAsync (binCount, sizeof (U32), counted, MG_HIST_STRIDE * sizeof (U32), sizeof (U32), nBins, hipMemcpyDeviceToDevice, st));   /* the scan counted them */
  else MG_HIP (hipMemsetAsync (binCount, 0, (size_t) nSeg * nBins * sizeof (U32), st));
  /* large sub-chunks where the kernel has them: packed elements, at most 256 bins */
  const int bigEnv = mgKnobs ()->partBig == MG_KNOB_UNSET ? 1 : (int) mgKnobs ()->partBig;      /* test knob: 0 = sub-chunks of MG_PART_SUB everywhere */
  const bool big = bigEnv && packed && nBins <= MG_PART_BIG_BINS && MG_PART_THREADS == 1024;
  const U32 chunkElems = 2u * (U32) (big ? MG_PART_SUB_BIG : MG_PART_SUB);
  if (subElems) *subElems = (U32) (big ? MG_PART_SUB_BIG : MG_PART_SUB);
  MG_LAUNCH (MG_K_PART, st, mgPartChunksKernel, dim3 (1), dim3 (MG_PART_MAXBINS), 0, st, segStart, nSeg, chunkElems, chunkBase);
  unsigned maxChunks = (unsigned) (n / chunkElems + nSeg + 1);
  if (maxChunksOut) *maxChunksOut = maxChunks;
  const int sgEnv = mgKnobs ()->scatterGrid == MG_KNOB_UNSET ? 0 : (int) mgKnobs ()->scatterGrid;   /* dev knob */
  unsigned scatterGrid = maxChunks < (unsigned) (sgEnv > 0 ? sgEnv : 1024) ? maxChunks : (unsigned) (sgEnv > 0 ? sgEnv : 1024);
  const dim3 hg (maxChunks < 4096 ? maxChunks : 4096), sg (scatterGrid);
  if (!counted && digitIn)
    MG_LAUNCH (MG_K_PART_HIST, st, mgPartHistBytesKernel, hg, dim3 (256), 0, st, digitIn, nBins, segStart, chunkBase, nSeg, chunkElems, binCount);
  else if (!counted)
    { if (inMode == MG_EL_DENSE)
        MG_LAUNCH (MG_K_PART_HIST, st, mgPartHistKernel<MG_EL_DENSE>, hg, dim3 (256), 0, st, kIn, g, f, shift, nBins, segStart, chunkBase, nSeg, chunkElems, binCount);
      else if (inMode == MG_EL_WIDE)
        MG_LAUNCH (MG_K_PART_HIST, st, mgPartHistKernel<MG_EL_WIDE>, hg, dim3 (256), 0, st, kIn, g, f, shift, nBins, segStart, chunkBase, nSeg, chunkElems, binCount);
      else
        MG_LAUNCH (MG_K_PART_HIST, st, mgPartHistKernel<MG_EL_PACKED>, hg, dim3 (256), 0, st, kIn, g, f, shift, nBins, segStart, chunkBase, nSeg, chunkElems, binCount);
    }
  /* one segment (the first pass): every workgroup reserves in the same few hundred cursors -- one cache line each
     (scatter 0.88 -> 0.73 ms: returning atomics on cursors that share a line queue behind each other) */
  const U32 cstride = nSeg == 1 ? 16u : 1u;                  /* (the second pass's 65536 cursors: no gain from padding) */
  MG_LAUNCH (MG_K_PART, st, mgPartScanKernel, dim3 (nSeg), dim3 (MG_PART_MAXBINS), 0, st, binCount, nBins, segStart, binStart, cursor, cstride, nSeg, n);
#define MG_SCATTER(IN, PK, SUB) MG_LAUNCH (MG_K_PART_SCATTER, st, (mgPartScatterKernel<IN, PK, SUB>), sg, dim3 (MG_PART_THREADS), 0, st, \
                                           kIn, tIn, src, subSeg, g, f, shift, nBins, segStart, chunkBase, nSeg, chunkElems, cursor, cstride, kOut, tOut, runTab, runMode, \
                                           digitOut, nextShift, nextBins ? nextBins - 1 : 0u)
#define MG_SCATTER_P(IN) do { if (big) MG_SCATTER (IN, true, MG_PART_SUB_BIG); else MG_SCATTER (IN, true, MG_PART_SUB); } while (0)
  if (inMode == MG_EL_DENSE) { if (packed) MG_SCATTER_P (MG_EL_DENSE); else MG_SCATTER (MG_EL_DENSE, false, MG_PART_SUB); }
  else if (inMode == MG_EL_SEG) { if (packed) MG_SCATTER_P (MG_EL_SEG); else MG_SCATTER (MG_EL_SEG, false, MG_PART_SUB); }
  else if (inMode == MG_EL_WIDE) MG_SCATTER (MG_EL_WIDE, false, MG_PART_SUB);
  else MG_SCATTER_P (MG_EL_PACKED);
#undef MG_SCATTER_P
#undef MG_SCATTER
  MG_HIP (hipGetLastError ());
  return MG_OK;
}

/* the dedup kernel's per-bucket counts of distinct k-mers: out[0] = their sum, out[1] = the largest (out[] zeroed by the launcher) */
__global__ __launch_bounds__ (256)
void mgUniqStatsKernel (const U32 *__restrict__ uniqCount, U32 nBuckets, unsigned long long *__restrict__ out)
{
  unsigned long long sum = 0; U32 mx = 0;
  for (U32 b = blockIdx.x * 256 + threadIdx.x ; b < nBuckets ; b += gridDim.x * 256) { const U32 c = uniqCount[b]; sum += c; mx = c > mx ? c : mx; }
  for (int off = 32 ; off ; off >>= 1)
    { sum += ((unsigned long long) (U32) __shfl_xor ((int) (U32) (sum >> 32), off) << 32) | (U32) __shfl_xor ((int) (U32) sum, off);
      const U32 o = (U32) __shfl_xor ((int) mx, off); mx = o > mx ? o : mx;
    }
  if ((threadIdx.x & 63) == 0) { if (sum) atomicAdd (&out[0], sum); if (mx) atomicMax (&out[1], (unsigned long long) mx); }
}

#define MG_TIGHT_PCT_DEFAULT 50      /* see MgTable.tightPct */
#define MG_TIGHT_MIN_R 1024u         /* a bucket keeps room for the spread of a later add's share around its mean (mgTableEnsure sizes by the mean) */

/* insert a batch (ordinal order = array order); counters[0] = number of new entries afterwards */
MgStatus mgTableAdd (MgTable *t, const U64 *dKmer, U64 n, int withDepth, void *scratch, hipStream_t st,
                     const MgHistReq *counted, const MgSegSrc *segSrc)
{
  if (!n) return MG_OK;
  if (segSrc && !mgTableAddTakesSegments (t, n, counted)) { mgSetError ("internal: this insert needs the dense k-mers"); return MG_ERR_ARG; }
  MgSegSrc noSrc; noSrc.segKmer = 0; noSrc.segCount = 0; noSrc.segStart = 0; noSrc.segCap = 0; noSrc.nSegs = 0;
  t->liveHistValid = false;                          /* set again below if this add is the set's only one */
  if (withDepth) t->pendingDepth = true;
  const bool wasEmpty = t->empty && t->max == 0;
  t->empty = false; ++t->version;
  MgGeom g = mgGeomOf (t);
  char *wb = (char *) scratch;
  unsigned char *flags = (unsigned char *) wb;       wb += mgAl (n);
  U64 *blockCount = (U64 *) wb;                      wb += mgAl ((MG_RANK_UNITS + 8) * 8);
  U64 *blockBase = (U64 *) wb;                       wb += mgAl ((MG_RANK_UNITS + 8) * 8);
  MgRankGrp *grp = (MgRankGrp *) wb;                 wb += mgAl ((n / 64 + 2) * sizeof (MgRankGrp));
  U32 nRankBlocks; U64 rankTiles = mgRankRowsPerUnit (n, &nRankBlocks);
  MG_HIP (hipMemsetAsync (blockCount, 0, (MG_RANK_UNITS + 8) * 8, st));

  if (!mgTableUseBuckets (t, n))
    { U32 *slotId = (U32 *) wb;
      { MgStatus cs = mgTableClean (t, st); if (cs) return cs; }
      MG_LAUNCH (MG_K_TABLE_INSERT, st, mgTableInsertKernel, dim3 (mgGrid (n)), dim3 (256), 0, st, t->slots, g, dKmer, n, slotId, withDepth, t->counters);
      MG_LAUNCH (MG_K_TABLE_FLAG, st, mgDirectFlagKernel, dim3 (mgGrid (n)), dim3 (256), 0, st, t->slots, slotId, n, flags);
      MG_LAUNCH (MG_K_RANK_COUNT, st, mgRankCountKernel, dim3 (nRankBlocks), dim3 (256), 0, st, flags, n, rankTiles, blockCount);
      MG_LAUNCH (MG_K_RANK_SCAN, st, mgRankScanKernel, dim3 (1), dim3 (1024), 0, st, blockCount, nRankBlocks * 4, blockBase, t->counters);
      MG_LAUNCH (MG_K_TABLE_ASSIGN, st, (mgRankAssignKernel<true, false>), dim3 (nRankBlocks), dim3 (256), 0, st,
                 flags, dKmer, noSrc, (const MgSubSeg *) 0, n, rankTiles, blockBase, t->max, t->size, t->value, t->slots, slotId, grp);
      MG_HIP (hipGetLastError ());
      /* occ[] is kept exact only by the bucketed path and the loader; the direct path marks buckets non-empty */
      return mgTableMarkOccupied (t, dKmer, n, st);
    }

  /* ---- bucketed ---- */
  const U64 NB = (U64) 1 << t->log2NB;
  U64 *kA = (U64 *) wb;  wb += mgAl (n * 8);
  U32 *tA = (U32 *) wb;  wb += mgAl (n * 4);
  U64 *kB = (U64 *) wb;  wb += mgAl (n * 8);
  U32 *tB = (U32 *) wb;  wb += mgAl (n * 4);
  U32 *cB = (U32 *) wb;  wb += mgAl (n * 4);
  U64 *fineStart = (U64 *) wb;                wb += mgAl ((NB + 2) * 8);
  unsigned long long *fineCursor = (unsigned long long *) wb; wb += mgAl ((NB + 2 + (U64) MG_PART_MAXBINS * 16) * 8);
  U64 *spare64 = (U64 *) wb;                  wb += mgAl ((NB + 2) * 8);
  U32 *fineCount = (U32 *) wb;                wb += mgAl ((NB + 2) * 4);
  U32 *uniqCount = (U32 *) wb;                wb += mgAl ((NB + 2) * 4);
  U64 *coarseStart = (U64 *) wb;              wb += mgAl (((U64) MG_PART_MAXBINS + 2) * 8);
  unsigned long long *coarseCursor = (unsigned long long *) wb; wb += mgAl (((U64) MG_PART_MAXBINS + 2) * 8 * 16);
  U64 *whole = (U64 *) wb;                    wb += mgAl (((U64) MG_PART_MAXBINS + 2) * 8);
  U32 *coarseCount = (U32 *) wb;              wb += mgAl ((MG_PART_MAXBINS + 2) * 4);
  U32 *chunkBase = (U32 *) wb;                wb += mgAl ((MG_PART_MAXBINS + 2 + n / MG_PART_CHUNK + MG_PART_MAXBINS + 2) * 4);   /* + the segment of every chunk */
  unsigned short *sliceOff = (unsigned short *) wb;   wb += mgAl ((size_t) (MG_RANK_GROUPS + 2) * NB * sizeof (unsigned short));
  MgSubSeg *subSeg = (MgSubSeg *) wb;         wb += mgAl ((n / MG_PART_SUB + 2) * sizeof (MgSubSeg));
  U64 *hotItems = (U64 *) wb;                 wb += mgAl (mgHotItemsCap (n) * 8);
  U32 *hotBuckets = (U32 *) wb;               wb += mgAl (mgHotItemsCap (n) * 4);
  unsigned long long *hotCount = (unsigned long long *) wb; wb += 256;
  (void) spare64;

  /* split the bucket-id bits into a coarse digit (high) and a fine digit (low), each <= 9 bits */
  const int j = t->log2NB;
  int hiB, loB;
  mgPartSplit (j, &hiB, &loB);
  const U32 *pre = (counted && counted->binCount && counted->log2NB == j && counted->kbits == t->kbits && !counted->hiB) ? counted->binCount : 0;
  U64 segInit[2] = { 0, n };
  MG_HIP (hipMemcpyAsync (whole, segInit, 16, hipMemcpyHostToDevice, st));
  /* element format: one packed 8-byte word when the mixed k-mer without its coarse digit and the ordinal fit in 64 bits */
  MgPartFmt f;
  f.ordBits = mgLog2 (n) > 1 ? mgLog2 (n) : 1; f.remBits = t->kbits - hiB; f.loB = loB;
  const int packEnv = mgKnobs ()->partPacked == MG_KNOB_UNSET ? 1 : (int) mgKnobs ()->partPacked;   /* test knob: 0 forces the wide format */
  const bool packed = packEnv && t->kbits >= j + 4 && f.remBits + f.ordBits <= 64;
  MgStatus s;
  const U64 *bucketStart = fineStart;
  const int firstMode = segSrc ? MG_EL_SEG : MG_EL_DENSE;
  if (!loB)
    { if ((s = mgPartPass (t, firstMode, packed, f, dKmer, 0, n, whole, 1, 0, (U32) 1 << hiB, kB, tB, fineStart, fineCursor, fineCount, chunkBase, st, pre, segSrc, subSeg))) return s; }
  else
    { /* the first pass leaves every element's fine digit in a byte beside it (in cB, which the dedup kernel only writes later): the second
         pass counts 156 MB of bytes instead of reading 1.25 GB of elements twice */
      unsigned char *digits = (loB <= 8 && mgKnobs ()->partDigits != 0) ? reinterpret_cast<unsigned char *> (cB) : 0;
      if ((s = mgPartPass (t, firstMode, packed, f, dKmer, 0, n, whole, 1, loB, (U32) 1 << hiB, kA, tA, coarseStart, coarseCursor, coarseCount, chunkBase, st, pre, segSrc, subSeg,
                           0, 0, 0, 0, digits, 0, (U32) 1 << loB))) return s;
      if ((s = mgPartPass (t, packed ? MG_EL_PACKED : MG_EL_WIDE, packed, f, kA, tA, n, coarseStart, (U32) 1 << hiB, 0, (U32) 1 << loB, kB, tB, fineStart, fineCursor, fineCount, chunkBase, st,
                           0, 0, 0, 0, 0, 0, 0, 0, 0, 0, digits))) return s;
    }

  MgBucketArgs a;
  /* flag polarity from what the previous bucketed add saw (MgTable.newPct: new entries per 100 modimizers) */
  { const int polEnv = mgKnobs ()->flagPolarity == MG_KNOB_UNSET ? -1 : (int) mgKnobs ()->flagPolarity;   /* test knob: 0 / 1 force it */
    a.markDup = polEnv >= 0 ? (polEnv ? 1 : 0) : (t->newPct > 50 ? 1 : 0);
  }
  MG_HIP (hipMemsetAsync (flags, a.markDup ? 1 : 0, n, st));
  /* the merge kernel takes the slots from the dedup kernel when the buckets will be more than half full (there its
     probing costs more than the dedup kernel's extra work: a 12.5 Gbp block at load 0.62 gains 0.4 ms, config 2 at 0.38
     nothing); the load is estimated from the share of new k-mers the previous add saw */
  { const int slotEnv = mgKnobs ()->mergeSlots == MG_KNOB_UNSET ? -1 : (int) mgKnobs ()->mergeSlots;   /* test knob: 0 / 1 force it */
    const U64 expectNew = t->newPct > 0 ? n * (U64) t->newPct / 100 : n;
    const bool dense = ((U64) t->max + expectNew) * 2 > t->nSlots;
    a.slotShift = (t->kbits <= MG_SLOT_SHIFT && (slotEnv >= 0 ? slotEnv != 0 : dense)) ? MG_SLOT_SHIFT : 0;
  }
  /* An add into an EMPTY table (a set built from one batch: every step of the benchmarks, the first file of a run) does not know its
     entries until the dedup kernel has counted them -- the table was sized from the batch's occurrences, an upper bound -- and nothing is
     in the table yet, so its geometry is still free: after the dedup kernel R can be brought down to what the entries need at the tight
     load, and the merge kernel streams back that much less.  The price: the dedup kernel's image is then not the merge kernel's, so no
     slots are carried over and the merge kernel claims its own -- and a workgroup waits for its LONGEST probe chain (8 links at load 0.38,
     30 at 0.6, 80 at 0.75; profiles/r06_ab_table_geometry.txt: the merge kernel 1.07 / 1.20 / 1.44 / 2.5 ms at tight loads 50 / 60 / 70 / 80
     per cent on config 2, against 0.65 ms with carried slots at load 0.77).  So the table is only tightened where that pays: when the share
     of new k-mers the previous add saw (newPct; unknown: all new) says the entries will leave a quarter of the slots and more unused even at
     the tight load -- reads of deep coverage with few errors (config 5: a sixth of the modimizers are new; 4.3 -> 1.07 GB of bucket images). */
  int tightPct = t->tightPct ? t->tightPct : MG_TIGHT_PCT_DEFAULT;
  { const long tk = mgKnobs ()->tightLoad; if (tk != MG_KNOB_UNSET && tk >= 0 && tk <= 95) tightPct = (int) tk; }
  bool tighten = wasEmpty && tightPct > 0 && t->log2NB > 0 && t->pin;
  if (tighten && mgKnobs ()->tightLoad == MG_KNOB_UNSET)          /* (the knob forces it: tests, sweeps) */
    { const U64 expectNew = t->newPct > 0 ? n * (U64) t->newPct / 100 : n;
      tighten = expectNew * 100 / (U64) tightPct < t->nSlots - t->nSlots / 4;
    }
  if (tighten) a.slotShift = 0;
  a.slots = t->slots; a.g = g; a.nBuckets = (U32) NB; a.bucketStart = bucketStart;
  a.pK = kB; a.pT = tB; a.pC = cB; a.uniqCount = uniqCount; a.occ = t->occ; a.flags = flags;
  a.grp = grp; a.baseMax = t->max; a.size = t->size; a.withDepth = withDepth;
  /* slices of the ordinal range for the rank lookups: 2^sliceShift ordinals each (2^22: 1 MiB of rank records, and a
     bucket's share of a slice is usually shorter than a wave), at most MG_RANK_GROUPS - 1 of them */
  { a.sliceShift = 22;
    /* a batch much smaller than the headline's would be three or four slices -- and the lookup kernel gives a slice to an XCD, so half
       the chip would idle (1 Gbp batches, as the file entry points make them: the rank lookups 0.31 ms of 1.39, with 2^18-ordinal slices
       0.11 of 1.11): smaller slices until there are 32 of them */
    while (a.sliceShift > 16 && ((n - 1) >> a.sliceShift) + 1 < 32) --a.sliceShift;
    if (mgKnobs ()->rankSliceShift != MG_KNOB_UNSET) a.sliceShift = (int) mgKnobs ()->rankSliceShift;   /* dev knob */
    while (((n - 1) >> a.sliceShift) + 1 > (U64) (MG_RANK_GROUPS - 1)) ++a.sliceShift;
    a.nSlices = (U32) (((n - 1) >> a.sliceShift) + 1);
    a.sliceOff = sliceOff;
  }
  a.counters = t->counters; a.f = f;
#ifdef MG_ABLATE
  { const long dbg = mgKnobs ()->bucketDebug; a.debug = dbg != MG_KNOB_UNSET ? (int) dbg : 0; }
#endif
  /* depth histogram on the fly: possible when this add builds the whole set (empty before, no host depths) */
  const bool track = t->max == 0 && t->baseZero;
  a.liveHist = 0;
  if (track)
    { if (!t->liveHist) MG_HIP (hipMalloc ((void **) &t->liveHist, 65536 * sizeof (U64)));
      MG_HIP (hipMemsetAsync (t->liveHist, 0, 65536 * sizeof (U64), st));
      a.liveHist = (unsigned long long *) t->liveHist;
    }
  t->liveHistValid = track;
  size_t lds = (size_t) t->R * 16 + 16 + MG_LIVE_BINS * 4;
  { const size_t ldsDedup = (size_t) t->R * 16 + MG_RANK_GROUPS * 4; if (ldsDedup > lds) lds = ldsDedup; }
  const bool bigR = t->R > MG_DEDUP_PER * 1024u;       /* R = 8192: the dedup kernel's threads take eight slots each */
  if (t->R > MG_DEDUP_PER_BIG * 1024u) { mgSetError ("internal: bucket of %u slots", t->R); return MG_ERR_ARG; }
  if (lds > 48 * 1024)
    {
#define MG_DEDUP_ATTR(PK, SL, PER) do { MG_HIP (hipFuncSetAttribute ((const void *) mgBucketDedupKernel<PK, SL, PER, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); \
                                        MG_HIP (hipFuncSetAttribute ((const void *) mgBucketDedupKernel<PK, SL, PER, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); } while (0)
      if (bigR) { MG_DEDUP_ATTR (true, true, MG_DEDUP_PER_BIG); MG_DEDUP_ATTR (true, false, MG_DEDUP_PER_BIG); MG_DEDUP_ATTR (false, true, MG_DEDUP_PER_BIG); MG_DEDUP_ATTR (false, false, MG_DEDUP_PER_BIG); }
      else      { MG_DEDUP_ATTR (true, true, MG_DEDUP_PER); MG_DEDUP_ATTR (true, false, MG_DEDUP_PER); MG_DEDUP_ATTR (false, true, MG_DEDUP_PER); MG_DEDUP_ATTR (false, false, MG_DEDUP_PER); }
#undef MG_DEDUP_ATTR
      MG_HIP (hipFuncSetAttribute ((const void *) mgBucketMergeKernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    }
  const int bThreadsEnv = mgKnobs ()->bucketT == MG_KNOB_UNSET ? 0 : (int) mgKnobs ()->bucketT;
  unsigned bThreads = bThreadsEnv ? (unsigned) bThreadsEnv : (t->R >= 4096 ? 1024u : (t->R >= 2048 ? 512u : 256u));
  while (bThreads < 1024 && (U64) bThreads * MG_DEDUP_PER < t->R) bThreads *= 2;
  unsigned bGrid = (unsigned) (NB < 4096 ? NB : 4096);
  U32 perBlock = (U32) ((NB + bGrid - 1) / bGrid);
  bGrid = (unsigned) ((NB + perBlock - 1) / perBlock);
  if ((U64) bThreads * (bigR ? MG_DEDUP_PER_BIG : MG_DEDUP_PER) < t->R)      /* every slot of the image must belong to a thread of the closing sweep */
    { mgSetError ("internal: %u threads for a bucket of %u slots", bThreads, t->R); return MG_ERR_ARG; }
  /* oversize buckets first: their chunks reduced to weighted entries by a workgroup each (nothing to do on ordinary data:
     the plan finds no bucket and the reduce kernel's workgroups leave at once) */
  mgHotKnobs (&a.hotSplit, &a.hotChunk);
  a.hotItems = hotItems; a.hotCount = hotCount; a.hotBuckets = hotBuckets;
  MG_HIP (hipMemsetAsync (hotCount, 0, 16, st));
  MG_LAUNCH (MG_K_HOT_REDUCE, st, mgHotPlanKernel, dim3 ((unsigned) ((NB + 255) / 256)), dim3 (256), 0, st, a);
  { const size_t ldsHot = (size_t) t->R * 16 + 16;
    if (packed) { if (ldsHot > 48 * 1024) MG_HIP (hipFuncSetAttribute ((const void *) mgHotReduceKernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsHot));
                  MG_LAUNCH (MG_K_HOT_REDUCE, st, mgHotReduceKernel<true>, dim3 (1024), dim3 (bThreads), ldsHot, st, a); }
    else        { if (ldsHot > 48 * 1024) MG_HIP (hipFuncSetAttribute ((const void *) mgHotReduceKernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsHot));
                  MG_LAUNCH (MG_K_HOT_REDUCE, st, mgHotReduceKernel<false>, dim3 (1024), dim3 (bThreads), ldsHot, st, a); }
  }
#define MG_DEDUP_LAUNCH(PK, SL, PER) do { MG_LAUNCH (MG_K_BUCKET_DEDUP, st, (mgBucketDedupKernel<PK, SL, PER, false>), dim3 (bGrid), dim3 (bThreads), lds, st, a, perBlock); \
                                          MG_LAUNCH (MG_K_HOT_REDUCE, st, (mgBucketDedupKernel<PK, SL, PER, true>), dim3 (64), dim3 (bThreads), lds, st, a, perBlock); } while (0)
#define MG_DEDUP_PICK(PER) do { if (packed) { if (a.slotShift) MG_DEDUP_LAUNCH (true, true, PER); else MG_DEDUP_LAUNCH (true, false, PER); } \
                                else        { if (a.slotShift) MG_DEDUP_LAUNCH (false, true, PER); else MG_DEDUP_LAUNCH (false, false, PER); } } while (0)
  if (bigR) MG_DEDUP_PICK (MG_DEDUP_PER_BIG); else MG_DEDUP_PICK (MG_DEDUP_PER);
#undef MG_DEDUP_PICK
#undef MG_DEDUP_LAUNCH
  if (tighten)
    { unsigned long long *st2 = (unsigned long long *) (t->counters + 2);
      MG_HIP (hipMemsetAsync (st2, 0, 16, st));
      MG_LAUNCH (MG_K_RANK_SCAN, st, mgUniqStatsKernel, dim3 ((unsigned) (NB / 256 < 256 ? (NB + 255) / 256 : 256)), dim3 (256), 0, st, uniqCount, (U32) NB, st2);
      MG_HIP (hipMemcpyAsync (t->pin, t->counters + 1, 24, hipMemcpyDeviceToHost, st));      /* overflow flag, entries, the fullest bucket's */
      MG_HIP (hipStreamSynchronize (st));
      const U64 over = t->pin[0], U = t->pin[1], M = t->pin[2];
      if (!over)
        { U64 Rn = (U * 100 / ((U64) NB * (U64) tightPct) + 1 + MG_R_QUANTUM - 1) / MG_R_QUANTUM * MG_R_QUANTUM;
          const U64 Rfit = (M + M / 8 + 16 + MG_R_QUANTUM - 1) / MG_R_QUANTUM * MG_R_QUANTUM;      /* the fullest bucket at load 0.89 at most */
          if (Rn < Rfit) Rn = Rfit;
          if (Rn < MG_TIGHT_MIN_R) Rn = MG_TIGHT_MIN_R;
          if (Rn < t->R)
            { t->R = (U32) Rn; t->nSlots = (U64) NB * Rn; a.g = mgGeomOf (t);
              lds = (size_t) t->R * 16 + 16 + MG_LIVE_BINS * 4;
            }
        }
    }
  MG_LAUNCH (MG_K_RANK_COUNT, st, mgRankCountKernel, dim3 (nRankBlocks), dim3 (256), 0, st, flags, n, rankTiles, blockCount);
  MG_LAUNCH (MG_K_RANK_SCAN, st, mgRankScanKernel, dim3 (1), dim3 (1024), 0, st, blockCount, nRankBlocks * 4, blockBase, t->counters);
  if (segSrc)
    MG_LAUNCH (MG_K_TABLE_ASSIGN, st, (mgRankAssignKernel<false, true>), dim3 (nRankBlocks), dim3 (256), 0, st,
               flags, (const U64 *) 0, *segSrc, subSeg, n, rankTiles, blockBase, t->max, t->size, t->value, t->slots, (const U32 *) 0, grp);
  else
    MG_LAUNCH (MG_K_TABLE_ASSIGN, st, (mgRankAssignKernel<false, false>), dim3 (nRankBlocks), dim3 (256), 0, st,
               flags, dKmer, noSrc, (const MgSubSeg *) 0, n, rankTiles, blockBase, t->max, t->size, t->value, t->slots, (const U32 *) 0, grp);
  { const U32 groupsPerSlice = (U32) ((NB + 63) / 64);
    const U32 blocksPerSlice = (groupsPerSlice + 3) / 4, rounds = (a.nSlices + 7) / 8;
    MG_LAUNCH (MG_K_RANK_LOOKUP, st, mgRankLookupKernel, dim3 (8 * rounds * blocksPerSlice), dim3 (256), 0, st, a, groupsPerSlice);
  }
  MG_LAUNCH (MG_K_BUCKET_MERGE, st, mgBucketMergeKernel, dim3 (bGrid), dim3 (bThreads), lds, st, a, perBlock);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}

__global__ void mgMarkOccKernel (MgGeom g, const U64 *__restrict__ kmer, U64 n, U32 *__restrict__ occ)
{
  U64 o = (U64) blockIdx.x * blockDim.x + threadIdx.x;
  const U64 stride = (U64) gridDim.x * blockDim.x;
  for ( ; o < n ; o += stride)
    { U32 b = mgBucketOfM (mgMixK (kmer[o], g.kbits), g);
      if (!occ[b]) occ[b] = 1;
    }
}

MgStatus mgTableMarkOccupied (MgTable *t, const U64 *dKmer, U64 n, hipStream_t st)
{
  MG_LAUNCH (MG_K_TABLE_FLAG, st, mgMarkOccKernel, dim3 (mgGrid (n)), dim3 (256), 0, st, mgGeomOf (t), dKmer, n, t->occ);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}

MgStatus mgTableFind (MgTable *t, const U64 *dKmer, U64 n, U32 *dIndexOut, hipStream_t st)
{
  if (!n) return MG_OK;
  /* with the never-written buckets zeroed once, a probe needs no look at occ[] first */
  { MgStatus cs = mgTableClean (t, st); if (cs) return cs; }
  const unsigned fgrid = mgGrid ((n + MG_FIND_PER - 1) / MG_FIND_PER);
  MG_LAUNCH (MG_K_TABLE_FIND, st, mgTableFindKernel<false>, dim3 (fgrid), dim3 (256), 0, st, t->slots, t->occ, mgGeomOf (t), dKmer, n, dIndexOut);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}

MgStatus mgTableFindSegments (MgTable *t, const MgSegSrc &src, U64 n, U32 *dIndexOut, hipStream_t st)
{
  if (!n) return MG_OK;
  { MgStatus cs = mgTableClean (t, st); if (cs) return cs; }
  const U64 nRows = (n + 63) / 64;
  U64 waves = 256ull * 4 * 8;                           /* eight waves per SIMD */
  if (waves > nRows) waves = nRows;
  const U64 rowsPerWave = (nRows + waves - 1) / waves;
  waves = (nRows + rowsPerWave - 1) / rowsPerWave;
  MG_LAUNCH (MG_K_TABLE_FIND_SEG, st, mgTableFindSegKernel, dim3 ((unsigned) ((waves + 3) / 4)), dim3 (256), 0, st, t->slots, mgGeomOf (t), src, n, rowsPerWave, dIndexOut);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}

/* scratch of the partitioned lookup for a batch of n: the run table and the partition's small arrays (the packed elements,
   8 bytes each, go where the caller says: the scan's unused dense k-mer array) */
size_t mgTableFindPartScratchBytes (U64 n)
{
  return mgAl ((n / MG_PART_SUB + 2) * (size_t) MG_PART_MAXBINS * 8) + mgAl (((U64) MG_PART_MAXBINS + 2) * 8) * 2
       + mgAl (((U64) MG_PART_MAXBINS + 2) * 8 * 16) + mgAl ((MG_PART_MAXBINS + 2) * 4)
       + mgAl ((MG_PART_MAXBINS + 2 + n / MG_PART_CHUNK + MG_PART_MAXBINS + 2) * 4) + mgAl ((n / MG_PART_SUB + 2) * sizeof (MgSubSeg)) + 4096;
}

/* the two-level path's extra scratch: the second pass's output (8 bytes an element), the results beside the first pass's output
   (4), the second run table, the fine starts / cursors / counts and its chunk table */
size_t mgTableFindPart2ScratchBytes (U64 n)
{
  const U64 NB = (U64) 1 << 18;
  return mgAl (n * 8) + mgAl (n * 4) + mgAl ((2 * (n / (2 * (U64) MG_PART_SUB) + MG_PART_MAXBINS + 2)) * (size_t) MG_PART_MAXBINS * 8)
       + mgAl ((NB + 2) * 8) + mgAl ((NB + 2 + (U64) MG_PART_MAXBINS * 16) * 8) + mgAl ((NB + 2) * 4)
       + mgAl ((MG_PART_MAXBINS + 2 + n / MG_PART_CHUNK + MG_PART_MAXBINS + 2) * 4) + 4096;
}

/* does a lookup batch take the partitioned path?  It needs the scan's digit counts for this table geometry, elements that
   fit one word, and a table the direct probes would have to fetch from HBM */
/* The digit of the partitioned lookup: the top bits of the bucket id one pass sorts by; a bin's piece of the table is
   nSlots * 16 bytes >> bits (MODGPU_FIND_BITS: dev, 3..9) */
int mgTableFindDigitBits (const MgTable *t)
{
  const long kb = mgKnobs ()->findBits;
  int hiB, loB; mgPartSplit (t->log2NB, &hiB, &loB);          /* the build's own first digit: 256 bins of 8 MB at config 3 (512 bins of 4 MB: scatter and pull-back cost 0.7 ms more, the lookups gain nothing) */
  int bits = kb != MG_KNOB_UNSET && kb >= 3 && kb <= 9 ? (int) kb : hiB;
  if (bits > t->log2NB) bits = t->log2NB;
  return bits;
}

bool mgTableFindTakesPartition (const MgTable *t, U64 n, const MgHistReq *counted)
{
  const long kp = mgKnobs ()->findPath;                      /* test knob: 'p' / 'd' force it */
  if (kp == 'd') return false;
  if (!n || !counted || !counted->binCount || counted->log2NB != t->log2NB || counted->kbits != t->kbits) return false;
  const int hiB = counted->hiB;
  if (hiB != mgTableFindDigitBits (t)) return false;
  const int ordBits = mgLog2 (n) > 1 ? mgLog2 (n) : 1;
  if (!(t->kbits >= t->log2NB + 4 && t->kbits - hiB + ordBits <= 64) || hiB < 3) return false;
  if (t->kbits < 24 || mgKnobs ()->scanHist == 0) return false;          /* (the counts must be the scan's own: mgLaunchScanRange) */
  if (kp == 'p' || kp == '2') return true;
  /* by itself: the two-level path where the direct probes would fetch a line from HBM per lookup -- a table of 256 MB and more,
     a batch that fills the chip (config 3: 3.5 against 4.2-4.6 ms per 1.56e8 lookups; one level ties with the direct probes) */
  return n >= ((U64) 1 << 24) && t->nSlots >= ((U64) 1 << 24) && t->log2NB > 9;
}

MgStatus mgTableFindPartitioned (MgTable *t, const MgSegSrc &segSrc, U64 n, const MgHistReq *counted, U32 *dIndexOut, U64 *el, void *scratch, hipStream_t st,
                                 void *scratch2)
{
  if (!n) return MG_OK;
  { MgStatus cs = mgTableClean (t, st); if (cs) return cs; }
  char *wb = (char *) scratch;
  unsigned long long *runTab = (unsigned long long *) wb;    wb += mgAl ((n / MG_PART_SUB + 2) * (size_t) MG_PART_MAXBINS * 8);
  U64 *binStart = (U64 *) wb;                                wb += mgAl (((U64) MG_PART_MAXBINS + 2) * 8);
  U64 *whole = (U64 *) wb;                                   wb += mgAl (((U64) MG_PART_MAXBINS + 2) * 8);
  unsigned long long *cursor = (unsigned long long *) wb;    wb += mgAl (((U64) MG_PART_MAXBINS + 2) * 8 * 16);
  U32 *binCount = (U32 *) wb;                                wb += mgAl ((MG_PART_MAXBINS + 2) * 4);
  U32 *chunkBase = (U32 *) wb;                               wb += mgAl ((MG_PART_MAXBINS + 2 + n / MG_PART_CHUNK + MG_PART_MAXBINS + 2) * 4);
  MgSubSeg *subSeg = (MgSubSeg *) wb;                        wb += mgAl ((n / MG_PART_SUB + 2) * sizeof (MgSubSeg));
  const int hiB = counted->hiB, loB = t->log2NB - hiB;
  MgPartFmt f; f.ordBits = mgLog2 (n) > 1 ? mgLog2 (n) : 1; f.remBits = t->kbits - hiB; f.loB = loB;
  const U32 nBins = (U32) 1 << hiB;
  U64 segInit[2] = { 0, n };
  MG_HIP (hipMemcpyAsync (whole, segInit, 16, hipMemcpyHostToDevice, st));
  U32 subElems = 0;
  const bool twoLevels = scratch2 && loB > 0 && ((U32) 1 << loB) <= MG_PART_MAXBINS;
  /* (two levels) the fine digits as bytes beside the first pass's output, for the second pass's counts: in idxA, which is only written by the first pull */
  unsigned char *digits = (twoLevels && loB <= 8 && mgKnobs ()->partDigits != 0) ? reinterpret_cast<unsigned char *> ((char *) scratch2 + mgAl (n * 8)) : 0;
  MgStatus s = mgPartPass (t, MG_EL_SEG, true, f, (const U64 *) 0, 0, n, whole, 1, loB, nBins, el, (U32 *) 0, binStart, cursor, binCount, chunkBase, st,
                           counted->binCount, &segSrc, subSeg, runTab, &subElems, 0, 0, digits, 0, (U32) 1 << loB);
  if (s) return s;
  if (twoLevels)      /* ---- two levels: the lookups bucket by bucket out of LDS (a table of few buckets has no fine digit: one level) ---- */
    { char *w2 = (char *) scratch2;
      const U64 NB = (U64) 1 << t->log2NB;
      U64 *el2 = (U64 *) w2;                                   w2 += mgAl (n * 8);
      U32 *idxA = (U32 *) w2;                                  w2 += mgAl (n * 4);
      unsigned long long *runTab2 = (unsigned long long *) w2; w2 += mgAl ((2 * (n / (2 * (U64) MG_PART_SUB) + MG_PART_MAXBINS + 2)) * (size_t) MG_PART_MAXBINS * 8);
      U64 *fineStart = (U64 *) w2;                             w2 += mgAl ((NB + 2) * 8);
      unsigned long long *fineCursor = (unsigned long long *) w2; w2 += mgAl ((NB + 2 + (U64) MG_PART_MAXBINS * 16) * 8);
      U32 *fineCount = (U32 *) w2;                             w2 += mgAl ((NB + 2) * 4);
      U32 *chunkBase2 = (U32 *) w2;                            w2 += mgAl ((MG_PART_MAXBINS + 2 + n / MG_PART_CHUNK + MG_PART_MAXBINS + 2) * 4);
      U32 sub2 = 0, maxChunks2 = 0;
      const U32 nBins2 = (U32) 1 << loB;
      s = mgPartPass (t, MG_EL_PACKED, true, f, el, (const U32 *) 0, n, binStart, nBins, 0, nBins2, el2, (U32 *) 0, fineStart, fineCursor, fineCount, chunkBase2, st,
                      (const U32 *) 0, (const MgSegSrc *) 0, (MgSubSeg *) 0, runTab2, &sub2, 1, &maxChunks2, 0, 0, 0, digits);
      if (s) return s;
      unsigned bGrid = (unsigned) (NB < 4096 ? NB : 4096);
      const U32 perBlock = (U32) ((NB + bGrid - 1) / bGrid);
      bGrid = (unsigned) ((NB + perBlock - 1) / perBlock);
      const int remB = t->kbits - t->log2NB;
      if (remB >= 1 && remB <= 32 && mgKnobs ()->find8 != 0)          /* (rem + 1 <= 2^32 above a 31-bit index: one word) */       /* (test knob MODGPU_FIND8=0: the 16-byte table itself) */
        { if (!t->find8 || t->find8Version != t->version || t->find8Cap < t->nSlots)
            { if (t->find8Cap < t->nSlots)
                { if (t->find8) { MG_HIP (hipStreamSynchronize (st)); MG_HIP (hipFree (t->find8)); t->find8 = 0; t->find8Cap = 0; }
                  MG_HIP (hipMalloc ((void **) &t->find8, t->nSlots * sizeof (U64)));
                  t->find8Cap = t->nSlots;
                }
              MG_LAUNCH (MG_K_TABLE_LOAD, st, mgTablePack8Kernel, dim3 ((unsigned) (NB < 8192 ? NB : 8192)), dim3 (256), 0, st, t->slots, t->occ, (U32) NB, t->R, remB, t->find8);
              t->find8Version = t->version;
            }
          const size_t lds8 = (size_t) t->R * 8 + 16;
          if (lds8 > 48 * 1024) MG_HIP (hipFuncSetAttribute ((const void *) mgBucketFind8Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds8));
          MG_LAUNCH (MG_K_BUCKET_FIND, st, mgBucketFind8Kernel, dim3 (bGrid), dim3 (1024), lds8, st, t->find8, t->occ, mgGeomOf (t), f, (U32) NB, remB, fineStart, el2, perBlock);
        }
      else
        { const size_t lds = (size_t) t->R * 12 + 16;
          if (lds > 48 * 1024) MG_HIP (hipFuncSetAttribute ((const void *) mgBucketFindKernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
          MG_LAUNCH (MG_K_BUCKET_FIND, st, mgBucketFindKernel, dim3 (bGrid), dim3 (1024), lds, st, t->slots, t->occ, mgGeomOf (t), f, (U32) NB, fineStart, el2, perBlock);
        }
      const unsigned g2 = 2 * maxChunks2 < 2048 ? 2 * maxChunks2 : 2048;
      if (sub2 == MG_PART_SUB_BIG) MG_LAUNCH (MG_K_UNPART, st, mgUnpartPosKernel<MG_PART_SUB_BIG>, dim3 (g2), dim3 (1024), 0, st, el2, runTab2, nBins2, binStart, chunkBase2, nBins, 2 * sub2, idxA);
      else                         MG_LAUNCH (MG_K_UNPART, st, mgUnpartPosKernel<MG_PART_SUB>, dim3 (g2), dim3 (1024), 0, st, el2, runTab2, nBins2, binStart, chunkBase2, nBins, 2 * sub2, idxA);
      const U64 nSub1 = (n + subElems - 1) / subElems;
      const unsigned g1 = (unsigned) (nSub1 < 2048 ? nSub1 : 2048);
      if (subElems == MG_PART_SUB_BIG) MG_LAUNCH (MG_K_UNPART, st, mgUnpartOrdKernel<MG_PART_SUB_BIG>, dim3 (g1), dim3 (1024), 0, st, el, idxA, f.ordBits, runTab, nBins, n, dIndexOut);
      else                             MG_LAUNCH (MG_K_UNPART, st, mgUnpartOrdKernel<MG_PART_SUB>, dim3 (g1), dim3 (1024), 0, st, el, idxA, f.ordBits, runTab, nBins, n, dIndexOut);
      MG_HIP (hipGetLastError ());
      return MG_OK;
    }
  const U32 wgPerXcd = 256;                                  /* 2048 workgroups of 256: eight waves per SIMD */
  MG_LAUNCH (MG_K_BUCKET_FIND, st, mgBinFindKernel, dim3 (8 * wgPerXcd), dim3 (256), 0, st, t->slots, mgGeomOf (t), f, el, binStart, nBins, wgPerXcd);
  const U64 nSub = (n + subElems - 1) / subElems;
  const unsigned ug = (unsigned) (nSub < 2048 ? nSub : 2048);
  if (subElems == MG_PART_SUB_BIG) MG_LAUNCH (MG_K_UNPART, st, mgUnpartKernel<MG_PART_SUB_BIG>, dim3 (ug), dim3 (1024), 0, st, el, runTab, nBins, n, dIndexOut);
  else                             MG_LAUNCH (MG_K_UNPART, st, mgUnpartKernel<MG_PART_SUB>, dim3 (ug), dim3 (1024), 0, st, el, runTab, nBins, n, dIndexOut);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}

MgStatus mgTableLoadHost (MgTable *t, const U64 *dValue, U32 first, U32 last, hipStream_t st)
{
  if (last < first) return MG_OK;
  t->liveHistValid = false; t->empty = false; ++t->version;
  { MgStatus cs = mgTableClean (t, st); if (cs) return cs; }
  MG_LAUNCH (MG_K_TABLE_LOAD, st, mgTableLoadKernel, dim3 (mgGrid ((U64) last - first + 1)), dim3 (256), 0, st,
             t->slots, mgGeomOf (t), dValue, first, last, t->occ, t->counters);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}

MgStatus mgTableExportDepth (MgTable *t, U16 *dDelta, hipStream_t st)
{
  if (!t->max) return MG_OK;
  MG_HIP (hipMemsetAsync (dDelta, 0, (size_t) t->max * sizeof (U16), st));
  t->baseZero = false;                                   /* the fold below writes baseDepth */
  t->pendingDepth = false;
  t->liveHistValid = false;
  { const U32 NB = (U32) 1 << t->log2NB;
    MG_LAUNCH (MG_K_TABLE_EXPORT, st, mgTableExportDepthKernel, dim3 (NB < 8192 ? NB : 8192), dim3 (256), 0, st,
               t->slots, NB, t->occ, t->R, t->baseDepth, dDelta, t->max);
  }
  MG_HIP (hipGetLastError ());
  return MG_OK;
}

/* dHist[i] += live[i] */
__global__ void mgHistAddKernel (const U64 *__restrict__ live, unsigned long long *__restrict__ hist)
{ U32 i = blockIdx.x * blockDim.x + threadIdx.x; if (i < 65536 && live[i]) atomicAdd (&hist[i], (unsigned long long) live[i]); }

MgStatus mgTableHistogram (MgTable *t, U64 *dHist, hipStream_t st)
{
  if (!t->max) return MG_OK;
  if (t->liveHistValid && t->liveHist)                 /* the merge kernel kept it while it built the set */
    { MG_LAUNCH (MG_K_TABLE_HIST, st, mgHistAddKernel, dim3 (256), dim3 (256), 0, st, t->liveHist, (unsigned long long *) dHist);
      MG_HIP (hipGetLastError ());
      return MG_OK;
    }
  { const U32 NB = (U32) 1 << t->log2NB;
    MG_LAUNCH (MG_K_TABLE_HIST, st, mgTableHistKernel, dim3 (NB < 2048 ? NB : 2048), dim3 (256), 0, st,
               t->slots, NB, t->occ, t->R, t->baseZero ? (const U16 *) 0 : t->baseDepth, (unsigned long long *) dHist);
  }
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

/* ======================================================================================== */
/* table life cycle: allocation, lazy zeroing, growth                                         */

/* zero the buckets nothing has been written to (occ == 0); everything else is left alone */
__global__ __launch_bounds__ (256)
void mgCleanEmptyBucketsKernel (MgSlot *__restrict__ slots, MgGeom g, const U32 *__restrict__ occ, U32 nBuckets)
{
  const uint4 z = make_uint4 (0, 0, 0, 0);
  for (U32 b = blockIdx.x ; b < nBuckets ; b += gridDim.x)
    { if (occ[b]) continue;
      for (U32 i = threadIdx.x ; i < g.R ; i += blockDim.x) *reinterpret_cast<uint4 *> (&slots[(U64) b * g.R + i]) = z;
    }
}

/* old table -> new table (another geometry): every assigned entry is re-inserted with its index and count */
__global__ void mgRehashKernel (const MgSlot *__restrict__ oldSlots, U32 oldNB, const U32 *__restrict__ oldOcc, U32 oldR,
                                MgSlot *__restrict__ slots, MgGeom g, U32 *__restrict__ occ, U64 *counters)
{
  for (U32 bk = blockIdx.x ; bk < oldNB ; bk += gridDim.x)
    { if (!oldOcc[bk]) continue;
      const MgSlot *from = oldSlots + (U64) bk * oldR;
      for (U32 i = threadIdx.x ; i < oldR ; i += blockDim.x)
        { uint4 v = *reinterpret_cast<const uint4 *> (&from[i]);
          if (!(v.x | v.y) || !mgIsAssigned (v.z)) continue;
          const unsigned long long key = ((unsigned long long) v.y << 32) | v.x;
          const U64 m = key - 1;                                  /* the key IS the mixed k-mer: no re-hash needed to re-place it */
          const U32 b = mgBucketOfM (m, g);
          const U64 base = (U64) b * g.R;
          U32 at = mgHomeOfM (m, g);
          bool placed = false;
          for (U32 probes = 0 ; probes < g.R ; ++probes)
            { if (slots[base + at].key == 0 && atomicCAS ((unsigned long long *) &slots[base + at].key, 0ull, key) == 0) { placed = true; break; }
              at = mgNextSlot (at, g.R);
            }
          if (!placed) { counters[1] = 1; continue; }
          slots[base + at].ord = v.z; slots[base + at].cnt = v.w;
          if (!occ[b]) occ[b] = 1;
        }
    }
}

/* The same bucket by bucket (round 6): a bucket is a self-contained table and its id a prefix of its keys, so when NB stays or grows a NEW bucket's
   entries all come from ONE old bucket (its id's leading bits).  A workgroup per new bucket reads the parent's slots as a stream, claims the entries
   that are its own in an LDS image of the new geometry and streams the image out: no global atomic, every byte read and written once or twice
   (tools/incremental_probe.py: 4.4 -> 1.3 ms for 9.3e7 entries). */
__global__ __launch_bounds__ (1024)
void mgRehashBucketKernel (const MgSlot *__restrict__ oldSlots, int oldLog2NB, const U32 *__restrict__ oldOcc, U32 oldR,
                           MgSlot *__restrict__ slots, MgGeom g, U32 *__restrict__ occ, U64 *counters)
{
  unsigned long long *sKey = reinterpret_cast<unsigned long long *> (mgDynLds);
  U32 *sOrd = reinterpret_cast<U32 *> (mgDynLds + (size_t) g.R * 8);
  U32 *sCnt = sOrd + g.R;
  U32 *sN = sCnt + g.R;
  const U32 T = blockDim.x, tid = threadIdx.x, NB = 1u << g.log2NB;
  const int up = g.log2NB - oldLog2NB;                                /* >= 0 */
  for (U32 b = blockIdx.x ; b < NB ; b += gridDim.x)
    { const U32 parent = b >> up;
      if (!oldOcc[parent]) continue;                                  /* (uniform) nothing there: the new bucket stays unwritten, occ 0 */
      for (U32 i = tid ; i < g.R ; i += T) { sKey[i] = 0; sOrd[i] = 0; sCnt[i] = 0; }
      if (tid == 0) sN[0] = 0;
      __syncthreads ();
      const MgSlot *from = oldSlots + (U64) parent * oldR;
      U32 mine = 0;
      for (U32 i = tid ; i < oldR ; i += T)
        { const uint4 v = *reinterpret_cast<const uint4 *> (&from[i]);
          if (!(v.x | v.y) || !mgIsAssigned (v.z)) continue;
          const unsigned long long key = ((unsigned long long) v.y << 32) | v.x;
          const U64 m = key - 1;
          if (mgBucketOfM (m, g) != b) continue;
          const U32 at = mgLdsClaim (sKey, g.R, mgHomeOfM (m, g), key);
          if (at == g.R) { counters[1] = 1; continue; }
          sOrd[at] = v.z; sCnt[at] = v.w; ++mine;
        }
      if (mine) atomicAdd (&sN[0], mine);
      __syncthreads ();
      const U32 n = sN[0];
      if (n)                                                          /* (uniform) */
        { for (U32 i = tid ; i < g.R ; i += T)
            { const unsigned long long k = sKey[i];
              typedef unsigned v4u __attribute__ ((ext_vector_type (4)));
              v4u vv = { (U32) k, (U32) (k >> 32), sOrd[i], sCnt[i] };
              __builtin_nontemporal_store (vv, reinterpret_cast<v4u *> (&slots[(U64) b * g.R + i]));
            }
          if (tid == 0) occ[b] = n;
        }
      __syncthreads ();
    }
}

/* Geometry for `want` slots: NB a power of two, R = want / NB rounded up to a multiple of 64, between half of wantR and wantR where the size
   allows (R = 4096: a bucket's image is 64 KiB of LDS); at most 2^18 buckets (two 9-bit partition passes), R up to 8192 beyond that. */
static void mgSetGeometry (MgTable *t, U64 want)
{
  U32 Rmax = t->wantR ? t->wantR : 4096;
  if (Rmax > 8192) Rmax = 8192;                   /* the dedup kernel's threads hold MG_DEDUP_PER (R <= 4096) or MG_DEDUP_PER_BIG slots each */
  if (Rmax < MG_R_QUANTUM) Rmax = MG_R_QUANTUM;
  if (want < MG_R_QUANTUM) want = MG_R_QUANTUM;
  int lgNB = 0;
  while (lgNB < 18 && (want + ((U64) 1 << lgNB) - 1) / ((U64) 1 << lgNB) > Rmax) ++lgNB;
  U64 R = (want + ((U64) 1 << lgNB) - 1) >> lgNB;
  R = (R + MG_R_QUANTUM - 1) / MG_R_QUANTUM * MG_R_QUANTUM;
  if (R > 8192) R = 8192;
  t->R = (U32) R; t->log2NB = lgNB; t->nSlots = R << lgNB;
}

/* an empty table of (at least) `want` slots; memory is only allocated when the capacity is short */
MgStatus mgTableAlloc (MgTable *t, U64 want, hipStream_t st)
{
  mgSetGeometry (t, want);
  const U32 NB = (U32) 1 << t->log2NB;
  if (!t->slots || t->nSlots > t->capSlots)
    { if (t->slots) { MG_HIP (hipStreamSynchronize (st)); MG_HIP (hipFree (t->slots)); t->slots = 0; t->capSlots = 0; }
      MG_HIP (hipMalloc ((void **) &t->slots, t->nSlots * sizeof (MgSlot)));
      t->capSlots = t->nSlots;
    }
  if (!t->occ || NB > t->capNB)
    { if (t->occ) { MG_HIP (hipStreamSynchronize (st)); MG_HIP (hipFree (t->occ)); t->occ = 0; t->capNB = 0; }
      MG_HIP (hipMalloc ((void **) &t->occ, (size_t) NB * sizeof (U32)));
      t->capNB = NB;
    }
  mgTableForget (t, st);
  return MG_OK;
}

void mgTableForget (MgTable *t, hipStream_t st)
{
  (void) hipMemsetAsync (t->occ, 0, ((size_t) 1 << t->log2NB) * sizeof (U32), st);
  t->dirty = true;               /* no memset of the slots: a bucket is defined once something wrote all of it */
  t->liveHistValid = false;
  t->empty = true; ++t->version;
}

MgStatus mgTableClean (MgTable *t, hipStream_t st)
{
  if (!t->dirty) return MG_OK;
  U32 NB = (U32) 1 << t->log2NB;
  MG_LAUNCH (MG_K_MEMSET, st, mgCleanEmptyBucketsKernel, dim3 (NB < 4096 ? NB : 4096), dim3 (256), 0, st, t->slots, mgGeomOf (t), t->occ, NB);
  MG_HIP (hipGetLastError ());
  t->dirty = false;
  return MG_OK;
}

/* slots needed so that `entries` fit at the table's load (0.6 unless the caller set loadPct); at most 2^maxLog2Slots */
static U64 mgSlotsFor (const MgTable *t, U64 entries)
{
  const long lk = mgKnobs ()->tableLoad;
  const int envPct = lk != MG_KNOB_UNSET && lk >= 10 && lk <= 150 ? (int) lk : 0;   /* dev knob (above 100: experiments only) */
  const int loadPct = envPct ? envPct : (t->loadPct ? t->loadPct : 60);
  U64 need = entries * 100 / (U64) loadPct + 1;      /* entries / 0.6 by default */
  if (need < ((U64) 1 << 16)) need = (U64) 1 << 16;
  if (need > ((U64) 1 << t->maxLog2Slots)) need = (U64) 1 << t->maxLog2Slots;
  return need;
}

/* the table in another geometry: the assigned entries re-placed (their keys are their hashes) */
static MgStatus mgTableRehashTo (MgTable *t, U64 want, hipStream_t st)
{
  MgSlot *oldSlots = t->slots; U32 *oldOcc = t->occ; const U32 oldNB = (U32) 1 << t->log2NB, oldR = t->R;
  t->slots = 0; t->occ = 0; t->capSlots = 0; t->capNB = 0;
  MgStatus s = mgTableAlloc (t, want, st); if (s) return s;
  int oldLog2NB = 0; while (((U32) 1 << oldLog2NB) < oldNB) ++oldLog2NB;
  if (t->log2NB >= oldLog2NB && mgKnobs ()->tablePath != 'd')             /* bucket by bucket, out of LDS (the new buckets nothing goes into stay unwritten: dirty again) */
    { const U32 NB = (U32) 1 << t->log2NB;
      const size_t lds = (size_t) t->R * 16 + 16;
      if (lds > 48 * 1024) MG_HIP (hipFuncSetAttribute ((const void *) mgRehashBucketKernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
      MG_HIP (hipMemsetAsync (t->occ, 0, (size_t) NB * sizeof (U32), st));
      MG_LAUNCH (MG_K_TABLE_LOAD, st, mgRehashBucketKernel, dim3 (NB < 2048 ? NB : 2048), dim3 (1024), lds, st,
                 oldSlots, oldLog2NB, oldOcc, oldR, t->slots, mgGeomOf (t), t->occ, t->counters);
      t->dirty = true;
    }
  else
    { if ((s = mgTableClean (t, st))) return s;                          /* (the atomic path claims slots in a zeroed table) */
      MG_LAUNCH (MG_K_TABLE_LOAD, st, mgRehashKernel, dim3 (oldNB < 8192 ? oldNB : 8192), dim3 (256), 0, st,
                 oldSlots, oldNB, oldOcc, oldR, t->slots, mgGeomOf (t), t->occ, t->counters);
    }
  MG_HIP (hipGetLastError ());
  MG_HIP (hipStreamSynchronize (st));
  MG_HIP (hipFree (oldSlots)); MG_HIP (hipFree (oldOcc));
  t->empty = false; ++t->version;
  return MG_OK;
}

MgStatus mgTableEnsure (MgTable *t, U64 nIncoming, hipStream_t st)
{
  U64 want = mgSlotsFor (t, (U64) t->max + nIncoming);
  if (t->slots && want <= t->nSlots) return MG_OK;
  if (!t->slots || !t->max) return mgTableAlloc (t, want, st);
  /* grow: a table that takes adds doubles at least (sized to the entry, every add would re-place everything; doubling re-places every
     entry twice over a set's life, growing by half three times: tools/incremental_probe.py, ten 1 Gbp batches into one set, 34.9 against 30 ms) */
  if (nIncoming && want < 2 * t->nSlots) want = 2 * t->nSlots;
  if (want > ((U64) 1 << t->maxLog2Slots)) want = (U64) 1 << t->maxLog2Slots;
  if (want <= t->nSlots) return MG_OK;
  return mgTableRehashTo (t, want, st);
}

/* ======================================================================================== */
/* whole-set passes on the device: merge (modset.c:106-128) and depth prune (modset.c:64-77)  */

/* after ms2's values were inserted (idx[i] = their index in ms1): depth adds saturate at 65535;
 * copy bits add and saturate at 3 while the entry's other info bits are cleared (modset.c:120-126) */
__global__ void mgMergeApplyKernel (const U32 *__restrict__ idx, const U16 *__restrict__ depth2, const U8 *__restrict__ info2,
                                    U32 n2, U16 *__restrict__ baseDepth, U8 *__restrict__ info1)
{
  U64 i = (U64) blockIdx.x * blockDim.x + threadIdx.x;
  const U64 stride = (U64) gridDim.x * blockDim.x;
  for ( ; i < n2 ; i += stride)
    { U32 j = idx[i];
      if (!j) continue;
      U32 d = (U32) baseDepth[j] + depth2[i];
      baseDepth[j] = (U16) (d > 0xffffu ? 0xffffu : d);
      U32 a = info1[j] & 3, c = a + (info2[i] & 3);
      if (c > 3) c = 3;
      info1[j] = (U8) (a | c);
    }
}

/* keep[i-1] = 1 when entry i survives the prune */
__global__ void mgPruneFlagKernel (const U16 *__restrict__ depth, U32 n, int lo, int hi, unsigned char *__restrict__ keep)
{
  U64 i = (U64) blockIdx.x * blockDim.x + threadIdx.x;
  const U64 stride = (U64) gridDim.x * blockDim.x;
  for ( ; i < n ; i += stride)
    { int d = depth[i + 1];
      keep[i] = (d >= lo && (!hi || d < hi)) ? 1 : 0;
    }
}

/* depth/info of the survivors move to their new index (rank structure from mgRankAssignKernel<false>) */
__global__ void mgPruneMoveKernel (const unsigned char *__restrict__ keep, const MgRankGrp *__restrict__ grp, U32 n,
                                   const U16 *__restrict__ depth, const U8 *__restrict__ info,
                                   U16 *__restrict__ newDepth, U8 *__restrict__ newInfo)
{
  U64 i = (U64) blockIdx.x * blockDim.x + threadIdx.x;
  const U64 stride = (U64) gridDim.x * blockDim.x;
  for ( ; i < n ; i += stride)
    { if (!keep[i]) continue;
      MgRankGrp g = grp[i >> 6];
      U32 r = 1 + g.rank + (U32) __popcll (g.bits & (((U64) 1 << (i & 63)) - 1));
      newDepth[r] = depth[i + 1]; newInfo[r] = info[i + 1];
    }
}

MgStatus mgTableMergeApply (const U32 *dIdx, const U16 *dDepth2, const U8 *dInfo2, U32 n2, U16 *dBaseDepth, U8 *dInfo1, hipStream_t st)
{
  if (!n2) return MG_OK;
  MG_LAUNCH (MG_K_TABLE_EXPORT, st, mgMergeApplyKernel, dim3 (mgGrid (n2)), dim3 (256), 0, st, dIdx, dDepth2, dInfo2, n2, dBaseDepth, dInfo1);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}

size_t mgTablePruneScratchBytes (U32 n)
{ return mgAl (n) + 2 * mgAl ((MG_RANK_UNITS + 8) * 8) + mgAl (((U64) n / 64 + 2) * sizeof (MgRankGrp)) + 4096; }

/* survivors of (lo <= depth < hi) keep their relative order: newValue/newDepth/newInfo[1..*]; counters[0] = how many */
MgStatus mgTablePrune (MgTable *t, const U8 *dInfo, int lo, int hi, U64 *dNewValue, U16 *dNewDepth, U8 *dNewInfo,
                       void *scratch, hipStream_t st)
{
  const U32 n = t->max;
  MG_HIP (hipMemsetAsync (t->counters, 0, 16, st));
  if (!n) return MG_OK;
  char *wb = (char *) scratch;
  unsigned char *keep = (unsigned char *) wb;        wb += mgAl (n);
  U64 *unitCount = (U64 *) wb;                       wb += mgAl ((MG_RANK_UNITS + 8) * 8);
  U64 *unitBase = (U64 *) wb;                        wb += mgAl ((MG_RANK_UNITS + 8) * 8);
  MgRankGrp *grp = (MgRankGrp *) wb;
  U32 nBlocks; U64 rows = mgRankRowsPerUnit (n, &nBlocks);
  MG_HIP (hipMemsetAsync (unitCount, 0, (MG_RANK_UNITS + 8) * 8, st));
  MG_LAUNCH (MG_K_TABLE_FLAG, st, mgPruneFlagKernel, dim3 (mgGrid (n)), dim3 (256), 0, st, t->baseDepth, n, lo, hi, keep);
  MG_LAUNCH (MG_K_RANK_COUNT, st, mgRankCountKernel, dim3 (nBlocks), dim3 (256), 0, st, keep, (U64) n, rows, unitCount);
  MG_LAUNCH (MG_K_RANK_SCAN, st, mgRankScanKernel, dim3 (1), dim3 (1024), 0, st, unitCount, nBlocks * 4, unitBase, t->counters);
  /* value[i+1] of survivor i -> newValue[1 + rank]: the assign kernel with "kmer" = value + 1, base 0 */
  MgSegSrc noSrc; noSrc.segKmer = 0; noSrc.segCount = 0; noSrc.segStart = 0; noSrc.segCap = 0; noSrc.nSegs = 0;
  MG_LAUNCH (MG_K_TABLE_ASSIGN, st, (mgRankAssignKernel<false, false>), dim3 (nBlocks), dim3 (256), 0, st,
             keep, t->value + 1, noSrc, (const MgSubSeg *) 0, (U64) n, rows, unitBase, 0u, 0xffffffffu, dNewValue, t->slots, (const U32 *) 0, grp);
  MG_LAUNCH (MG_K_TABLE_EXPORT, st, mgPruneMoveKernel, dim3 (mgGrid (n)), dim3 (256), 0, st, keep, grp, n, t->baseDepth, dInfo, dNewDepth, dNewInfo);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}
