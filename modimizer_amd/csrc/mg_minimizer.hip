/* mg_minimizer.hip — minimizerRCiterator / minimizerRCnext (seqhash.c:83-152) for a batch of reads.
 *
 * The reference's iterator is not the usual sliding-window minimizer: after it returns the minimum
 * at k-mer position p, the next window is the w k-mers p+1 .. p+w, so the reported positions form a
 * chain p0 -> p1 -> ... in which every link depends on the previous one.  What each link needs —
 * the smallest canonical hash among w consecutive k-mers — is data-parallel; the chain is not.
 * So: one wavefront per read walks the chain; at every link its lanes hash the w k-mers of the window
 * straight from the packed bases (no hash array in memory) and a wave reduction picks the winner.
 * Reads of a batch are independent, so a batch of long reads fills the chip; a single very long
 * sequence is one wavefront's work (the function has no caller in the reference, seqhash.h:45-46).
 *
 * Semantics kept from the reference (each pinned by the golden minimizer lists under tests/golden):
 *   - first window = k-mers 0..w-1, leftmost minimum (strict '<', seqhash.c:100-106);
 *   - later windows: ties go to the smallest ring slot, i.e. smallest (position mod w) (seqhash.c:146-147);
 *   - positions past the last k-mer hold U64MAX and never win (seqhash.c:77-78);
 *   - once the window has run off the end of the read only a hash strictly below the previous return
 *     value is accepted, otherwise the iteration ends (seqhash.c:127,142-149);
 *   - the iteration also ends after a return if the previous window already reached the last base
 *     (seqhash.c:125);
 *   - hashBuf[0] is never written during set-up, so if the first minimum is k-mer 0 the value returned
 *     for it is 0 (seqhash.c:101) — and that 0 is then the bound of the run-off rule.
 * Two passes (count, exclusive scan of the per-read counts, write) give the reads' minimizers in read order.
 */
#include <hip/hip_runtime.h>
#include "mg_common.h"

#define MG_MIN_NONE (~(U64) 0)

/* forward k-mer starting at base position `at` of the packed stream (first base in the top bits) */
__device__ __forceinline__ U64 mgKmerGlobal (const U32 *__restrict__ packed, U64 at, int sh1)
{
  const U64 wi = at >> 4; const int s = 2 * (int) (at & 15);
  const U32 x0 = packed[wi], x1 = packed[wi + 1], x2 = packed[wi + 2];
  U64 hi = ((U64) x0 << 32) | x1;
  if (s) hi = (hi << s) | (U64) (x2 >> (32 - s));
  return hi >> sh1;
}

struct MgMinPick { U64 h; U32 q; U32 key; bool fwd; };

/* the best of count k-mers starting at position q0 of the read: smallest (hash, key), where key is the
 * position itself (first window) or its ring slot; lanes take positions q0+lane, q0+lane+64, ... */
__device__ __forceinline__ MgMinPick mgWindowMin (const U32 *__restrict__ packed, U64 readStart, U32 nk, const MgHashParams &p,
                                                  U32 q0, U32 count, bool keyIsPos, U32 slot0, U32 w, int lane)
{
  MgMinPick b; b.h = MG_MIN_NONE; b.q = 0; b.key = 0xffffffffu; b.fwd = false;
  for (U32 i = (U32) lane ; i < count ; i += 64)
    { const U32 q = q0 + i;
      if (q >= nk) break;                              /* U64MAX: never below any bound */
      const U64 F = mgKmerGlobal (packed, readStart + q, p.shift1);
      const U64 R = mgRevComp (F, p.shift1);
      const U64 hF = (F * p.factor1) >> p.shift1, hR = (R * p.factor1) >> p.shift1;   /* seqhash.h:58 */
      const bool fwd = hF < hR;                        /* ties -> reverse (seqhash.c:66-67) */
      const U64 h = fwd ? hF : hR;
      U32 key = q;
      if (!keyIsPos) { key = slot0 + i; while (key >= w) key -= w; }
      if (h < b.h || (h == b.h && key < b.key)) { b.h = h; b.q = q; b.key = key; b.fwd = fwd; }
    }
#pragma unroll
  for (int off = 32 ; off ; off >>= 1)
    { const U32 hl = __shfl_xor ((U32) b.h, off), hh = __shfl_xor ((U32) (b.h >> 32), off);
      const U32 oq = __shfl_xor (b.q, off), ok = __shfl_xor (b.key, off);
      const int of = __shfl_xor ((int) b.fwd, off);
      const U64 oh = ((U64) hh << 32) | hl;
      if (oh < b.h || (oh == b.h && ok < b.key)) { b.h = oh; b.q = oq; b.key = ok; b.fwd = of != 0; }
    }
  return b;
}

template <bool WRITE>
__global__ __launch_bounds__ (256)
void mgMinimizerKernel (const U32 *__restrict__ packed, const U64 *__restrict__ readOff, U32 nReads,
                        const MgHashParams p, U32 w, U64 *__restrict__ perRead,
                        U64 *__restrict__ outHash, U32 *__restrict__ outPosF, U64 capacity)
{
  const int lane = threadIdx.x & 63;
  const U32 wavesPerGrid = gridDim.x * (blockDim.x >> 6);
  for (U32 r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6) ; r < nReads ; r += wavesPerGrid)
    { const U64 rs = readOff[r], len = readOff[r + 1] - rs;
      U64 n = 0;
      const U64 outBase = WRITE ? perRead[r] : 0;
      if (len >= (U64) p.k)                                                     /* seqhash.c:94 */
        { const U32 nk = (U32) (len - (U64) p.k + 1);
          MgMinPick cur = mgWindowMin (packed, rs, nk, p, 0, w < nk ? w : nk, true, 0, w, lane);
          U32 curSlot = cur.q % w;
          bool first = true; bool havePrev = false; U32 prev = 0;
          for (;;)
            { const U64 ret = (first && cur.q == 0) ? 0 : cur.h;               /* the hashBuf[0] quirk */
              if (WRITE && lane == 0 && outBase + n < capacity)
                { outHash[outBase + n] = ret;
                  outPosF[outBase + n] = cur.q | (cur.fwd ? MG_FWD_BIT : 0u);
                }
              ++n; first = false;
              /* all input consumed when this minimum was found (seqhash.c:125) */
              if (havePrev ? ((U64) prev + w >= (U64) nk - 1) : (w >= nk)) break;
              const bool full = (U64) cur.q + w < (U64) nk;                     /* k-mer cur+w exists (seqhash.c:142) */
              const U64 bound = full ? MG_MIN_NONE : ret;
              U32 slot0 = curSlot + 1; if (slot0 >= w) slot0 -= w;
              const MgMinPick nx = mgWindowMin (packed, rs, nk, p, cur.q + 1, w, false, slot0, w, lane);
              if (!(nx.h < bound)) break;                                       /* seqhash.c:148-149 */
              prev = cur.q; havePrev = true;
              cur = nx; curSlot = nx.key;
            }
        }
      if (!WRITE && lane == 0) perRead[r] = n;
    }
}

/* exclusive scan of n counts in place (one workgroup); a[n] = total */
__global__ __launch_bounds__ (1024)
void mgMinScanKernel (U64 *__restrict__ a, U32 n)
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

MgStatus mgLaunchMinimizers (const MgHashParams &p, U32 w, const U32 *dPacked, const U64 *dReadOffsets, U32 nReads,
                             U64 *dHash, U32 *dPosF, U64 *dReadStart, U64 capacity, U64 *totalOut, hipStream_t st)
{
  *totalOut = 0;
  if (!nReads) { MG_HIP (hipMemsetAsync (dReadStart, 0, 8, st)); return MG_OK; }
  unsigned grid = (nReads + 3) / 4; if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL (mgMinimizerKernel<false>, dim3 (grid), dim3 (256), 0, st, dPacked, dReadOffsets, nReads, p, w,
                      dReadStart, (U64 *) 0, (U32 *) 0, (U64) 0);
  hipLaunchKernelGGL (mgMinScanKernel, dim3 (1), dim3 (1024), 0, st, dReadStart, nReads);
  MG_HIP (hipGetLastError ());
  U64 total = 0;
  MG_HIP (hipMemcpyAsync (&total, dReadStart + nReads, 8, hipMemcpyDeviceToHost, st));
  MG_HIP (hipStreamSynchronize (st));
  *totalOut = total;
  if (total > capacity) { mgSetError ("%llu minimizers exceed the caller's capacity %llu", (unsigned long long) total, (unsigned long long) capacity); return MG_ERR_CAPACITY; }
  hipLaunchKernelGGL (mgMinimizerKernel<true>, dim3 (grid), dim3 (256), 0, st, dPacked, dReadOffsets, nReads, p, w,
                      dReadStart, dHash, dPosF, capacity);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}
