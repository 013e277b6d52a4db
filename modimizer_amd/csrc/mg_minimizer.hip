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
 *
 * Round 5: the links are not computed one at a time any more.  A full window's winner depends on the position it follows and on
 * nothing else, and the tie rule -- smallest ring slot, i.e. smallest position mod w -- makes it decomposable: cut the read into
 * blocks of w positions that start where the slot is 0; the window after position q is the tail of q's block from q + 1 (slots
 * s + 1 .. w - 1) and the head of the next block up to q + w (slots 0 .. s), so its winner is the leftmost minimum of the tail or --
 * on a tie too, its slots being smaller -- the leftmost minimum of the head: one comparison of a suffix minimum with a prefix
 * minimum (the sliding-window-minimum construction of van Herk / Gil and Werman, with the reference's tie rule built in).  A wave
 * stages a tile of about a thousand positions in LDS: the 64 lanes hash them in parallel, a lane per block makes the prefix and
 * suffix arg-minima, the lanes combine them into next[] for every position of the tile; the walk along the chain is then one LDS
 * read per link.  Windows that run off the read's end (the last one or two links), the first window and windows wider than
 * MG_MIN_WMAX take the one-link-at-a-time path as before.  24 -> 60 Gbp/s on ONT-like reads at k=19 w=31 (tools/n34_probe.py; tiles of 128 / 256 / 384 / 512 / 768 / 1024
 * positions: 33.5 / 52.1 / 58.9 / 60.2 / 46.5 / 29.2 Gbp/s -- a wave's LDS piece decides how many waves hide each other's walks).
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

#define MG_MIN_WMAX  256                                  /* windows up to this many k-mers go through the tiles */
#define MG_MIN_TILE  512                                  /* positions per tile, at most (a whole number of blocks of w) */
#define MG_MIN_WAVES 4
#define MG_MIN_NOLINK 0xffffu                              /* next[] of a position whose window runs off the read */
#define MG_MIN_WAVE_SYNC() do { __builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier (); \
                                __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "wavefront"); } while (0)
/* a wave's piece of the (dynamic) LDS, laid out for the launch's w: np = TB + w positions -- hashes, strands, the two arg-minimum arrays
   (the suffix one doubles as the list of a run's links), next[] for the tile's own positions, the packed words under them.  8 KB at
   w = 31: what is resident per CU, and with it how much of the walk's latency other waves hide, follows from this size. */
struct MgMinLds { U64 *h; unsigned short *suf, *pre, *nxt; U8 *fwd; U32 *words; };
static inline size_t mgMinWaveBytes (U32 w, U32 k, U32 tile)
{
  const size_t TB = (size_t) (tile / w) * w, np = TB + w, nWords = (15 + np + k - 1 + 15) / 16 + 3;
  size_t b = np * 8 + np * 2 * 2 + TB * 2 + np + nWords * 4;
  return (b + 15) & ~(size_t) 15;
}
__device__ __forceinline__ MgMinLds mgMinCarve (char *base, U32 TB, U32 w, U32 k)
{
  const U32 np = TB + w;
  MgMinLds L;
  L.h = (U64 *) base; base += (size_t) np * 8;
  L.suf = (unsigned short *) base; base += (size_t) np * 2;
  L.pre = (unsigned short *) base; base += (size_t) np * 2;
  L.nxt = (unsigned short *) base; base += (size_t) TB * 2;
  L.words = (U32 *) (((size_t) base + np + 3) & ~(size_t) 3);
  L.fwd = (U8 *) base;
  (void) k;
  return L;
}

/* tile [t0, t0 + TB) of the read (t0 a multiple of w, TB a whole number of blocks): hashes of [t0, t0 + TB + w), then next[] for every
   position of the tile whose window lies inside the read */
__device__ __forceinline__ void mgMinStageTile (const MgMinLds &L, const U32 *__restrict__ packed, U64 readStart, U32 nk, const MgHashParams &p,
                                                U32 t0, U32 TB, U32 w, int lane)
{
  const U32 span = TB + w;
  /* the tile's packed bases, once, coalesced (a thousand positions are 70 words): every k-mer is then cut out of LDS -- fetched per
     position from global memory, three dependent loads each, the staging spent its time waiting */
  const U64 g0 = readStart + t0;
  const U64 w0 = g0 >> 4; const U32 odd = (U32) (g0 & 15);
  const U32 nW = (odd + span + (U32) p.k - 1 + 15) / 16 + 2;
  /* ... but never past the read: the tile's span may reach up to TB bases beyond its last k-mer (all of them masked below), and the stream has
     only MG_PACK_PAD words of slack behind its last base -- the two words a k-mer's extraction reads past its own are covered by that
     (the unclamped load ran ~35 words past the end of the last read of a batch) */
  const U64 wLast = ((readStart + nk + (U32) p.k - 2) >> 4) + 2;
  for (U32 i = (U32) lane ; i < nW ; i += 64) L.words[i] = w0 + i <= wLast ? packed[w0 + i] : 0u;
  MG_MIN_WAVE_SYNC ();
  for (U32 i = (U32) lane ; i < span ; i += 64)
    { const U32 q = t0 + i;
      U64 h = MG_MIN_NONE; bool fwd = false;
      if (q < nk)
        { const U32 at = odd + i; const U32 wi = at >> 4; const int sb = 2 * (int) (at & 15);
          const U32 x0 = L.words[wi], x1 = L.words[wi + 1], x2 = L.words[wi + 2];
          U64 hi = ((U64) x0 << 32) | x1;
          if (sb) hi = (hi << sb) | (U64) (x2 >> (32 - sb));
          const U64 F = hi >> p.shift1;
          const U64 R = mgRevComp (F, p.shift1);
          const U64 hF = (F * p.factor1) >> p.shift1, hR = (R * p.factor1) >> p.shift1;   /* seqhash.h:58 */
          fwd = hF < hR;                                   /* ties -> reverse (seqhash.c:66-67) */
          h = fwd ? hF : hR;
        }
      L.h[i] = h; L.fwd[i] = fwd ? 1 : 0;
    }
  MG_MIN_WAVE_SYNC ();
  /* a lane per (block, direction): leftmost arg-minimum of [block start, i] (going right) and of [i, block end] (going left) */
  const U32 nBlocks = span / w;
  for (U32 job = (U32) lane ; job < 2 * nBlocks ; job += 64)
    { const U32 b = job >> 1, i0 = b * w;
      if (!(job & 1))
        { U32 best = i0; U64 hb = L.h[i0];
          L.pre[i0] = (unsigned short) i0;
          for (U32 j = 1 ; j < w ; ++j)
            { const U64 x = L.h[i0 + j];
              if (x < hb) { hb = x; best = i0 + j; }       /* strictly smaller: the leftmost of equals stays */
              L.pre[i0 + j] = (unsigned short) best;
            }
        }
      else
        { U32 best = i0 + w - 1; U64 hb = L.h[best];
          L.suf[best] = (unsigned short) best;
          for (U32 j = w - 1 ; j-- > 0 ; )
            { const U64 x = L.h[i0 + j];
              if (x <= hb) { hb = x; best = i0 + j; }      /* or equal: going right to left, the leftmost of equals wins */
              L.suf[i0 + j] = (unsigned short) best;
            }
        }
    }
  MG_MIN_WAVE_SYNC ();
  /* next[i]: the winner of positions i + 1 .. i + w -- the tail of i's block (larger slots) against the head of the next one (smaller slots: wins ties) */
  for (U32 i = (U32) lane ; i < TB ; i += 64)
    { U32 nx = MG_MIN_NOLINK;
      if ((U64) t0 + i + w < (U64) nk)                    /* the window lies inside the read (seqhash.c:142): its winner is a real k-mer */
        { const U32 a = L.suf[i + 1];
          nx = a;
          if ((i + 1) % w)
            { const U32 bb = L.pre[i + w];
              if (L.h[bb] <= L.h[a]) nx = bb;
            }
        }
      L.nxt[i] = (unsigned short) nx;
    }
  MG_MIN_WAVE_SYNC ();
}

template <bool WRITE, bool TILED>
__global__ __launch_bounds__ (64 * MG_MIN_WAVES)
void mgMinimizerKernel (const U32 *__restrict__ packed, const U64 *__restrict__ readOff, U32 nReads,
                        const MgHashParams p, U32 w, U64 *__restrict__ perRead,
                        U64 *__restrict__ outHash, U32 *__restrict__ outPosF, U64 capacity, U32 waveBytes, U32 tile)
{
  extern __shared__ __attribute__ ((aligned (16))) char sDyn[];
  const int lane = threadIdx.x & 63;
  const U32 TB = TILED ? (tile / w) * w : 0;              /* (tile >= w: at least one block) */
  const MgMinLds L = TILED ? mgMinCarve (sDyn + (size_t) (threadIdx.x >> 6) * waveBytes, TB, w, (U32) p.k) : MgMinLds ();
  const U32 wavesPerGrid = gridDim.x * (blockDim.x >> 6);
  for (U32 r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6) ; r < nReads ; r += wavesPerGrid)
    { const U64 rs = readOff[r], len = readOff[r + 1] - rs;
      U64 n = 0;
      const U64 outBase = WRITE ? perRead[r] : 0;
      if (len >= (U64) p.k)                                                     /* seqhash.c:94 */
        { const U32 nk = (U32) (len - (U64) p.k + 1);
          MgMinPick cur = mgWindowMin (packed, rs, nk, p, 0, w < nk ? w : nk, true, 0, w, lane);
          bool first = true; bool havePrev = false; U32 prev = 0;
          U32 tile0 = 0xffffffffu;                                              /* first position of the tile staged in LDS */
          for (;;)
            { if (TILED && (U64) cur.q + w < (U64) nk)
                { /* cur's window lies inside the read, and so may those of the links after it: the run of such links inside cur's tile,
                     listed by following next[] (one LDS read a link), written out by all lanes.  None of them can end the iteration: the
                     end rules (seqhash.c:125,142-149) need a window that reaches the read's end; the first link of all alone has the
                     hashBuf[0] quirk.  The walk goes on at the link the run leaves the tile by, or whose window is not full. */
                  const U32 t0 = (cur.q / TB) * TB;
                  if (t0 != tile0) { mgMinStageTile (L, packed, rs, nk, p, t0, TB, w, lane); tile0 = t0; }
                  unsigned short *lst = L.suf;                                  /* (the suffix minima are done with) */
                  U32 i = cur.q - t0, cnt = 0, nxl = L.nxt[i];
                  for (;;)
                    { lst[cnt++] = (unsigned short) i;
                      if (nxl >= TB) break;
                      const U32 nn = L.nxt[nxl];
                      if (nn == MG_MIN_NOLINK) break;
                      i = nxl; nxl = nn;
                    }
                  MG_MIN_WAVE_SYNC ();
                  if (WRITE)
                    for (U32 j = (U32) lane ; j < cnt ; j += 64)
                      if (outBase + n + j < capacity)
                        { const U32 ix = lst[j];
                          outHash[outBase + n + j] = (first && j == 0 && t0 + ix == 0) ? 0 : L.h[ix];
                          outPosF[outBase + n + j] = (t0 + ix) | (L.fwd[ix] ? MG_FWD_BIT : 0u);
                        }
                  n += cnt; first = false;
                  prev = t0 + i; havePrev = true;
                  cur.q = t0 + nxl; cur.h = L.h[nxl]; cur.fwd = L.fwd[nxl] != 0; cur.key = 0;
                  MG_MIN_WAVE_SYNC ();                                          /* (the list is read before the next run writes it) */
                  continue;
                }
              const U64 ret = (first && cur.q == 0) ? 0 : cur.h;               /* the hashBuf[0] quirk */
              if (WRITE && lane == 0 && outBase + n < capacity)
                { outHash[outBase + n] = ret;
                  outPosF[outBase + n] = cur.q | (cur.fwd ? MG_FWD_BIT : 0u);
                }
              ++n; first = false;
              /* all input consumed when this minimum was found (seqhash.c:125) */
              if (havePrev ? ((U64) prev + w >= (U64) nk - 1) : (w >= nk)) break;
              const bool full = (U64) cur.q + w < (U64) nk;                     /* k-mer cur+w exists (seqhash.c:142) */
              const U64 bound = full ? MG_MIN_NONE : ret;
              const U32 slot0 = (cur.q + 1) % w;
              const MgMinPick nx = mgWindowMin (packed, rs, nk, p, cur.q + 1, w, false, slot0, w, lane);
              if (!(nx.h < bound)) break;                                       /* seqhash.c:148-149 */
              prev = cur.q; havePrev = true;
              cur = nx;
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
  unsigned grid = (nReads + MG_MIN_WAVES - 1) / MG_MIN_WAVES; if (grid > 16384) grid = 16384;
  const bool tiled = w <= MG_MIN_WMAX && mgKnobs ()->minTiled != 0;      /* (test knob: 0 = one link at a time for every window) */
  U32 tile = MG_MIN_TILE;
  { const long tk = mgKnobs ()->minTile; if (tk != MG_KNOB_UNSET && tk >= 64 && tk <= 4096) tile = (U32) tk; }      /* dev */
  if (tile < w) tile = w;
  const U32 waveBytes = tiled ? (U32) mgMinWaveBytes (w, (U32) p.k, tile) : 0;
  const size_t lds = (size_t) waveBytes * MG_MIN_WAVES;
  if (tiled) hipLaunchKernelGGL ((mgMinimizerKernel<false, true>), dim3 (grid), dim3 (64 * MG_MIN_WAVES), lds, st, dPacked, dReadOffsets, nReads, p, w, dReadStart, (U64 *) 0, (U32 *) 0, (U64) 0, waveBytes, tile);
  else hipLaunchKernelGGL ((mgMinimizerKernel<false, false>), dim3 (grid), dim3 (64 * MG_MIN_WAVES), 0, st, dPacked, dReadOffsets, nReads, p, w, dReadStart, (U64 *) 0, (U32 *) 0, (U64) 0, 0u, 0u);
  hipLaunchKernelGGL (mgMinScanKernel, dim3 (1), dim3 (1024), 0, st, dReadStart, nReads);
  MG_HIP (hipGetLastError ());
  U64 total = 0;
  MG_HIP (hipMemcpyAsync (&total, dReadStart + nReads, 8, hipMemcpyDeviceToHost, st));
  MG_HIP (hipStreamSynchronize (st));
  *totalOut = total;
  if (total > capacity) { mgSetError ("%llu minimizers exceed the caller's capacity %llu", (unsigned long long) total, (unsigned long long) capacity); return MG_ERR_CAPACITY; }
  if (tiled) hipLaunchKernelGGL ((mgMinimizerKernel<true, true>), dim3 (grid), dim3 (64 * MG_MIN_WAVES), lds, st, dPacked, dReadOffsets, nReads, p, w, dReadStart, dHash, dPosF, capacity, waveBytes, tile);
  else hipLaunchKernelGGL ((mgMinimizerKernel<true, false>), dim3 (grid), dim3 (64 * MG_MIN_WAVES), 0, st, dPacked, dReadOffsets, nReads, p, w, dReadStart, dHash, dPosF, capacity, 0u, 0u);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}
