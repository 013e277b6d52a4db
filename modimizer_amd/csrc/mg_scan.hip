/* mg_scan.hip — K1 (2-bit pack) and K2 (scan + select + ordered compaction) for gfx950.
 *
 * K2 restates, for a whole batch at once, what modRCiterator/modRCnext (reference
 * seqhash.c:154-196) produce read by read: for every k-mer start of every read the forward and
 * reverse-complement k-mers are hashed with the multiply-shift hash (seqhash.h:58), the smaller
 * hash decides the strand (ties -> reverse, seqhash.c:60-68), and the k-mer is kept when that hash
 * is 0 mod d.  Output order is (read, pos), exactly the order the reference's loops see.
 *
 * Work decomposition (MI355X-first, not one wave per read): the batch is one concatenated 2-bit
 * stream cut into tiles of 16384 k-mer starts; a persistent grid of 256-thread workgroups draws
 * tiles from an atomic ticket, so short reads (150 b) and chromosomes (125 Mb) load-balance
 * alike.  A tile's 4 KiB of packed bases are fetched with one 16-byte load per lane and staged in
 * LDS (+ a (k-1)-base halo); each lane owns 64 consecutive k-mer starts, rolls both strands through
 * registers and records hits in a 64-bit lane mask (no divergent work in the hot loop).  Hits are
 * compacted in order: lane popcounts -> workgroup scan -> decoupled look-back across tiles
 * (single pass; no recount) -> k-mers re-extracted from LDS by the few hit lanes and written out.
 */
#include "mg_common.h"

/* ---------------------------------------------------------------------------------------- */
/* K1: bytes 0..3 -> 2-bit packed words (first base in the top bits)                          */

__global__ void mgPackKernel (const U8 *__restrict__ bases, U64 nBases, U32 *__restrict__ words, U64 nWords)
{
  U64 i = (U64) blockIdx.x * blockDim.x + threadIdx.x;
  U64 stride = (U64) gridDim.x * blockDim.x;
  const bool aligned = (((uintptr_t) bases) & 15) == 0;
  for ( ; i < nWords ; i += stride)
    { U64 b0 = i * 16;
      U32 w = 0;
      if (b0 + 16 <= nBases && aligned)
        { uint4 v = *reinterpret_cast<const uint4 *> (bases + b0);
          U32 q[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
          for (int j = 0 ; j < 4 ; ++j)
            { U32 x = q[j];
              U32 n = ((x & 3) << 6) | (((x >> 8) & 3) << 4) | (((x >> 16) & 3) << 2) | ((x >> 24) & 3);
              w |= n << (24 - 8 * j);
            }
        }
      else if (b0 < nBases)
        { for (int j = 0 ; j < 16 && b0 + j < nBases ; ++j)
            w |= (U32) (bases[b0 + j] & 3) << (30 - 2 * j);
        }
      words[i] = w;
    }
}

__global__ void mgUnpackKernel (const U32 *__restrict__ words, U64 nBases, U8 *__restrict__ bases)
{
  U64 i = (U64) blockIdx.x * blockDim.x + threadIdx.x;
  U64 stride = (U64) gridDim.x * blockDim.x;
  for ( ; i < nBases ; i += stride)
    bases[i] = (U8) ((words[i >> 4] >> (30 - 2 * (i & 15))) & 3);
}

MgStatus mgLaunchPack (const U8 *dBases, U64 nBases, U32 *dWords, hipStream_t st)
{
  U64 nWords = (U64) mgPackedWords (nBases);
  if (!nWords) return MG_OK;
  U64 blocks = (nWords + 255) / 256; if (blocks > 8192) blocks = 8192;
  MG_LAUNCH (MG_K_PACK, st, mgPackKernel, dim3 ((unsigned) blocks), dim3 (256), 0, st, dBases, nBases, dWords, nWords);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}

MgStatus mgLaunchUnpack (const U32 *dWords, U64 nBases, U8 *dBases, hipStream_t st)
{
  if (!nBases) return MG_OK;
  U64 blocks = (nBases + 255) / 256; if (blocks > 16384) blocks = 16384;
  MG_LAUNCH (MG_K_UNPACK, st, mgUnpackKernel, dim3 ((unsigned) blocks), dim3 (256), 0, st, dWords, nBases, dBases);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}

/* ---------------------------------------------------------------------------------------- */
/* K2 helper: read containing the first base of each tile                                     */

/* largest r in [lo,hi] with off[r] <= p  (off[lo] <= p is guaranteed by the callers) */
__device__ __forceinline__ U32 mgReadOf (const U64 *__restrict__ off, U32 lo, U32 hi, U64 p)
{
  while (lo < hi)
    { U32 mid = lo + (hi - lo + 1) / 2;
      if (off[mid] <= p) lo = mid; else hi = mid - 1;
    }
  return lo;
}

__global__ void mgTileFirstReadKernel (const U64 *__restrict__ readOff, U32 nReads, U64 nTiles,
                                       U32 *__restrict__ tileFirstRead)
{
  U64 t = (U64) blockIdx.x * blockDim.x + threadIdx.x;
  if (t > nTiles) return;
  if (t == nTiles) { tileFirstRead[t] = nReads - 1; return; }
  tileFirstRead[t] = mgReadOf (readOff, 0, nReads - 1, t * (U64) MG_TILE_BASES);
}

/* ---------------------------------------------------------------------------------------- */
/* K2: generic scan (any k in 1..31, any d >= 1)                                              */

struct MgScanArgs {
  MgHashParams p;
  const U32 *packed; U64 nWordsAlloc; U64 totalBases;
  const U64 *readOff; U32 nReads;
  const U32 *tileFirstRead; U64 nTiles;
  U64 *desc; U32 *ticket;
  U64 *outKmer; U32 *outPosF; U32 *outRead; U64 capacity;
  U64 *dCount;
};

/* which of this lane's 64 k-mer starts lie wholly inside a read (seqhash.c:162: len < k gives
 * nothing; the last start of a read is len-k) */
__device__ __forceinline__ U64 mgValidMask (const MgScanArgs &a, U64 p0, U32 rFirst)
{
  U64 valid = 0;
  const int k = a.p.k;
  U32 r = rFirst;
  for (;;)
    { int64_t start = (int64_t) a.readOff[r], end = (int64_t) a.readOff[r + 1];
      int64_t lo = start > (int64_t) p0 ? start : (int64_t) p0;
      int64_t hi = end - k; if (hi > (int64_t) p0 + 63) hi = (int64_t) p0 + 63;
      if (hi >= lo)
        { int n = (int) (hi - lo + 1);
          U64 m = (n == 64) ? ~0ull : (((1ull << n) - 1) << (int) (lo - (int64_t) p0));
          valid |= m;
        }
      if (end >= (int64_t) p0 + 64 || r + 1 >= a.nReads) break;
      ++r;
    }
  return valid;
}

template <bool POW2>
__global__ __launch_bounds__ (MG_SCAN_THREADS)
void mgScanKernel (const MgScanArgs a)
{
  __shared__ __attribute__ ((aligned (16))) U32 sWords[MG_TILE_WORDS + 8];
  __shared__ U32 sWaveTot[MG_SCAN_THREADS / 64];
  __shared__ U64 sBase;
  __shared__ U32 sTile;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const MgHashParams &p = a.p;
  const int k = p.k, sh1 = p.shift1;
  const U64 f1 = p.factor1;

  for (;;)
    { if (tid == 0) sTile = atomicAdd (a.ticket, 1u);
      __syncthreads ();
      const U64 tile = sTile;
      if (tile >= a.nTiles) break;

      /* ---- stage the tile's packed bases (+ halo) in LDS: one 16-byte load per lane ---- */
      const U64 w0 = tile * MG_TILE_WORDS;
      { U64 g = w0 + 4 * (U64) tid;
        uint4 v;
        if (g + 4 <= a.nWordsAlloc) v = *reinterpret_cast<const uint4 *> (a.packed + g);
        else
          { v.x = g     < a.nWordsAlloc ? a.packed[g]     : 0;
            v.y = g + 1 < a.nWordsAlloc ? a.packed[g + 1] : 0;
            v.z = g + 2 < a.nWordsAlloc ? a.packed[g + 2] : 0;
            v.w = 0;
          }
        *reinterpret_cast<uint4 *> (&sWords[4 * tid]) = v;
        if (tid < 8)
          { U64 gh = w0 + MG_TILE_WORDS + tid;
            sWords[MG_TILE_WORDS + tid] = gh < a.nWordsAlloc ? a.packed[gh] : 0;
          }
      }
      __syncthreads ();

      /* ---- this lane's 64 starts: which are inside a read ---- */
      const U64 p0 = tile * (U64) MG_TILE_BASES + (U64) tid * MG_POS_PER_THREAD;
      U64 valid = 0;
      U32 rFirst = 0;
      if (p0 < a.totalBases)
        { rFirst = mgReadOf (a.readOff, a.tileFirstRead[tile], a.tileFirstRead[tile + 1], p0);
          valid = mgValidMask (a, p0, rFirst);
        }

      /* ---- roll both strands over the lane's 64 starts ---- */
      U32 w[6];
      { uint4 q = *reinterpret_cast<const uint4 *> (&sWords[4 * tid]);
        uint2 h = *reinterpret_cast<const uint2 *> (&sWords[4 * tid + 4]);
        w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w; w[4] = h.x; w[5] = h.y;
      }
      /* incoming-base stream: in[j] holds bases k+16j .. k+16j+15 of the lane's window */
      U32 in[4];
      { const bool up = (2 * k) >= 32;
        const int r = (2 * k) & 31;
        U32 s0 = up ? w[1] : w[0], s1 = up ? w[2] : w[1], s2 = up ? w[3] : w[2],
            s3 = up ? w[4] : w[3], s4 = up ? w[5] : w[4];
        in[0] = __funnelshift_l (s1, s0, r); in[1] = __funnelshift_l (s2, s1, r);
        in[2] = __funnelshift_l (s3, s2, r); in[3] = __funnelshift_l (s4, s3, r);
      }
      U64 F = (((U64) w[0] << 32) | w[1]) >> sh1;
      U64 R = mgRevComp (F, sh1);
      const int top = 2 * (k - 1);
      /* 4 chunks of 16 starts; the incoming-base words rotate through in[0] so that the chunk
         loop stays rolled (a fully unrolled 64-step body costs 256 VGPRs and all occupancy) */
      U32 accH = 0, accF = 0, hitLo = 0, fwdLo = 0;
#pragma unroll 1
      for (int chunk = 0 ; chunk < 4 ; ++chunk)
        { const U32 cur = in[0];
          in[0] = in[1]; in[1] = in[2]; in[2] = in[3];
#pragma unroll
          for (int tt = 0 ; tt < 16 ; ++tt)
            { U64 hF = (F * f1) >> sh1, hR = (R * f1) >> sh1;
              bool fwd = hF < hR;
              U64 h = fwd ? hF : hR;
              bool hit;
              if (POW2) hit = (h & (U64) (p.d - 1)) == 0;
              else      hit = mgDivisible (h, p);
              accH = (accH << 1) | (hit ? 1u : 0u);
              accF = (accF << 1) | (fwd ? 1u : 0u);
              U32 b = (cur >> (30 - 2 * tt)) & 3;         /* base k + 16*chunk + tt enters */
              F = ((F << 2) & p.mask) | b;
              R = (R >> 2) | ((U64) (3 - b) << top);
            }
          if (chunk == 1) { hitLo = accH; fwdLo = accF; accH = accF = 0; }
        }
      /* accumulators hold start t at bit 31-(t&31): reverse to natural order */
      U64 hits = ((U64) __brev (accH) << 32) | __brev (hitLo);
      U64 fwds = ((U64) __brev (accF) << 32) | __brev (fwdLo);
      hits &= valid;

      /* ---- ordered compaction: lane counts -> workgroup scan -> look-back across tiles ---- */
      U32 cnt = (U32) __popcll (hits);
      U32 incl = cnt;
#pragma unroll
      for (int off = 1 ; off < 64 ; off <<= 1)
        { U32 v = __shfl_up (incl, off); if (lane >= off) incl += v; }
      if (lane == 63) sWaveTot[wave] = incl;
      __syncthreads ();
      U32 waveBase = 0, total = 0;
#pragma unroll
      for (int i = 0 ; i < MG_SCAN_THREADS / 64 ; ++i)
        { U32 v = sWaveTot[i]; if (i < wave) waveBase += v; total += v; }
      if (wave == 0)
        { U64 b = mgLookback (a.desc, tile, total);
          if (lane == 0)
            { sBase = b;
              if (tile == a.nTiles - 1) a.dCount[0] = b + total;
              if (b + total > a.capacity) a.dCount[1] = 1;
            }
        }
      __syncthreads ();
      U64 o = sBase + waveBase + (incl - cnt);

      /* ---- the (few) hit lanes re-extract their k-mers from LDS and write them out ---- */
      U32 r = rFirst;
      while (hits)
        { int t = __ffsll ((long long) hits) - 1;
          hits &= hits - 1;
          int wi = 4 * tid + (t >> 4), s = 2 * (t & 15);
          U32 x0 = sWords[wi], x1 = sWords[wi + 1], x2 = sWords[wi + 2];
          U64 hi = ((U64) x0 << 32) | x1;
          if (s) hi = (hi << s) | (U64) (x2 >> (32 - s));
          U64 Fk = hi >> sh1;
          bool fwd = (fwds >> t) & 1;
          U64 kmer = fwd ? Fk : mgRevComp (Fk, sh1);
          U64 pos = p0 + (U64) t;
          while (a.readOff[r + 1] <= pos) ++r;
          if (o < a.capacity)
            { a.outKmer[o] = kmer;
              a.outPosF[o] = (U32) (pos - a.readOff[r]) | (fwd ? MG_FWD_BIT : 0u);
              if (a.outRead) a.outRead[o] = r;
            }
          ++o;
        }
      __syncthreads ();     /* sWords / sTile are rewritten by the next tile */
    }
}

/* ---------------------------------------------------------------------------------------- */

static inline U64 mgNumTiles (U64 totalBases) { return (totalBases + MG_TILE_BASES - 1) / MG_TILE_BASES; }

/* work buffer layout: [0,256) ticket + pad | desc[nTiles] | tileFirstRead[nTiles+1] */
size_t mgScanWorkBytes (U64 totalBases, U32 nReads)
{
  (void) nReads;
  U64 nTiles = mgNumTiles (totalBases);
  size_t b = 256 + (size_t) nTiles * 8 + ((size_t) nTiles + 2) * 4;
  return (b + 255) & ~(size_t) 255;
}

MgStatus mgLaunchScan (const MgHashParams &p, const U32 *dPacked, U64 totalBases,
                       const U64 *dReadOffsets, U32 nReads,
                       U64 *dKmer, U32 *dPosF, U32 *dReadId, U64 capacity,
                       U64 *dCount, void *dWork, hipStream_t st)
{
  MG_HIP (hipMemsetAsync (dCount, 0, 2 * sizeof (U64), st));
  U64 nTiles = mgNumTiles (totalBases);
  if (!nTiles || !nReads) return MG_OK;
  char *wb = (char *) dWork;
  U32 *ticket = (U32 *) wb;
  U64 *desc = (U64 *) (wb + 256);
  U32 *tfr = (U32 *) (wb + 256 + nTiles * 8);
  MG_HIP (hipMemsetAsync (wb, 0, 256 + nTiles * 8, st));
  MG_LAUNCH (MG_K_TILE_FIRST_READ, st, mgTileFirstReadKernel, dim3 ((unsigned) ((nTiles + 1 + 255) / 256)), dim3 (256), 0, st,
                      dReadOffsets, nReads, nTiles, tfr);
  MG_HIP (hipGetLastError ());

  MgScanArgs a;
  a.p = p; a.packed = dPacked; a.nWordsAlloc = (U64) mgPackedWords (totalBases); a.totalBases = totalBases;
  a.readOff = dReadOffsets; a.nReads = nReads; a.tileFirstRead = tfr; a.nTiles = nTiles;
  a.desc = desc; a.ticket = ticket;
  a.outKmer = dKmer; a.outPosF = dPosF; a.outRead = dReadId; a.capacity = capacity; a.dCount = dCount;
  unsigned grid = (unsigned) (nTiles < 2048 ? nTiles : 2048);
  if (p.dOddInv == 1 && p.dOddLim == ~0ull)
    MG_LAUNCH (MG_K_SCAN, st, mgScanKernel<true>, dim3 (grid), dim3 (MG_SCAN_THREADS), 0, st, a);
  else
    MG_LAUNCH (MG_K_SCAN, st, mgScanKernel<false>, dim3 (grid), dim3 (MG_SCAN_THREADS), 0, st, a);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}
