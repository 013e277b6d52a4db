/* mg_scan.hip — K1 (2-bit pack) and K2 (scan + select + ordered compaction) for gfx950.
 *
 * K2 restates, for a whole batch at once, what modRCiterator/modRCnext (reference
 * seqhash.c:154-196) produce read by read: for every k-mer start of every read the forward and
 * reverse-complement k-mers are hashed with the multiply-shift hash (seqhash.h:58), the smaller
 * hash decides the strand (ties -> reverse, seqhash.c:60-68), and the k-mer is kept when that hash
 * is 0 mod d.  Output order is (read, pos), exactly the order the reference's loops see.
 *
 * Work decomposition (MI355X-first, not one wave per read): the batch is one concatenated 2-bit
 * stream cut into tiles of 4096 k-mer starts (MG_TILE_BASES: 64 lanes x 64 starts, 1 KiB of packed bases),
 * so short reads (150 b) and chromosomes (125 Mb) load-balance alike.  Every WAVEFRONT is an independent
 * worker: it owns a contiguous range of tiles and appends the modimizers it finds, in order, to its own
 * segment of a staging buffer; nothing crosses a wavefront (no workgroup barrier, no shared counter, no
 * inter-workgroup protocol: a first version handed tiles out by an atomic ticket and ordered the output
 * by decoupled look-back; at > 50 M tiles/s both the single ticket word and the descriptor polling
 * saturated and capped the kernel near 0.8 Tbp/s).  The per-worker counts are scanned by a one-block
 * kernel; the consumers of a large batch read the segments where they are (the modset build's first
 * partition pass and index assignment, the query's lookups: MgSegSrc), and a streaming compaction kernel
 * makes dense (read,pos)-ordered arrays where those are wanted (pos / read of the query path, small
 * batches, the scan API).  On the way the kernel counts every modimizer's first partition digit for the
 * modset build (mgMixTopOfKmer).  A tile's 1 KiB is fetched one tile ahead with one 16-byte load per lane (registers) and staged,
 * double-buffered, in the wavefront's LDS with a (k-1)-base halo; each lane owns 64 consecutive k-mer starts.
 */
#include <stdlib.h>
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
/* K2 prepass: per-tile read metadata                                                         */

/* largest r in [lo,hi] with off[r] <= p  (off[lo] <= p is guaranteed by the callers) */
__device__ __forceinline__ U32 mgReadOf (const U64 *__restrict__ off, U32 lo, U32 hi, U64 p)
{
  while (lo < hi)
    { U32 mid = lo + (hi - lo + 1) / 2;
      if (off[mid] <= p) lo = mid; else hi = mid - 1;
    }
  return lo;
}

/* 32 bytes per tile: the read holding the tile's first base, and that read's extent.  A tile that
 * lies wholly inside one read (the common case for long reads) needs no other read lookup. */
struct __attribute__ ((aligned (32))) MgTileInfo { U64 start, end; U32 firstRead, pad0; U64 pad1; };

__global__ void mgTileInfoKernel (const U64 *__restrict__ readOff, U32 nReads, U64 nTiles, U64 totalBases,
                                  MgTileInfo *__restrict__ info)
{
  U64 t = (U64) blockIdx.x * blockDim.x + threadIdx.x;
  if (t > nTiles) return;
  MgTileInfo ti;
  ti.pad0 = 0; ti.pad1 = 0;
  if (t == nTiles) { ti.firstRead = nReads - 1; ti.start = readOff[nReads - 1]; ti.end = totalBases; }
  else
    { U32 r = mgReadOf (readOff, 0, nReads - 1, t * (U64) MG_TILE_BASES);
      ti.firstRead = r; ti.start = readOff[r]; ti.end = readOff[r + 1];
    }
  info[t] = ti;
}

/* ---------------------------------------------------------------------------------------- */
/* K2: scan + select + ordered compaction                                                     */

struct MgScanArgs {
  MgHashParams p;
  const U32 *packed; U64 nWordsAlloc; U64 totalBases;
  const U64 *readOff; U32 nReads;
  const MgTileInfo *tileInfo;
  U64 tileBegin, tileLimit;      /* this launch covers tiles [tileBegin, tileLimit) */
  U64 tilesPerWorker;    /* worker (wavefront) v owns tiles [v*tilesPerWorker, (v+1)*tilesPerWorker) */
  U64 nWorkers;
  U64 segCap;            /* entries per output segment */
  U64 *segKmer; U32 *segPosF; U32 *segRead;        /* [nWorkers * segCap] */
  U64 *blockCount;       /* [nWorkers] true number of modimizers each worker found */
  U32 fS, thresh;        /* fast path: factor1 << (32-B), 2^(32-m) */
  U32 *histCount;        /* != 0: count the modimizers per coarse digit of their table bucket (what the first partition pass of the modset build sorts by) */
  int histKbits, histHiB;   /* the digit = top histHiB bits of mgMixK (kmer, histKbits); histKbits >= 24 */
#ifdef MG_ABLATE
  U32 debug;             /* ablation builds only (tools/ablate_scan.sh): bit0 = stop after phase B, bit1 = no stores, bit2 = no evaluation */
#endif
};

/* which of this lane's 64 k-mer starts lie wholly inside a read (seqhash.c:162: len < k gives
 * nothing; the last start of a read is len-k).  off[r - rBase] is the start of read r: the global offsets
 * (rBase 0) or the tile's copy of them in LDS. */
__device__ __forceinline__ U64 mgValidMask (const U64 *off, U32 rBase, U32 nReads, int k, U64 p0, U32 rFirst)
{
  U64 valid = 0;
  U32 r = rFirst;
  for (;;)
    { int64_t start = (int64_t) off[r - rBase], end = (int64_t) off[r + 1 - rBase];
      int64_t lo = start > (int64_t) p0 ? start : (int64_t) p0;
      int64_t hi = end - k; if (hi > (int64_t) p0 + 63) hi = (int64_t) p0 + 63;
      if (hi >= lo)
        { int n = (int) (hi - lo + 1);
          U64 m = (n == 64) ? ~0ull : (((1ull << n) - 1) << (int) (lo - (int64_t) p0));
          valid |= m;
        }
      if (end >= (int64_t) p0 + 64 || r + 1 >= nReads) break;
      ++r;
    }
  return valid;
}

/* a tile's 1 KiB: one 16-byte load per lane.  Only a batch's last tiles can reach past the allocation (the stream is followed
 * by MG_PACK_PAD zero words, a tile with its halo reads MG_TILE_WORDS + 8): `inside` (uniform) says this one does not. */
__device__ __forceinline__ bool mgTileInside (const MgScanArgs &a, U64 tile) { return (tile + 1) * MG_TILE_WORDS + 8 <= a.nWordsAlloc; }
__device__ __forceinline__ uint4 mgLoadTileWords (const MgScanArgs &a, U64 tile, int tid, bool inside)
{
  U64 g = tile * MG_TILE_WORDS + 4 * (U64) tid;
  uint4 v;
  if (inside || g + 4 <= a.nWordsAlloc) v = *reinterpret_cast<const uint4 *> (a.packed + g);
  else
    { v.x = g     < a.nWordsAlloc ? a.packed[g]     : 0;
      v.y = g + 1 < a.nWordsAlloc ? a.packed[g + 1] : 0;
      v.z = g + 2 < a.nWordsAlloc ? a.packed[g + 2] : 0;
      v.w = 0;
    }
  return v;
}
__device__ __forceinline__ U32 mgLoadTileHalo (const MgScanArgs &a, U64 tile, int lane, bool inside)
{
  U32 h = 0;
  if (lane < 8) { const U64 gh = tile * MG_TILE_WORDS + MG_TILE_WORDS + lane; h = (inside || gh < a.nWordsAlloc) ? a.packed[gh] : 0; }
  return h;
}

/* k-mer (forward strand) starting at position q of the tile staged in sWords */
__device__ __forceinline__ U64 mgKmerAt (const U32 *sWords, U32 q, int sh1)
{
  int wi = q >> 4, s = 2 * (q & 15);
  U32 x0 = sWords[wi], x1 = sWords[wi + 1], x2 = sWords[wi + 2];
  U64 hi = ((U64) x0 << 32) | x1;
  if (s) hi = (hi << s) | (U64) (x2 >> (32 - s));
  return hi >> sh1;
}

#define MG_MODE_ANY   0      /* any d: exact test in phase A via modular inverse */
#define MG_MODE_POW2  1      /* d = 2^m: exact test in phase A via a mask */
#define MG_MODE_FAST  2      /* d = 2^m, shift1+m <= 32, k >= 17: low-bits filter in phase A */
#define MG_MODE_ODD   3      /* odd d (the reference's default w = 31): the exact test is the inverse's product alone */
#define MG_MODE_ODD32 4      /* odd d < 2^15 and k <= 20 (the reference's defaults k = 19, w = 31): that test in 32-bit arithmetic (mgDivisibleOdd32) */
#define MG_MODE_ANY32 5      /* d = odd 2^s, d < 2^15, k <= 20: low s bits zero and the 32-bit test on the rest */
#define MG_LIST_UNROLL 4     /* candidates per half of a lane's mask listed by straight-line code */
#define MG_CAND_CAP   320    /* candidate list entries (LDS, per wavefront): up to 63 waiting from the tile before + a pass of this tile's */
#define MG_WAVES      (MG_SCAN_THREADS / 64)
#ifdef MG_ABLATE
#define MG_ABLATE_AND(x) && (x)
#else
#define MG_ABLATE_AND(x)
#endif

/* LDS written by some lanes of a wavefront and read by others of the same wavefront: DS operations of a
 * wave complete in issue order, so all that is needed is that the compiler keeps the order */
#define MG_WAVE_SYNC() do { __builtin_amdgcn_fence (__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier (); \
                            __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "wavefront"); } while (0)

/* One templated kernel; every WAVEFRONT is an independent worker.
 *
 * A worker owns a contiguous range of tiles (4096 k-mer starts each: 64 lanes x 64 starts) and its own
 * output segment, so nothing in the kernel crosses a wavefront: no workgroup barrier, no shared counters
 * (a workgroup is just four workers sharing a CU).  Per tile:
 *
 * Phase A  every lane owns 64 consecutive k-mer starts and produces a 64-bit candidate mask with
 *          no divergent work:
 *            ANY/POW2: both strands rolled through registers, both 64-bit multiply-shift hashes,
 *                      canonical = smaller hash, exact "0 mod d" test (candidates == modimizers);
 *            FAST:     hash % 2^m tests bits [shift1, shift1+m) of kmer*factor1.  With
 *                      B = shift1+m <= 32 those bits depend only on (kmer mod 2^32), i.e. on the
 *                      last 16 bases of the forward k-mer and the first 16 bases, reverse-
 *                      complemented, of the reverse one.  With fS = factor1 << (32-B):
 *                      candidate  <=>  min (winF*fS, winR*fS) mod 2^32 < 2^(32-m)
 *                      (one v_alignbit + one 32-bit multiply per strand, a min, a compare into VCC and an
 *                      add-with-carry that shifts the bit into the mask): a superset of the modimizers,
 *                      about 2/d of the starts.
 * Phase B  a DPP prefix sum over the lanes' candidate counts orders the candidates; they are listed,
 *          in order, in the wavefront's LDS list (rounds of MG_CAND_CAP).
 * Phase C  dense: one lane per candidate recomputes both full hashes from the tile in LDS, picks the
 *          strand (ties -> reverse, seqhash.c:66-67) and applies the exact test; a ballot ranks the
 *          survivors, which go straight to the worker's segment: kmer / pos|isF / read.
 *
 * Latency: the next tile's words, halo and metadata are fetched while the current tile is processed
 * (registers), and the tile staging in LDS is double-buffered.
 */
/* a wave has made its last count: the workgroup's last wave adds the LDS counts to the global ones.  The tick on sDone is a
 * RELEASE at workgroup scope -- neither the compiler nor the hardware may let a wave's counts on sHist sink below it -- and the wave
 * that finds itself last makes an ACQUIRE before it reads sHist, so that what it reads are the other waves' finished counts.  (Rounds
 * 1 - 4 used a relaxed atomicAdd and leaned on the LDS unit taking a CU's operations in order: true of the hardware, a promise the
 * compiler never made.) */
__device__ __forceinline__ void mgScanHistDone (const MgScanArgs &a, U32 *sHist, U32 *sDone, int lane)
{
  U32 before = 0;
  if (lane == 0) before = __hip_atomic_fetch_add (sDone, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  before = (U32) __builtin_amdgcn_readfirstlane ((int) before);
  if (before != MG_WAVES - 1) return;
  __builtin_amdgcn_fence (__ATOMIC_ACQUIRE, "workgroup");
  const U32 bins = (U32) 1 << a.histHiB;
  for (U32 b = (U32) lane ; b < bins ; b += 64) { const U32 v = sHist[b]; if (v) atomicAdd (&a.histCount[b * MG_HIST_STRIDE], v); }
}

/* the work of one worker (wavefront): its tiles, its segment; returns the number of modimizers it found (uniform).
 * SINGLE: the batch is one read [0, totalBases) (the per-read iterator facade): no per-tile metadata array, every tile's
 * first read is read 0. */
template <int MODE, bool SINGLE, bool WHERE>        /* WHERE: pos / read of the modimizers are wanted, not the k-mers alone (the modset build) */
__device__ __forceinline__ U64 mgScanWorker (const MgScanArgs &a, const U64 worker, const int lane,
                                             U32 (*sWordsW)[MG_TILE_WORDS + 8], unsigned short *sCand, U64 *sOff, U32 *sHist)
{
  const MgHashParams &p = a.p;
  const int k = p.k, sh1 = p.shift1;
  const U64 f1 = p.factor1;
  const U64 dMask = (U64) (p.d - 1);

  U64 tile = a.tileBegin + worker * a.tilesPerWorker;
  U64 tileEnd = tile + a.tilesPerWorker; if (tileEnd > a.tileLimit) tileEnd = a.tileLimit;
  const U64 segBase = worker * a.segCap;
  U32 found = 0;                                   /* modimizers this worker has found so far (uniform; a worker's tiles hold fewer than 2^32 starts) */
  const U32 segCap32 = a.segCap > 0xffffffffull ? 0xffffffffu : (U32) a.segCap;
  U64 *const segK = a.segKmer + segBase;
  uint4 curV = make_uint4 (0, 0, 0, 0); U32 curHalo = 0;
  MgTileInfo ti, tiNext;
  ti.start = ti.end = 0; ti.firstRead = 0; tiNext = ti;
  U32 nextFirstRead = 0;
  if (tile < tileEnd)
    { const bool inside = mgTileInside (a, tile);
      curV = mgLoadTileWords (a, tile, lane, inside);
      curHalo = mgLoadTileHalo (a, tile, lane, inside);
      if (SINGLE) { ti.start = 0; ti.end = a.totalBases; ti.firstRead = 0; }
      else { ti = a.tileInfo[tile]; nextFirstRead = a.tileInfo[tile + 1].firstRead; }
    }

  int buf = 0;
  U32 qCount = 0, qOld = 0;                          /* the candidate queue: entries waiting, and how many of them are of the previous tile (uniform) */
  while (tile < tileEnd)
    { U32 *sWords = sWordsW[buf]; buf ^= 1;
      *reinterpret_cast<uint4 *> (&sWords[4 * lane]) = curV;
      if (lane < 8) sWords[MG_TILE_WORDS + lane] = curHalo;

      const U64 tile0 = tile * (U64) MG_TILE_BASES;
      const U64 p0 = tile0 + (U64) lane * MG_POS_PER_THREAD;
      /* ---- which of the lane's 64 starts lie inside a read ---- */
      U64 valid = 0;
      U32 rFirst = ti.firstRead;
      const bool oneRead = ti.end >= tile0 + MG_TILE_BASES + (U64) k - 1;   /* tile + halo inside one read */
      /* a tile with read boundaries: the offsets of its reads (first read of this tile .. first read of the next
         tile, and one more) are fetched once, coalesced, into LDS, and the per-lane searches run there; only a
         tile with more than 62 reads (reads under ~66 bases) searches the global array */
      const bool offInLds = !oneRead && nextFirstRead - ti.firstRead + 2 <= 64;
      if (offInLds)
        { const U32 r = ti.firstRead + (U32) lane;
          if (SINGLE) sOff[lane] = lane == 0 ? 0ull : (lane == 1 ? a.totalBases : ~0ull);      /* one read [0, totalBases): nothing to fetch (for the iterator the offsets would come over PCIe) */
          else sOff[lane] = r <= a.nReads ? a.readOff[r] : ~0ull;
        }
      MG_WAVE_SYNC ();                                                           /* tile (and offsets) staged */
      if (oneRead) valid = ~0ull;
      else if (p0 < a.totalBases)
        { if (offInLds)
            { rFirst = ti.firstRead + mgReadOf (sOff, 0, nextFirstRead - ti.firstRead, p0);
              valid = mgValidMask (sOff, ti.firstRead, a.nReads, k, p0, rFirst);
            }
          else
            { rFirst = mgReadOf (a.readOff, ti.firstRead, nextFirstRead, p0);
              valid = mgValidMask (a.readOff, 0, a.nReads, k, p0, rFirst);
            }
        }

      /* prefetch the next tile (registers; consumed at the top of the next iteration) */
      const U64 nextTile = tile + 1;
      uint4 nextV = make_uint4 (0, 0, 0, 0); U32 nextHalo = 0; U32 nextNextFirst = 0;
      if (nextTile < tileEnd)
        { const bool inside = mgTileInside (a, nextTile);
          nextV = mgLoadTileWords (a, nextTile, lane, inside);
          nextHalo = mgLoadTileHalo (a, nextTile, lane, inside);
          if (SINGLE) tiNext = ti;
          else { tiNext = a.tileInfo[nextTile]; nextNextFirst = a.tileInfo[nextTile + 1].firstRead; }
        }

      /* ---- Phase A ---- */
      U32 w[6];
      { uint2 h = *reinterpret_cast<const uint2 *> (&sWords[4 * lane + 4]);
        w[0] = curV.x; w[1] = curV.y; w[2] = curV.z; w[3] = curV.w; w[4] = h.x; w[5] = h.y;
      }
      U64 cand;
      if (MODE == MG_MODE_FAST)
        { /* forward: last 16 bases of the k-mer at start t = bases [t+k-16, t+k): stream shifted by
             k-16 bases; reverse: first 16 bases [t, t+16), reverse-complemented word by word */
          U32 fw[6], rw[6];
          const int c2 = 2 * (k - 16);                         /* 2..30 */
#pragma unroll
          for (int j = 0 ; j < 5 ; ++j) fw[j] = __builtin_amdgcn_alignbit (w[j], w[j + 1], 32 - c2);   /* = funnelshift_l (w[j + 1], w[j], c2): c2 is never 0 here, so one v_alignbit */
          fw[5] = w[5] << c2;
#pragma unroll
          for (int j = 0 ; j < 6 ; ++j) rw[j] = mgRevComp16 (w[j]);
          const U32 fS = a.fS, thresh = a.thresh;
          U32 acc = 0, candLo = 0;
#pragma unroll
          for (int chunk = 0 ; chunk < 4 ; ++chunk)
            { const U32 f0 = fw[0], f1w = fw[1], r0 = rw[0], r1 = rw[1];
              fw[0] = fw[1]; fw[1] = fw[2]; fw[2] = fw[3]; fw[3] = fw[4]; fw[4] = fw[5];
              rw[0] = rw[1]; rw[1] = rw[2]; rw[2] = rw[3]; rw[3] = rw[4]; rw[4] = rw[5];
#pragma unroll
              for (int tt = 0 ; tt < 16 ; ++tt)
                { U32 xf = __funnelshift_l (f1w, f0, 2 * tt);     /* window, first base on top */
                  U32 xr = __funnelshift_r (r0, r1, 2 * tt);      /* reverse-complemented window */
                  U32 hf = xf * fS, hr = xr * fS;
                  U32 mn = hf < hr ? hf : hr;
                  /* acc = 2*acc + (mn < thresh): compare into vcc, add-with-carry of acc to itself */
                  asm ("v_cmp_gt_u32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc"
                       : "+v" (acc) : "s" (thresh), "v" (mn) : "vcc");
                }
              if (chunk == 1) { candLo = acc; acc = 0; }
            }
          cand = ((U64) __brev (acc) << 32) | __brev (candLo);
        }
      else
        { /* exact modes: both k-mers of every start straight out of the lane's six words (96 bases: its 64 starts and
             the k - 1 <= 30 bases after them), two v_alignbit each -- no rolled k-mer, no chain from start to start.
             Forward: the 64 bits from the start's first base, shifted down to the k-mer (F = W >> shift1).  Reverse: word-wise
             reverse complements make a little-endian stream X with base n at bits [2n, 2n + 2); R = (X >> 2 start) & mask.
             (Round 4: the rolled form -- F = ((F << 2) & mask) | b, R = (R >> 2) | (3 - b) << 2(k-1) -- cost 11 instructions a
             start against 7 here; with the test by the odd part's inverse alone where d is odd, 40.7 -> VALU per start.) */
          U32 rw[6];
#pragma unroll
          for (int j = 0 ; j < 6 ; ++j) rw[j] = mgRevComp16 (w[j]);
          const U32 maskLo = (U32) p.mask, maskHi = (U32) (p.mask >> 32);
          U32 acc = 0, hitLo = 0;
#pragma unroll 1
          for (int chunk = 0 ; chunk < 4 ; ++chunk)
            { const U32 f0 = w[0], f1w = w[1], f2 = w[2], r0 = rw[0], r1 = rw[1], r2 = rw[2];
              w[0] = w[1]; w[1] = w[2]; w[2] = w[3]; w[3] = w[4]; w[4] = w[5];
              rw[0] = rw[1]; rw[1] = rw[2]; rw[2] = rw[3]; rw[3] = rw[4]; rw[4] = rw[5];
#pragma unroll
              for (int tt = 0 ; tt < 16 ; ++tt)
                { const U32 wHi = tt ? __builtin_amdgcn_alignbit (f0, f1w, 32 - 2 * tt) : f0;
                  const U32 wLo = tt ? __builtin_amdgcn_alignbit (f1w, f2, 32 - 2 * tt) : f1w;
                  const U32 xLo = tt ? __builtin_amdgcn_alignbit (r1, r0, 2 * tt) : r0;
                  const U32 xHi = tt ? __builtin_amdgcn_alignbit (r2, r1, 2 * tt) : r1;
                  const U64 F = (((U64) wHi << 32) | wLo) >> sh1;
                  const U64 R = ((U64) (xHi & maskHi) << 32) | (xLo & maskLo);
                  U64 hF = (F * f1) >> sh1, hR = (R * f1) >> sh1;
                  U64 h = hF < hR ? hF : hR;
                  bool hit;
                  if (MODE == MG_MODE_POW2)     hit = (h & dMask) == 0;
                  else if (MODE == MG_MODE_ODD) hit = h * p.dOddInv <= p.dOddLim;
                  else if (MODE == MG_MODE_ODD32) hit = mgDivisibleOdd32 (h, p);
                  else if (MODE == MG_MODE_ANY32) hit = !((U32) h & (((U32) 1 << p.dShift) - 1u)) && mgDivisibleOdd32 (h >> p.dShift, p);
                  else                          hit = mgDivisible (h, p);
                  acc = (acc << 1) | (hit ? 1u : 0u);
                }
              if (chunk == 1) { hitLo = acc; acc = 0; }
            }
          cand = ((U64) __brev (acc) << 32) | __brev (hitLo);
        }
      cand &= valid;

      /* ---- Phase B: order the candidates ---- */
      const U32 cntLo = (U32) __popc ((U32) cand);
      const U32 cnt = cntLo + (U32) __popc ((U32) (cand >> 32));
      const U32 incl = mgWaveInclusiveSum (cnt);
      const U32 nc = (U32) __builtin_amdgcn_readlane ((int) incl, 63);
      const U32 myFirst = incl - cnt;                        /* ordinal of this lane's first candidate */
#ifdef MG_ABLATE
      if (a.debug & 1)                                      /* phases A+B only (output meaningless) */
        { tile = nextTile; curV = nextV; curHalo = nextHalo; ti = tiNext; nextFirstRead = nextNextFirst;
          found += nc >> 20;
          continue;
        }
#endif
      /* The candidate list is a queue (sCand[0 .. qCount)): a tile's candidates are appended, in order, and evaluated 64 at
         a time from the front.  When only the k-mers are wanted (the modset build) what is left of a tile, fewer than 64,
         waits for the next tile's candidates instead of costing a round of its own with most lanes idle -- an entry says which
         of the two tile buffers it points into, and the previous tile's is intact until the tile after this one is staged; so
         entries of the previous tile (qOld of them, at the front) must be gone when this tile is done.  A tile with more
         candidates than the list holds (small d) lists them in passes. */
      constexpr bool wantWhere = WHERE;
      constexpr int LIST_UNROLL = WHERE ? 0 : MG_LIST_UNROLL;    /* (with pos / read wanted the kernel has no scalar registers to spare for the straight-line form's lane masks) */
      const bool lastTile = nextTile >= tileEnd;
      const U32 bufTag = (U32) (buf ^ 1) << 12;                  /* the buffer this tile was staged in */
      U32 listed = 0;                                            /* candidates of this tile in the list or done with */
      do
        { const U32 room = MG_CAND_CAP - qCount;
          const U32 take = nc - listed < room ? nc - listed : room;
          const U32 lo = listed, hi = listed + take;             /* this pass lists the tile's candidates lo .. hi-1 */
          /* the lane's candidates, low half of the mask then high half: 32-bit bit tricks, one LDS store each */
          if (lo == 0 && hi == nc)                               /* the usual case: every candidate has its slot, nothing to clamp */
            { unsigned short *pp = sCand + qCount + myFirst;
#pragma unroll
              for (int half = 0 ; half < 2 ; ++half)
                { U32 c = half ? (U32) (cand >> 32) : (U32) cand;
                  const U32 base = bufTag | ((U32) lane * MG_POS_PER_THREAD + 32u * half);
                  /* the first few candidates of the half as straight-line code -- constant store offsets, no pointer to
                     carry round a loop: the fullest of 64 lanes holds 4 or 5 -- then the rare rest */
#pragma unroll
                  for (int j = 0 ; j < LIST_UNROLL ; ++j)
                    { if (!c) break;
                      pp[j] = (unsigned short) (base | (U32) __builtin_ctz (c));
                      c &= c - 1;
                    }
                  if (c)
                    { unsigned short *qq = pp + LIST_UNROLL;
                      do { *qq++ = (unsigned short) (base | (U32) __builtin_ctz (c)); c &= c - 1; } while (c);
                    }
                  if (!half) pp += cntLo;
                }
            }
          else if (take && myFirst < hi && myFirst + cnt > lo)
            { U32 o = myFirst - lo;                              /* place in this pass; wraps below lo, so "o < take" alone selects the pass's entries */
#pragma unroll
              for (int half = 0 ; half < 2 ; ++half)
                { U32 c = half ? (U32) (cand >> 32) : (U32) cand;
                  const U32 base = bufTag | ((U32) lane * MG_POS_PER_THREAD + 32u * half);
                  while (c)
                    { const U32 t = (U32) __builtin_ctz (c);
                      c &= c - 1;
                      sCand[o < take ? qCount + o : MG_CAND_CAP] = (unsigned short) (base | t);   /* [MG_CAND_CAP]: where other passes' entries land; no branch in the loop */
                      ++o;
                    }
                }
            }
          MG_WAVE_SYNC ();                                                       /* candidates listed */
          qCount += take; listed = hi;
          U32 nEval = qCount & ~63u;                                             /* whole rounds ... */
          if (listed == nc && (wantWhere || lastTile || qOld > nEval)) nEval = qCount;   /* ... or everything: pos / read come from this tile's state; nothing may outlive its tile buffer */
          U32 waveRun = 0;
#pragma unroll 1
          for (U32 i0 = 0 ; i0 < nEval ; i0 += 64)
            { const U32 i = i0 + (U32) lane;
              /* every lane evaluates what its list slot holds (a stale entry in the slots past the queue's end: any
                 position of either tile buffer is readable) and is masked afterwards: no divergent region, no values to merge */
              const U32 e = sCand[i];
              const U32 q = e & (MG_TILE_BASES - 1);
              U64 F = mgKmerAt (sWordsW[(e >> 12) & 1], q, sh1);
              bool surv, fwd;
              { const U64 R = mgRevComp (F, sh1);
                if (MODE == MG_MODE_FAST)
                  { /* hashes compared and tested where they sit in the products: no 64-bit shifts.  hash = P >> sh1,
                       so hF < hR <=> (PF with its low sh1 bits cleared) < (PR likewise), and hash % 2^m == 0 <=>
                       bits [sh1, sh1+m) of P are zero -- all inside the low word because sh1 + m <= 32 */
                    const U64 PF = F * f1, PR = R * f1;
                    const U32 keep = ~((1u << sh1) - 1u);
                    const U64 cF = (PF & 0xffffffff00000000ull) | ((U32) PF & keep), cR = (PR & 0xffffffff00000000ull) | ((U32) PR & keep);
                    fwd = cF < cR;
                    const U32 lowWord = (U32) (fwd ? PF : PR);
                    surv = (lowWord & ((U32) dMask << sh1)) == 0;
                  }
                else
                  { U64 hF = (F * f1) >> sh1, hR = (R * f1) >> sh1;
                    fwd = hF < hR;
                    U64 h = fwd ? hF : hR;
                    if (MODE == MG_MODE_POW2)     surv = (h & dMask) == 0;
                    else if (MODE == MG_MODE_ODD) surv = h * p.dOddInv <= p.dOddLim;
                    else if (MODE == MG_MODE_ODD32) surv = mgDivisibleOdd32 (h, p);
                    else if (MODE == MG_MODE_ANY32) surv = !((U32) h & (((U32) 1 << p.dShift) - 1u)) && mgDivisibleOdd32 (h >> p.dShift, p);
                    else                          surv = mgDivisible (h, p);
                  }
                if (!fwd) F = R;
              }
              surv = surv && i < nEval MG_ABLATE_AND (!(a.debug & 4));
              const U64 bs = __ballot (surv);
              U32 r = 0;
              if (wantWhere && (!oneRead || a.segRead)) r = (U32) __shfl ((int) rFirst, (int) (q >> 6));   /* first read of the owner lane's starts */
              if (surv)
                { const U32 o = found + waveRun + (U32) __popcll (bs & (((U64) 1 << lane) - 1));
                  const U64 pos = tile0 + q;                               /* (with pos / read wanted every entry is of this tile) */
                  U64 rs = ti.start;
                  if (!oneRead && wantWhere)
                    { if (offInLds) { while (sOff[r + 1 - ti.firstRead] <= pos) ++r; rs = sOff[r - ti.firstRead]; }
                      else { while (a.readOff[r + 1] <= pos) ++r; rs = a.readOff[r]; }
                    }
                  if (a.histCount) atomicAdd (&sHist[mgMixTopOfKmer (F, a.histKbits, a.histHiB)], 1u);
                  if (o < segCap32 MG_ABLATE_AND (!(a.debug & 2)))
                    { segK[o] = F;
                      if (WHERE && a.segPosF) a.segPosF[segBase + o] = (U32) (pos - rs) | (fwd ? MG_FWD_BIT : 0u);
                      if (WHERE && a.segRead) a.segRead[segBase + o] = r;
                    }
                }
              waveRun += (U32) __popcll (bs);
            }
          found += waveRun;
          /* what was not evaluated moves to the front */
          const U32 rem = qCount - nEval;
          if (rem && nEval)
            { const U32 v = (U32) lane < rem ? sCand[nEval + lane] : 0u;
              MG_WAVE_SYNC ();
              if ((U32) lane < rem) sCand[lane] = (unsigned short) v;
            }
          if (nEval) MG_WAVE_SYNC ();                                            /* the list may be written again */
          qOld = qOld > nEval ? qOld - nEval : 0;
          qCount = rem;
        }
      while (listed < nc);
      qOld = qCount;                                     /* what waits now is of this tile: the previous one for the next */
      tile = nextTile;
      curV = nextV; curHalo = nextHalo; ti = tiNext; nextFirstRead = nextNextFirst;
    }
  return found;
}

template <int MODE, bool WHERE>
__global__ __launch_bounds__ (MG_SCAN_THREADS)
void mgScanKernel (const MgScanArgs a)
{
  __shared__ __attribute__ ((aligned (16))) U32 sWordsAll[MG_WAVES][2][MG_TILE_WORDS + 8];
  __shared__ unsigned short sCandAll[MG_WAVES][MG_CAND_CAP + 2];      /* [MG_CAND_CAP]: where stores of other rounds' entries land */
  __shared__ U64 sOffAll[MG_WAVES][64];                              /* read offsets of a tile that holds read boundaries */
  __shared__ U32 sHist[512]; __shared__ U32 sDone;                   /* the workgroup's digit counts; waves that have finished */

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane ((int) (threadIdx.x >> 6));   /* uniform: the worker's state lives in SGPRs */
  const U64 worker = (U64) blockIdx.x * MG_WAVES + wave;
  if (a.histCount)
    { for (U32 b = threadIdx.x ; b < 512 ; b += MG_SCAN_THREADS) sHist[b] = 0;
      if (threadIdx.x == 0) sDone = 0;
      __syncthreads ();
    }
  if (worker >= a.nWorkers) { if (a.histCount) mgScanHistDone (a, sHist, &sDone, lane); return; }
  const U64 found = mgScanWorker<MODE, false, WHERE> (a, worker, lane, sWordsAll[wave], sCandAll[wave], sOffAll[wave], sHist);
  if (lane == 0) a.blockCount[worker] = found;
  if (a.histCount) mgScanHistDone (a, sHist, &sDone, lane);
}

/* The per-read iterator facade (modRCiterator, seqhash.c:154-177) in ONE launch: a single workgroup of MG_ITER_WAVES
 * workers scans one read of up to MG_ITER_MAX_TILES tiles -- its packed bases read straight from the caller's pinned host
 * buffer -- each worker into its own segment of a device scratch; then the workgroup concatenates the segments, in order,
 * into the replay block the iterator hands to modRCnext: out[0] = n, out[1 .. n] the k-mers, then n words pos | isF << 31
 * -- in pinned host memory, followed by a completion flag the host polls (no device-to-host copy, no stream wait). */
#define MG_ITER_WAVES     8
#define MG_ITER_MAX_TILES 64
struct MgIterOut { U64 *out; U64 capEntries; volatile U64 *flag; U64 seq; };

template <int MODE>
__global__ __launch_bounds__ (MG_ITER_WAVES * 64)
void mgIterScanKernel (const MgScanArgs a, const MgIterOut o)
{
  __shared__ __attribute__ ((aligned (16))) U32 sWordsAll[MG_ITER_WAVES][2][MG_TILE_WORDS + 8];
  __shared__ unsigned short sCandAll[MG_ITER_WAVES][MG_CAND_CAP + 2];
  __shared__ U64 sOffAll[MG_ITER_WAVES][64];
  __shared__ U64 sFound[MG_ITER_WAVES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane ((int) (threadIdx.x >> 6));
  U64 found = 0;
  if ((U64) wave < a.nWorkers) found = mgScanWorker<MODE, true, true> (a, (U64) wave, lane, sWordsAll[wave], sCandAll[wave], sOffAll[wave], (U32 *) 0);
  if (lane == 0) sFound[wave] = found;
  __syncthreads ();                                   /* the segments (global memory, written by this workgroup) and the counts are in */
  U64 before = 0, total = 0;
#pragma unroll
  for (int w = 0 ; w < MG_ITER_WAVES ; ++w) { const U64 v = sFound[w]; if (w < wave) before += v; total += v; }
  if (total <= o.capEntries)
    { U32 *posOut = reinterpret_cast<U32 *> (o.out + 1 + total);
      const U64 src = (U64) wave * a.segCap;
      for (U64 i = (U64) lane ; i < found ; i += 64)
        { o.out[1 + before + i] = a.segKmer[src + i];
          posOut[before + i] = a.segPosF[src + i];
        }
    }
  if (threadIdx.x == 0) o.out[0] = total;              /* > capEntries: nothing else was written, the host retries with room for it */
  __threadfence_system ();
  __syncthreads ();
  if (threadIdx.x == 0) __hip_atomic_store (const_cast<U64 *> (o.flag), o.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

/* exclusive scan of the per-block counts (one workgroup): segStart[b], and
 * dCount = { total, 1 if some segment overflowed or total > capacity, largest block count } */
__global__ __launch_bounds__ (1024)
void mgSegScanKernel (const U64 *__restrict__ blockCount, U32 nBlocks, U64 segCap, U64 capacity,
                      U64 *__restrict__ segStart, U64 *__restrict__ dCount)
{
  /* 8192 counts at a time: coalesced into LDS, eight consecutive ones per thread scanned there, the prefixes
     coalesced back out (one workgroup reading 48 scattered counts per thread spent 0.09 ms in the address unit) */
  __shared__ U64 sV[8192 + 1024];                    /* entry e at e + e/8: the threads' runs of eight do not collide */
  __shared__ U64 sWaveSum[16], sWaveMax[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  U64 carry = 0, gmx = 0;
  for (U32 base = 0 ; base < nBlocks ; base += 8192)
    {
#pragma unroll
      for (int j = 0 ; j < 8 ; ++j)
        { U32 e = (U32) j * 1024 + tid, b = base + e;
          sV[e + (e >> 3)] = b < nBlocks ? blockCount[b] : 0;
        }
      __syncthreads ();
      U64 c[8], sum = 0, mx = 0;
#pragma unroll
      for (int j = 0 ; j < 8 ; ++j) { c[j] = sV[tid * 9 + j]; sum += c[j]; if (c[j] > mx) mx = c[j]; }
      U64 incl = sum, wmx = mx;
#pragma unroll
      for (int off = 1 ; off < 64 ; off <<= 1)
        { U32 lo = __shfl_up ((U32) incl, off), hi = __shfl_up ((U32) (incl >> 32), off);
          if (lane >= off) incl += ((U64) hi << 32) | lo;
          U32 ml = __shfl_xor ((U32) wmx, off), mh = __shfl_xor ((U32) (wmx >> 32), off);
          U64 o = ((U64) mh << 32) | ml; if (o > wmx) wmx = o;
        }
      if (lane == 63) { sWaveSum[wave] = incl; sWaveMax[wave] = wmx; }
      __syncthreads ();
      U64 before = 0, total = 0;
#pragma unroll
      for (int w = 0 ; w < 16 ; ++w) { U64 v = sWaveSum[w]; if (w < wave) before += v; total += v; if (sWaveMax[w] > gmx) gmx = sWaveMax[w]; }
      U64 run = carry + before + incl - sum;
#pragma unroll
      for (int j = 0 ; j < 8 ; ++j) { sV[tid * 9 + j] = run; run += c[j]; }
      __syncthreads ();
#pragma unroll
      for (int j = 0 ; j < 8 ; ++j)
        { U32 e = (U32) j * 1024 + tid, b = base + e;
          if (b < nBlocks) segStart[b] = sV[e + (e >> 3)];
        }
      carry += total;
      __syncthreads ();
    }
  if (tid == 0)
    { segStart[nBlocks] = carry;                    /* one entry more than segments: ordinal -> segment searches need no bound */
      dCount[0] = carry;
      dCount[1] = (gmx > segCap || carry > capacity) ? 1 : 0;
      dCount[2] = gmx;
      U64 need = gmx * (U64) nBlocks;               /* capacity whose per-segment share covers the fullest segment */
      dCount[3] = need > carry ? need : carry;
    }
}

/* segments -> dense (read,pos)-ordered arrays.  A workgroup takes a run of consecutive segments.  With
 * histBins != 0 it also counts, per coarse digit of the table bucket, the k-mers it copies (what the first
 * partition pass of the modset build needs): a few ALU instructions per element in a kernel that only copies. */
__global__ __launch_bounds__ (256)
void mgSegCompactKernel (const U64 *__restrict__ segKmer, const U32 *__restrict__ segPosF,
                         const U32 *__restrict__ segRead, U64 segCap, U32 nSegs,
                         const U64 *__restrict__ blockCount, const U64 *__restrict__ segStart,
                         U64 *__restrict__ outKmer, U32 *__restrict__ outPosF, U32 *__restrict__ outRead,
                         U64 capacity, const U64 *__restrict__ dCount,
                         int histLog2NB, int histKbits, int histShift, U32 histBins, U32 *__restrict__ histCount)
{
  __shared__ U32 sH[512];
  if (dCount[1]) return;                       /* overflow: the caller retries with the reported sizes */
  if (histBins) { for (U32 b = threadIdx.x ; b < histBins ; b += 256) sH[b] = 0; __syncthreads (); }
  const U32 per = (nSegs + gridDim.x - 1) / gridDim.x;
  U32 sg = blockIdx.x * per;
  const U32 sgEnd = sg + per < nSegs ? sg + per : nSegs;
  for ( ; sg < sgEnd ; ++sg)
    { const U64 n = blockCount[sg], dst = segStart[sg], src = (U64) sg * segCap;
      for (U64 i0 = 0 ; i0 < n ; i0 += 4 * 256)            /* four elements per lane: every load in flight, and landed, before the first store (with loads and stores both outstanding each store waits for the one before) */
        { U64 km[4]; U32 pf[4], rd[4];
#pragma unroll
          for (int j = 0 ; j < 4 ; ++j)
            { const U64 i = i0 + (U64) j * 256 + threadIdx.x;
              km[j] = 0; pf[j] = 0; rd[j] = 0;
              if (i < n)
                { if (outKmer || histBins) km[j] = segKmer[src + i];
                  if (outPosF) pf[j] = segPosF[src + i];
                  if (outRead) rd[j] = segRead[src + i];
                }
            }
#pragma unroll
          for (int j = 0 ; j < 4 ; ++j) asm volatile ("" : "+v" (km[j]), "+v" (pf[j]), "+v" (rd[j]));
#pragma unroll
          for (int j = 0 ; j < 4 ; ++j)
            { const U64 i = i0 + (U64) j * 256 + threadIdx.x;
              if (i >= n) continue;
              if (outKmer) outKmer[dst + i] = km[j];       /* 0: only count (the dense copy is not wanted, or not yet) */
              if (outPosF) outPosF[dst + i] = pf[j];
              if (outRead) outRead[dst + i] = rd[j];
              if (histBins)
                { MgGeom hg; hg.R = 0; hg.log2NB = histLog2NB; hg.kbits = histKbits;
                  const U32 bucket = mgBucketOfM (mgMixK (km[j], histKbits), hg);
                  atomicAdd (&sH[(bucket >> histShift) & (histBins - 1)], 1u);
                }
            }
        }
    }
  if (histBins)
    { __syncthreads ();
      for (U32 b = threadIdx.x ; b < histBins ; b += 256) if (sH[b]) atomicAdd (&histCount[b * MG_HIST_STRIDE], sH[b]);
    }
}

/* ---------------------------------------------------------------------------------------- */

static inline U64 mgNumTiles (U64 totalBases) { return (totalBases + MG_TILE_BASES - 1) / MG_TILE_BASES; }

/* Scan geometry: G workers (wavefronts), each owning tilesPerBlock consecutive tiles and one output segment. */
struct MgScanGeom { U64 nTiles, tilesPerBlock; U32 nBlocks; U64 segCap; };       /* nBlocks = number of workers = segments */
#define MG_SCAN_MAX_BLOCKS 49152   /* workers per launch: many short ranges balance the tail (as workgroups: 2048 -> 4.63 ms, 12288 -> 4.07 per 10 Gbp) */

static MgScanGeom mgScanGeometryTiles (U64 nTiles, U64 capacity)
{
  MgScanGeom g;
  g.nTiles = nTiles;
  const long gridKnob = mgKnobs ()->scanGrid;                 /* test knob: several tiles in a worker's range */
  const long maxBlocks = gridKnob != MG_KNOB_UNSET && gridKnob > 0 ? gridKnob : MG_SCAN_MAX_BLOCKS;
  U64 want = g.nTiles < (U64) maxBlocks ? g.nTiles : (U64) maxBlocks;
  /* a batch below the headline's size: eight tiles a worker rather than a worker per tile or two, down to 8192 workers (the chip's wave
     slots).  Every workgroup ends by adding its digit counts to the same few hundred words, and every segment is one more piece for
     the kernels that walk them: 0.2 Gbp at a tile per worker -- scan 0.24 ms, step 0.70 -- against six tiles per worker -- 0.08, 0.46
     (tools/size_sweep_probe.py) */
  if (gridKnob == MG_KNOB_UNSET || gridKnob <= 0)
    { U64 fewer = g.nTiles / 8; if (fewer < 8192) fewer = 8192;
      if (want > fewer) want = fewer;
    }
  if (!want) want = 1;
  g.tilesPerBlock = (g.nTiles + want - 1) / want; if (!g.tilesPerBlock) g.tilesPerBlock = 1;
  g.nBlocks = (U32) ((g.nTiles + g.tilesPerBlock - 1) / g.tilesPerBlock); if (!g.nBlocks) g.nBlocks = 1;
  /* a block's fair share of the caller's capacity three times over, never more than its k-mer starts.  The room costs memory only
     (a segment is read up to its count), and a worker's range is about one long read: reads from a satellite array whose monomer
     is rich in modimizers, or with a long homopolymer, hold two or three times the average -- with an eighth of slack every real
     batch had a worker that overflowed, and an overflow means the whole scan again with more room (tools/realistic_probe.py:
     the scan 0.72 instead of 0.36 ms per Gbp) */
  /* Footprint (what mgScanWorkBytes asks for): 16 bytes per segment entry (k-mer 8, pos 4, read 4), segments three times the
     fair share of the caller's capacity, capacity itself 1.25 N / d + 65536: 60 N / d bytes -- 9.4 GB for a 10 Gbp batch at d = 64;
     at small d the factor is 1.5 (below): 7.5 bytes per base of the batch at d = 4 (config 5, 1 Gbp: 7.5 GB; rounds 1 - 4: 16).  Memory only: a segment is read up to its count.  Callers that scan
     several large batches at once on one device should size for that (MODGPU_SEG_SLACK, 1..8: the factor, default 3; 1 = an
     eighth of slack as in round 2, at the price of a second scan when a worker overflows). */
  /* Dense selections (round 5): where more than one start in sixteen is expected to be a modimizer (d < 16: capacity is 1.25 N / d) the
     factor is 1.5, not 3 -- at d = 4 three times the share IS every start of the range (16 bytes per base: a 10 Gbp batch of config 5's
     shape asked for 160 GB), while the spread of a range's count around its share shrinks with density (32 768 starts at 1/4: 8192 +- 78):
     7.5 bytes per base instead.  What does overflow -- a homopolymer run whose one k-mer is a modimizer -- takes the second scan. */
  U64 share = capacity / g.nBlocks;
  const long slackKnob = mgKnobs ()->segSlack;
  const bool dense = capacity > (g.nTiles * (U64) MG_TILE_BASES) / 16;
  const U64 slack = slackKnob != MG_KNOB_UNSET && slackKnob >= 1 && slackKnob <= 8 ? (U64) slackKnob : 3;
  U64 seg = slack > 1 ? slack * share + 64 : share + share / 8 + 64;
  if (dense && (slackKnob == MG_KNOB_UNSET || slackKnob < 1 || slackKnob > 8)) seg = share + share / 2 + 64;
  U64 most = g.tilesPerBlock * (U64) MG_TILE_BASES;
  g.segCap = seg < most ? seg : most;
  return g;
}

static inline size_t mgAl (size_t n) { return (n + 255) & ~(size_t) 255; }

U64 mgScanTiles (U64 totalBases) { return mgNumTiles (totalBases); }
size_t mgScanInfoBytes (U64 totalBases) { return mgAl ((mgNumTiles (totalBases) + 2) * sizeof (MgTileInfo)); }

/* per-tile read metadata for a whole batch (shared by every range launch over it) */
MgStatus mgScanPrepare (const U64 *dReadOffsets, U32 nReads, U64 totalBases, void *dInfo, hipStream_t st)
{
  U64 nTiles = mgNumTiles (totalBases);
  if (!nTiles || !nReads) return MG_OK;
  MG_LAUNCH (MG_K_TILE_FIRST_READ, st, mgTileInfoKernel, dim3 ((unsigned) ((nTiles + 1 + 255) / 256)), dim3 (256), 0, st,
             dReadOffsets, nReads, nTiles, totalBases, (MgTileInfo *) dInfo);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}

/* range work buffer (all 256-byte aligned): blockCount[G] | segStart[G] | segKmer[G*segCap] | segPosF[..] | segRead[..] */
size_t mgScanRangeWorkBytes (U64 nTilesRange, U64 capacity)
{
  MgScanGeom g = mgScanGeometryTiles (nTilesRange, capacity);
  size_t segN = (size_t) g.nBlocks * g.segCap;
  return mgAl (g.nBlocks * 8) + mgAl ((g.nBlocks + 1) * 8) + mgAl (segN * 8) + 2 * mgAl (segN * 4) + 256;
}

size_t mgScanWorkBytes (U64 totalBases, U32 nReads, U64 capacity)
{
  (void) nReads;
  return mgScanRangeWorkBytes (mgNumTiles (totalBases), capacity) + mgScanInfoBytes (totalBases);
}

/* which phase-A mode a hasher gets; sets the filter's constants in *a */
static int mgScanMode (const MgHashParams &p, MgScanArgs *a)
{
  const bool pow2 = (p.dOddInv == 1 && p.dOddLim == ~0ull);
  const int B = p.shift1 + p.dShift;
  const int forceGeneric = mgKnobs ()->scanGeneric == 1;       /* test knob */
  a->fS = 0; a->thresh = 0;
  /* the filter passes 2/d of the starts to the exact evaluation: from d = 8 up that beats computing both full
     hashes everywhere (measured at k=31: d=4 1.80 ms/Gbp exact vs 2.29 filtered) */
  if (pow2 && p.dShift >= 3 && B <= 32 && p.k >= 17 && !forceGeneric)
    { a->fS = (U32) (p.factor1 << (32 - B));
      a->thresh = (U32) 1 << (32 - p.dShift);
      return MG_MODE_FAST;
    }
  if (pow2) return MG_MODE_POW2;
  const bool small = p.small32 && mgKnobs ()->scanDiv64 != 1;      /* (test knob: 1 = the 64-bit test whatever k and d) */
  return p.dShift == 0 ? (small ? MG_MODE_ODD32 : MG_MODE_ODD) : (small ? MG_MODE_ANY32 : MG_MODE_ANY);
}

/* scan tiles [tile0, tile1) of the batch: the modimizers of those k-mer starts, dense and in order */
MgStatus mgLaunchScanRange (const MgHashParams &p, const U32 *dPacked, U64 totalBases,
                            const U64 *dReadOffsets, U32 nReads, const void *dInfo, U64 tile0, U64 tile1,
                            U64 *dKmer, U32 *dPosF, U32 *dReadId, U64 capacity,
                            U64 *dCount, void *dWork, hipStream_t st, const MgHistReq *hist, MgSegSrc *lazy)
{
  if (lazy) { lazy->segKmer = 0; lazy->nSegs = 0; }
  MG_HIP (hipMemsetAsync (dCount, 0, 4 * sizeof (U64), st));
  if (hist && hist->binCount) MG_HIP (hipMemsetAsync (hist->binCount, 0, 512 * MG_HIST_STRIDE * sizeof (U32), st));
  if (tile1 <= tile0 || !nReads) return MG_OK;
  MgScanGeom g = mgScanGeometryTiles (tile1 - tile0, capacity);
  char *wb = (char *) dWork;
  size_t segN = (size_t) g.nBlocks * g.segCap;
  U64 *blockCount = (U64 *) wb;                  wb += mgAl (g.nBlocks * 8);
  U64 *segStart = (U64 *) wb;                    wb += mgAl ((g.nBlocks + 1) * 8);
  U64 *segKmer = (U64 *) wb;                     wb += mgAl (segN * 8);
  U32 *segPosF = (U32 *) wb;                     wb += mgAl (segN * 4);
  U32 *segRead = (U32 *) wb;

  MgScanArgs a;
  a.p = p; a.packed = dPacked; a.nWordsAlloc = (U64) mgPackedWords (totalBases); a.totalBases = totalBases;
  a.readOff = dReadOffsets; a.nReads = nReads; a.tileInfo = (const MgTileInfo *) dInfo;
  a.tileBegin = tile0; a.tileLimit = tile1;
  a.tilesPerWorker = g.tilesPerBlock; a.nWorkers = g.nBlocks; a.segCap = g.segCap;
  a.segKmer = segKmer; a.segPosF = dPosF ? segPosF : 0; a.segRead = dReadId ? segRead : 0; a.blockCount = blockCount;
  a.fS = 0; a.thresh = 0;
  /* the first partition digit, counted by the scan itself when the table hash allows it (2k >= 24); otherwise by the compaction kernel */
  int hHiB = 0, hLoB = 0;
  if (hist && hist->binCount)
    { mgPartSplit (hist->log2NB, &hHiB, &hLoB);
      if (hist->hiB > 0 && hist->hiB <= hist->log2NB && hist->hiB <= 9) { hHiB = hist->hiB; hLoB = hist->log2NB - hHiB; }     /* (the partitioned lookup's finer first digit; sHist holds 512) */
    }
  const int histEnv = mgKnobs ()->scanHist == 0 ? 0 : 1;   /* test knob: 0 = the compaction kernel counts */
  const bool scanCounts = histEnv && hist && hist->binCount && hist->kbits >= 24 && hHiB >= 1 && hHiB <= MG_MIX_TOP && hLoB + hHiB == hist->log2NB;
  a.histCount = scanCounts ? hist->binCount : 0; a.histKbits = hist ? hist->kbits : 64; a.histHiB = hHiB;
#ifdef MG_ABLATE
  { const long dbg = mgKnobs ()->scanDebug; a.debug = dbg != MG_KNOB_UNSET ? (U32) dbg : 0u; }
#endif
  const unsigned grid = (g.nBlocks + MG_WAVES - 1) / MG_WAVES;
  const int mode = mgScanMode (p, &a);
  const bool where = a.segPosF || a.segRead;
#define MG_SCAN_LAUNCH(M) do { if (where) MG_LAUNCH (MG_K_SCAN, st, (mgScanKernel<M, true>), dim3 (grid), dim3 (MG_SCAN_THREADS), 0, st, a); \
                               else       MG_LAUNCH (MG_K_SCAN, st, (mgScanKernel<M, false>), dim3 (grid), dim3 (MG_SCAN_THREADS), 0, st, a); } while (0)
  if (mode == MG_MODE_FAST) MG_SCAN_LAUNCH (MG_MODE_FAST); else if (mode == MG_MODE_POW2) MG_SCAN_LAUNCH (MG_MODE_POW2);
  else if (mode == MG_MODE_ODD) MG_SCAN_LAUNCH (MG_MODE_ODD); else if (mode == MG_MODE_ODD32) MG_SCAN_LAUNCH (MG_MODE_ODD32);
  else if (mode == MG_MODE_ANY32) MG_SCAN_LAUNCH (MG_MODE_ANY32); else MG_SCAN_LAUNCH (MG_MODE_ANY);
#undef MG_SCAN_LAUNCH
  MG_HIP (hipGetLastError ());
  MG_LAUNCH (MG_K_SEG_SCAN, st, mgSegScanKernel, dim3 (1), dim3 (1024), 0, st, blockCount, g.nBlocks, g.segCap, capacity, segStart, dCount);
  MG_HIP (hipGetLastError ());
  { int hiB = 0, loB = 0; U32 bins = 0;
    if (hist && hist->binCount && !scanCounts) { mgPartSplit (hist->log2NB, &hiB, &loB); bins = (U32) 1 << hiB; }
    const unsigned cgrid = g.nBlocks < 4096 ? g.nBlocks : 4096;
    if (lazy) { lazy->segKmer = segKmer; lazy->segCount = blockCount; lazy->segStart = segStart; lazy->segCap = g.segCap; lazy->nSegs = g.nBlocks; }
    if (!lazy || bins || dPosF || dReadId)                /* lazy: the k-mers stay in the segments; pos / read, when asked for, are made dense all the same */
      MG_LAUNCH (MG_K_SEG_COMPACT, st, mgSegCompactKernel, dim3 (cgrid), dim3 (256), 0, st,
                 segKmer, a.segPosF, a.segRead, g.segCap, g.nBlocks, blockCount, segStart, lazy ? (U64 *) 0 : dKmer, dPosF, dReadId, capacity, dCount,
                 hist ? hist->log2NB : 0, hist ? hist->kbits : 64, loB, bins, hist ? hist->binCount : (U32 *) 0);
  }
  MG_HIP (hipGetLastError ());
  return MG_OK;
}

/* the dense k-mer array from the segments of a lazy scan */
MgStatus mgLaunchSegCompact (const MgSegSrc &src, U64 *dKmer, U64 capacity, const U64 *dCount, hipStream_t st)
{
  if (!src.nSegs) return MG_OK;
  const unsigned cgrid = src.nSegs < 4096 ? src.nSegs : 4096;
  MG_LAUNCH (MG_K_SEG_COMPACT, st, mgSegCompactKernel, dim3 (cgrid), dim3 (256), 0, st,
             src.segKmer, (const U32 *) 0, (const U32 *) 0, src.segCap, src.nSegs, src.segCount, src.segStart, dKmer, (U32 *) 0, (U32 *) 0, capacity, dCount,
             0, 64, 0, 0u, (U32 *) 0);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}

MgStatus mgLaunchScan (const MgHashParams &p, const U32 *dPacked, U64 totalBases,
                       const U64 *dReadOffsets, U32 nReads,
                       U64 *dKmer, U32 *dPosF, U32 *dReadId, U64 capacity,
                       U64 *dCount, void *dWork, hipStream_t st, const MgHistReq *hist, MgSegSrc *lazy)
{
  const U64 nTiles = mgNumTiles (totalBases);
  char *info = (char *) dWork + mgScanRangeWorkBytes (nTiles, capacity);
  MgStatus s = mgScanPrepare (dReadOffsets, nReads, totalBases, info, st); if (s) return s;
  return mgLaunchScanRange (p, dPacked, totalBases, dReadOffsets, nReads, info, 0, nTiles,
                            dKmer, dPosF, dReadId, capacity, dCount, dWork, st, hist, lazy);
}

/* ---------------------------------------------------------------------------------------- */
/* the per-read iterator facade: one launch per read (see mgIterScanKernel)                   */

U64 mgIterMaxBases (void) { return (U64) MG_ITER_MAX_TILES * MG_TILE_BASES; }
/* entries of device scratch (k-mers: 8 bytes, pos: 4 bytes each) the launch below needs */
U64 mgIterSegEntries (void) { return (U64) MG_ITER_MAX_TILES * MG_TILE_BASES; }

/* dPacked / dReadOff / out / flag: device-visible addresses of pinned host memory (or device memory); dReadOff = {0, totalBases} */
MgStatus mgLaunchIterScan (const MgHashParams &p, const U32 *dPacked, U64 totalBases, const U64 *dReadOff,
                           U64 *dSegKmer, U32 *dSegPosF, U64 *out, U64 capEntries, U64 *flag, U64 seq, hipStream_t st)
{
  const U64 nTiles = mgNumTiles (totalBases);
  if (!nTiles || nTiles > MG_ITER_MAX_TILES) { mgSetError ("internal: iterator scan of %llu bases", (unsigned long long) totalBases); return MG_ERR_ARG; }
  MgScanArgs a;
  a.p = p; a.packed = dPacked; a.nWordsAlloc = (U64) mgPackedWords (totalBases); a.totalBases = totalBases;
  a.readOff = dReadOff; a.nReads = 1; a.tileInfo = 0;
  a.tileBegin = 0; a.tileLimit = nTiles;
  a.tilesPerWorker = (nTiles + MG_ITER_WAVES - 1) / MG_ITER_WAVES;
  a.nWorkers = (nTiles + a.tilesPerWorker - 1) / a.tilesPerWorker;
  a.segCap = a.tilesPerWorker * (U64) MG_TILE_BASES;                 /* every start of a worker's tiles: cannot overflow */
  a.segKmer = dSegKmer; a.segPosF = dSegPosF; a.segRead = 0; a.blockCount = 0;
  a.histCount = 0; a.histKbits = 64; a.histHiB = 0;
#ifdef MG_ABLATE
  a.debug = 0;
#endif
  MgIterOut o; o.out = out; o.capEntries = capEntries; o.flag = flag; o.seq = seq;
  const int mode = mgScanMode (p, &a);
  if (mode == MG_MODE_FAST)      hipLaunchKernelGGL (mgIterScanKernel<MG_MODE_FAST>, dim3 (1), dim3 (MG_ITER_WAVES * 64), 0, st, a, o);
  else if (mode == MG_MODE_POW2) hipLaunchKernelGGL (mgIterScanKernel<MG_MODE_POW2>, dim3 (1), dim3 (MG_ITER_WAVES * 64), 0, st, a, o);
  else if (mode == MG_MODE_ODD)  hipLaunchKernelGGL (mgIterScanKernel<MG_MODE_ODD>, dim3 (1), dim3 (MG_ITER_WAVES * 64), 0, st, a, o);
  else if (mode == MG_MODE_ODD32) hipLaunchKernelGGL (mgIterScanKernel<MG_MODE_ODD32>, dim3 (1), dim3 (MG_ITER_WAVES * 64), 0, st, a, o);
  else if (mode == MG_MODE_ANY32) hipLaunchKernelGGL (mgIterScanKernel<MG_MODE_ANY32>, dim3 (1), dim3 (MG_ITER_WAVES * 64), 0, st, a, o);
  else                           hipLaunchKernelGGL (mgIterScanKernel<MG_MODE_ANY>, dim3 (1), dim3 (MG_ITER_WAVES * 64), 0, st, a, o);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}
