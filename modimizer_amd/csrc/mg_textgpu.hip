/* mg_textgpu.hip — the file front end on the device: plain FASTA text -> 2-bit packed reads, parsed by the GPU.
 *
 * What the reference's callers get from seqIOread for FASTA (seqio.c:302-323 with dna2indexConv after the N -> 0 patch,
 * modutils.c:39): a record starts at a '>' that is the first byte of a line, its header runs to the end of that line, its
 * sequence is every A/a C/c G/g T/t N/n (0 1 2 3 0) up to the next record start, all other bytes dropped.  The host parser
 * (mg_seqio.c) does that with a pool of threads at about 19 GB/s of text on the box's 16 allowed CPUs, which is what bounds
 * mgAddSequenceFile (12-14 Gbp/s file -> modset) under kernels that take 1 Tbp/s.  Here the host only moves bytes: the
 * file is read window by window straight into pinned memory (parallel pread: the copy out of the page cache), the window
 * goes across PCIe as it is, and three small kernels per window do the parsing:
 *
 *   "\n>" is a record start whatever came before (a header ends at its newline, so after a newline the state is always
 *   "sequence"); a byte lies in a header iff the LAST EVENT before it -- record start or newline -- is a record start.
 *   Events carry their position, so "last event" is a running maximum:
 *     K1  per tile of 4 KiB: the code (position << 1 | isStart) of its last event;
 *     K2  one workgroup: running maximum over the tiles (+ the state the previous window ended in) -> the state at every
 *         tile's first byte;
 *     K3  per tile: the bytes again, the running maximum inside the tile, per thread the bases and record starts of its
 *         16 bytes -> the tile's counts;
 *     K4  one workgroup: prefix sums of the counts, on top of what the batch holds already (device-resident totals);
 *     K5  per tile: the bytes a third time; bases (one byte each) and record offsets written where the sums say.
 *   The bases of complete records are packed to 2 bits (mgLaunchPack) and handed to mgAddReadsDevice a batch at a time; the
 *   record the batch ends in the middle of is carried to the front of the next batch on the device.
 *
 * FASTQ goes the same way with other kernels (see "FASTQ" below): the line a byte belongs to is the number of newlines before
 * it, line mod 4 says what the byte is; the format's rules are checked on the device and never judged there.
 *
 * Return codes of the parser (txParseFile, mgAddSequenceFileDevice, mgTextForEachBatchDevice): 0 = the whole file; -1 = error
 * (mgLastError); -2 = not a file for this path -- gzip / blocked gzip, a file whose last byte is not a newline (the reference reports
 * the unfinished record, seqio.c:213-217), a FASTA file whose last line is a header, anything that is not a regular file, no
 * device: the caller uses the host parser from the start; -3 = FASTQ text that breaks a rule, or ends inside a record, somewhere
 * after byte *resumeOff: the records before it have been handed on, the host parser continues from there (and says what the
 * reference says).  mgTextParseFileDevice (the test hook) maps -3 to -2 and hands nothing on.  Record ids (seqio.c:303-304) are
 * extracted for the callers that print them (mgReferenceFastaRead, mgQueryFile), not for mgAddSequenceFile (modutils.c:35 ignores them).
 */
#include <fcntl.h>
#include <pthread.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>
#include <mutex>
#include <thread>
#include <vector>
#include "mg_common.h"
#include "mg_internal.h"

#define TX_THREADS 256
#define TX_PER     16                                   /* bytes per thread: one 16-byte load */
#define TX_TILE    (TX_THREADS * TX_PER)                 /* 4 KiB of text per workgroup */

struct TxState {                                         /* device resident, carried from window to window */
  U64 lastEvent;        /* code of the last event of the text so far: bit 0 = it was a record start (we are in a header) */
  U64 accBases;         /* bases in the batch accumulator */
  U64 accRecs;          /* record starts in the batch accumulator */
};
/* per window, from the host: the file offset of its first byte (event codes carry file positions) and the byte before it (a
   newline before the file's first byte) */

/* base code of a text byte, 4 = not a base (seqio.c:643-652 + N -> 0): branch-free on the letter */
__device__ __forceinline__ U32 txCode (U32 c)
{
  const U32 u = c & 0xdfu;                                /* upper case */
  U32 code = 4;
  code = u == 'A' ? 0u : code; code = u == 'C' ? 1u : code; code = u == 'G' ? 2u : code; code = u == 'T' ? 3u : code; code = u == 'N' ? 0u : code;
  return code;
}

/* a thread's 16 bytes and the byte before them */
__device__ __forceinline__ void txLoad (const unsigned char *text, U64 n, U64 at, U32 prevByte, unsigned char *b, U32 *prev)
{
  uint4 v = make_uint4 (0, 0, 0, 0);
  if (at + 16 <= n) v = *reinterpret_cast<const uint4 *> (text + at);
  else { unsigned char t[16]; for (int j = 0 ; j < 16 ; ++j) t[j] = at + j < n ? text[at + j] : 0; memcpy (&v, t, 16); }
  memcpy (b, &v, 16);
  *prev = at ? text[at - 1] : prevByte;
}

/* the event code of byte i of the window (0: no event): newline -> even, record start -> odd; a later event has a larger code */
__device__ __forceinline__ U64 txEvent (U32 c, U32 prev, U64 filePos)
{
  if (c == '\n') return (filePos + 1) << 1;
  if (c == '>' && prev == '\n') return ((filePos + 1) << 1) | 1;
  return 0;
}

__device__ __forceinline__ U64 txBlockMax (U64 v, U64 *sRed)
{
  for (int off = 32 ; off ; off >>= 1)
    { const U64 o = ((U64) (U32) __shfl_xor ((int) (U32) (v >> 32), off) << 32) | (U32) __shfl_xor ((int) (U32) v, off);
      if (o > v) v = o;
    }
  if ((threadIdx.x & 63) == 0) sRed[threadIdx.x >> 6] = v;
  __syncthreads ();
  U64 m = 0;
  for (int w = 0 ; w < TX_THREADS / 64 ; ++w) if (sRed[w] > m) m = sRed[w];
  __syncthreads ();
  return m;
}

/* K1: the last event of every tile */
__global__ __launch_bounds__ (TX_THREADS)
void mgTextEventKernel (const unsigned char *__restrict__ text, U64 n, U64 textBase, U32 prevByte, U64 *__restrict__ tileEvent)
{
  __shared__ U64 sRed[TX_THREADS / 64];
  const U64 at = ((U64) blockIdx.x * TX_THREADS + threadIdx.x) * TX_PER;
  unsigned char b[16]; U32 prev = 0;
  U64 last = 0;
  if (at < n)
    { txLoad (text, n, at, prevByte, b, &prev);
      const U64 base = textBase + at;
#pragma unroll
      for (int j = 0 ; j < 16 ; ++j)
        { if (at + j < n) { const U64 e = txEvent (b[j], prev, base + j); if (e) last = e; }
          prev = b[j];
        }
    }
  const U64 m = txBlockMax (last, sRed);
  if (threadIdx.x == 0) tileEvent[blockIdx.x] = m;
}

/* K2 / K4 helper: one workgroup scans nTiles values (inclusive), running maximum or running sum, with a carry-in */
template <bool MAX>
__device__ __forceinline__ void txScanTiles (U64 *v, U64 nTiles, U64 carryIn, U64 *totalOut)
{
  __shared__ U64 sPart[1024];
  const int tid = threadIdx.x;
  const U64 per = (nTiles + 1023) / 1024;
  const U64 lo = (U64) tid * per, hi = lo + per < nTiles ? lo + per : nTiles;
  U64 acc = 0;
  for (U64 i = lo ; i < hi ; ++i) { const U64 x = v[i]; acc = MAX ? (x > acc ? x : acc) : acc + x; }
  sPart[tid] = acc;
  __syncthreads ();
  for (int off = 1 ; off < 1024 ; off <<= 1)
    { const U64 o = tid >= off ? sPart[tid - off] : 0;
      __syncthreads ();
      sPart[tid] = MAX ? (o > sPart[tid] ? o : sPart[tid]) : sPart[tid] + o;
      __syncthreads ();
    }
  U64 run = tid ? sPart[tid - 1] : 0;                     /* exclusive over the threads' pieces */
  run = MAX ? (carryIn > run ? carryIn : run) : run + carryIn;
  for (U64 i = lo ; i < hi ; ++i)                          /* v[i] becomes the EXCLUSIVE value: what holds at the tile's first byte */
    { const U64 x = v[i]; v[i] = run; run = MAX ? (x > run ? x : run) : run + x; }
  if (tid == 1023) { const U64 t = sPart[1023]; *totalOut = MAX ? (carryIn > t ? carryIn : t) : t + carryIn; }
}

__global__ __launch_bounds__ (1024)
void mgTextStateScanKernel (U64 *tileEvent, U64 nTiles, TxState *st)
{
  __shared__ U64 total;
  txScanTiles<true> (tileEvent, nTiles, st->lastEvent, &total);
  __syncthreads ();
  if (threadIdx.x == 0) st->lastEvent = total;            /* the state the next window starts in */
}

/* the per-thread walk shared by K3 and K5: state at the thread's first byte from the tile's incoming event and the events of
 * the threads before it in the tile */
__device__ __forceinline__ U64 txIncoming (U64 myLast, U64 tileIn, U64 *sScan)
{
  /* exclusive running maximum over the threads of the workgroup */
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  U64 v = myLast;
  for (int off = 1 ; off < 64 ; off <<= 1)
    { const U64 o = ((U64) (U32) __shfl_up ((int) (U32) (v >> 32), off) << 32) | (U32) __shfl_up ((int) (U32) v, off);
      if (lane >= off && o > v) v = o;
    }
  if (lane == 63) sScan[wave] = v;
  __syncthreads ();
  U64 before = tileIn;
  for (int w = 0 ; w < wave ; ++w) if (sScan[w] > before) before = sScan[w];
  U64 excl = ((U64) (U32) __shfl_up ((int) (U32) (v >> 32), 1) << 32) | (U32) __shfl_up ((int) (U32) v, 1);
  if (lane == 0) excl = 0;
  __syncthreads ();
  return excl > before ? excl : before;
}

/* K3: bases and record starts of every tile */
__global__ __launch_bounds__ (TX_THREADS)
void mgTextCountKernel (const unsigned char *__restrict__ text, U64 n, U64 textBase, U32 prevByte, const U64 *__restrict__ tileIn,
                        U32 *__restrict__ tileBases, U32 *__restrict__ tileStarts)
{
  __shared__ U64 sScan[TX_THREADS / 64];
  __shared__ U32 sB[TX_THREADS / 64], sS[TX_THREADS / 64];
  const U64 at = ((U64) blockIdx.x * TX_THREADS + threadIdx.x) * TX_PER;
  unsigned char b[16]; U32 prev0 = 0;
  U64 last = 0;
  if (at < n)
    { txLoad (text, n, at, prevByte, b, &prev0);
      U32 prev = prev0;
      const U64 base = textBase + at;
#pragma unroll
      for (int j = 0 ; j < 16 ; ++j) { if (at + j < n) { const U64 e = txEvent (b[j], prev, base + j); if (e) last = e; } prev = b[j]; }
    }
  const U64 in = txIncoming (last, tileIn[blockIdx.x], sScan);
  U32 nb = 0, ns = 0;
  if (at < n)
    { bool header = (in & 1) != 0;
      U32 prev = prev0;
#pragma unroll
      for (int j = 0 ; j < 16 ; ++j)
        { if (at + j < n)
            { const U32 c = b[j];
              if (c == '>' && prev == '\n') { header = true; ++ns; }
              else if (c == '\n') header = false;
              else if (!header && txCode (c) < 4) ++nb;
            }
          prev = b[j];
        }
    }
  for (int off = 32 ; off ; off >>= 1) { nb += __shfl_xor (nb, off); ns += __shfl_xor (ns, off); }
  if ((threadIdx.x & 63) == 0) { sB[threadIdx.x >> 6] = nb; sS[threadIdx.x >> 6] = ns; }
  __syncthreads ();
  if (threadIdx.x == 0)
    { U32 tb = 0, ts = 0;
      for (int w = 0 ; w < TX_THREADS / 64 ; ++w) { tb += sB[w]; ts += sS[w]; }
      tileBases[blockIdx.x] = tb; tileStarts[blockIdx.x] = ts;
    }
}

/* K4: where every tile's bases and record offsets go in the batch accumulator; the new totals */
__global__ __launch_bounds__ (1024)
void mgTextOffsetScanKernel (const U32 *__restrict__ tileBases, const U32 *__restrict__ tileStarts, U64 nTiles,
                             U64 *__restrict__ tileBaseOff, U64 *__restrict__ tileStartOff, TxState *st, U64 *hostCounts)
{
  __shared__ U64 totB, totS;
  for (U64 i = threadIdx.x ; i < nTiles ; i += 1024) { tileBaseOff[i] = tileBases[i]; tileStartOff[i] = tileStarts[i]; }
  __syncthreads ();
  txScanTiles<false> (tileBaseOff, nTiles, st->accBases, &totB);
  __syncthreads ();
  txScanTiles<false> (tileStartOff, nTiles, st->accRecs, &totS);
  __syncthreads ();
  if (threadIdx.x == 0)
    { st->accBases = totB; st->accRecs = totS;
      hostCounts[0] = totB; hostCounts[1] = totS;           /* pinned host words: the host decides about the batch from them */
    }
}

/* K5: the bases (one byte each) and the record offsets, written where the sums say */
__global__ __launch_bounds__ (TX_THREADS)
void mgTextEmitKernel (const unsigned char *__restrict__ text, U64 n, U64 textBase, U32 prevByte,
                       const U64 *__restrict__ tileIn, const U64 *__restrict__ tileBaseOff, const U64 *__restrict__ tileStartOff,
                       unsigned char *__restrict__ bases, U64 basesCap, U64 *__restrict__ recOff, U64 recCap, U32 *__restrict__ overflow,
                       U64 *__restrict__ recPos)      /* != 0: the file position of every record's '>' (the callers that print record ids) */
{
  __shared__ U64 sScan[TX_THREADS / 64];
  __shared__ U32 sB[TX_THREADS / 64], sS[TX_THREADS / 64];
  const U64 at = ((U64) blockIdx.x * TX_THREADS + threadIdx.x) * TX_PER;
  unsigned char b[16]; U32 prev0 = 0;
  U64 last = 0;
  if (at < n)
    { txLoad (text, n, at, prevByte, b, &prev0);
      U32 prev = prev0;
      const U64 base = textBase + at;
#pragma unroll
      for (int j = 0 ; j < 16 ; ++j) { if (at + j < n) { const U64 e = txEvent (b[j], prev, base + j); if (e) last = e; } prev = b[j]; }
    }
  const U64 in = txIncoming (last, tileIn[blockIdx.x], sScan);
  /* the thread's counts (its place inside the tile), then the walk that writes */
  U32 nb = 0, ns = 0;
  if (at < n)
    { bool header = (in & 1) != 0;
      U32 prev = prev0;
#pragma unroll
      for (int j = 0 ; j < 16 ; ++j)
        { if (at + j < n)
            { const U32 c = b[j];
              if (c == '>' && prev == '\n') { header = true; ++ns; }
              else if (c == '\n') header = false;
              else if (!header && txCode (c) < 4) ++nb;
            }
          prev = b[j];
        }
    }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  U32 ib = nb, is = ns;
  for (int off = 1 ; off < 64 ; off <<= 1)
    { const U32 ob = __shfl_up (ib, off), os = __shfl_up (is, off);
      if (lane >= off) { ib += ob; is += os; }
    }
  if (lane == 63) { sB[wave] = ib; sS[wave] = is; }
  __syncthreads ();
  U32 wb = 0, ws = 0;
  for (int w = 0 ; w < wave ; ++w) { wb += sB[w]; ws += sS[w]; }
  U64 myB = tileBaseOff[blockIdx.x] + wb + ib - nb;
  U64 myS = tileStartOff[blockIdx.x] + ws + is - ns;
  if (!nb && !ns) return;
  if (myB + nb > basesCap || myS + ns > recCap) { *overflow = 1; return; }
  bool header = (in & 1) != 0;
  U32 prev = prev0;
#pragma unroll
  for (int j = 0 ; j < 16 ; ++j)
    { if (at + j < n)
        { const U32 c = b[j];
          if (c == '>' && prev == '\n') { header = true; if (recPos) recPos[myS] = textBase + at + (U64) j; recOff[myS++] = myB; }      /* a record's offset: the bases before its '>' */
          else if (c == '\n') header = false;
          else if (!header) { const U32 code = txCode (c); if (code < 4) bases[myB++] = (unsigned char) code; }
        }
      prev = b[j];
    }
}

/* the carried record moves to the front of the accumulator (regions may overlap: chunks in order, one workgroup) */
__global__ __launch_bounds__ (1024)
void mgTextCarryKernel (unsigned char *bases, U64 from, U64 count)
{
  for (U64 i0 = 0 ; i0 < count ; i0 += 1024)
    { const U64 i = i0 + threadIdx.x;
      unsigned char v = 0;
      if (i < count) v = bases[from + i];
      __syncthreads ();
      if (i < count) bases[i] = v;
      __syncthreads ();
    }
}

/* ... and when the regions do not overlap (the carried record is shorter than what was handed on: always, unless one record is longer than a
   whole batch) the move is a plain copy by the whole chip -- the one-workgroup form took 6 ms per carry of a 100 Mbp chromosome, a quarter of a
   reference file's read (tools/longfile_trace.sh) */
__global__ __launch_bounds__ (256)
void mgTextCarryWideKernel (unsigned char *bases, U64 from, U64 count)
{
  const U64 stride = (U64) gridDim.x * blockDim.x * 4;
  for (U64 i = ((U64) blockIdx.x * blockDim.x + threadIdx.x) * 4 ; i < count ; i += stride)
    { if (i + 4 <= count && !((from + i) & 3))
        *reinterpret_cast<U32 *> (bases + i) = *reinterpret_cast<const U32 *> (bases + from + i);
      else
        for (int j = 0 ; j < 4 && i + j < count ; ++j) bases[i + j] = bases[from + i + j];
    }
}
static void txCarry (unsigned char *bases, U64 from, U64 count, hipStream_t st)
{
  if (!count) return;
  if (count <= from)
    { unsigned grid = (unsigned) ((count / 4 + 255) / 256); if (grid > 8192) grid = 8192; if (!grid) grid = 1;
      hipLaunchKernelGGL (mgTextCarryWideKernel, dim3 (grid), dim3 (256), 0, st, bases, from, count);
    }
  else hipLaunchKernelGGL (mgTextCarryKernel, dim3 (1), dim3 (1024), 0, st, bases, from, count);
}

/* ======================================================================================== */
/* FASTQ: four lines a record (seqio.c:325-339).  The line a byte belongs to is the number of newlines before it, a prefix
 * sum; line mod 4 says what the byte is: 0 header ('@' first), 1 sequence (EVERY byte but the newline is a base: A/a C/c G/g
 * T/t N/n as in FASTA, anything else stays in the sequence as (char) -2, i.e. 2 once packed -- as the host parser and the
 * reference leave it), 2 the '+' line, 3 qualities (as many bytes as the sequence line).  A record is complete at its fourth
 * newline; there the kernels note the bases and the quality bytes counted so far and the newline's file position: the first
 * gives the read offsets, the first two the length check, the third where the host parser would take over.  The rules
 * ('@', '+', equal lengths, whole records at the end of the file) are CHECKED here and never judged: any breach, and the file
 * goes back to the host parser from the first record not yet added, which then says what the reference says. */
struct TqState {
  U64 nlCount;          /* newlines of the file so far */
  U64 accBases, accQual, accRecs;      /* of the batch accumulator: bases, quality bytes, completed records */
  U64 bad;              /* != 0: a rule was broken somewhere in the accumulator's text */
};

__device__ __forceinline__ U32 tqBlockSum (U32 v, U32 *sRed)
{
  for (int off = 32 ; off ; off >>= 1) v += __shfl_xor (v, off);
  if ((threadIdx.x & 63) == 0) sRed[threadIdx.x >> 6] = v;
  __syncthreads ();
  U32 t = 0;
  for (int w = 0 ; w < TX_THREADS / 64 ; ++w) t += sRed[w];
  __syncthreads ();
  return t;
}
/* exclusive prefix of v over the workgroup's threads */
__device__ __forceinline__ U32 tqBlockExcl (U32 v, U32 *sScan)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  U32 incl = v;
  for (int off = 1 ; off < 64 ; off <<= 1) { const U32 o = __shfl_up (incl, off); if (lane >= off) incl += o; }
  if (lane == 63) sScan[wave] = incl;
  __syncthreads ();
  U32 before = 0;
  for (int w = 0 ; w < wave ; ++w) before += sScan[w];
  __syncthreads ();
  return before + incl - v;
}

/* Kq1: newlines per tile */
__global__ __launch_bounds__ (TX_THREADS)
void mgTextNewlineKernel (const unsigned char *__restrict__ text, U64 n, U64 *__restrict__ tileNL)
{
  __shared__ U32 sRed[TX_THREADS / 64];
  const U64 at = ((U64) blockIdx.x * TX_THREADS + threadIdx.x) * TX_PER;
  unsigned char b[16]; U32 prev = 0; U32 c = 0;
  if (at < n)
    { txLoad (text, n, at, 0, b, &prev);
#pragma unroll
      for (int j = 0 ; j < 16 ; ++j) c += (at + j < n && b[j] == '\n') ? 1u : 0u;
    }
  const U32 t = tqBlockSum (c, sRed);
  if (threadIdx.x == 0) tileNL[blockIdx.x] = t;
}

/* Kq2: the line number at every tile's first byte */
__global__ __launch_bounds__ (1024)
void mgTextLineScanKernel (U64 *tileNL, U64 nTiles, TqState *st)
{
  __shared__ U64 total;
  txScanTiles<false> (tileNL, nTiles, st->nlCount, &total);
  __syncthreads ();
  if (threadIdx.x == 0) st->nlCount = total;
}

/* the walk over a thread's 16 bytes, shared by Kq3 (counts) and Kq5 (writes): EMIT = false counts only */
template <bool EMIT>
__device__ __forceinline__ void tqWalk (const unsigned char *b, U64 at, U64 n, U32 prev, U64 line, U64 filePos,
                                        U32 *nb, U32 *nq, U32 *ne, U32 *bad,
                                        unsigned char *bases, U64 myB, U64 myQ, U64 *endB, U64 *endQ, U64 *endP, U64 myE)
{
#pragma unroll
  for (int j = 0 ; j < 16 ; ++j)
    { if (at + j < n)
        { const U32 c = b[j];
          const U32 phase = (U32) line & 3u;
          if (prev == '\n' && ((phase == 0 && c != '@') || (phase == 2 && c != '+'))) *bad = 1;     /* seqio.c:326,333 */
          if (c == '\n')
            { if (phase == 3)
                { if (EMIT) { endB[myE] = myB; endQ[myE] = myQ; endP[myE] = filePos + (U64) j; ++myE; }
                  ++*ne;
                }
              ++line;
            }
          else if (phase == 1) { if (EMIT) { const U32 code = txCode (c); bases[myB++] = (unsigned char) (code < 4 ? code : 2u); } ++*nb; }
          else if (phase == 3) { if (EMIT) ++myQ; ++*nq; }
        }
      prev = b[j];
    }
}

/* Kq3: bases, quality bytes and completed records of every tile; rule checks at the line starts */
__global__ __launch_bounds__ (TX_THREADS)
void mgTextFastqCountKernel (const unsigned char *__restrict__ text, U64 n, U32 prevByte, const U64 *__restrict__ tileLine,
                             U32 *__restrict__ tileBases, U32 *__restrict__ tileQual, U32 *__restrict__ tileEnds, TqState *st)
{
  __shared__ U32 sScan[TX_THREADS / 64], sRed[TX_THREADS / 64];
  const U64 at = ((U64) blockIdx.x * TX_THREADS + threadIdx.x) * TX_PER;
  unsigned char b[16]; U32 prev = 0; U32 nl = 0;
  if (at < n)
    { txLoad (text, n, at, prevByte, b, &prev);
#pragma unroll
      for (int j = 0 ; j < 16 ; ++j) nl += (at + j < n && b[j] == '\n') ? 1u : 0u;
    }
  const U64 line = tileLine[blockIdx.x] + tqBlockExcl (nl, sScan);
  U32 nb = 0, nq = 0, ne = 0, bad = 0;
  if (at < n) tqWalk<false> (b, at, n, prev, line, 0, &nb, &nq, &ne, &bad, 0, 0, 0, 0, 0, 0, 0);
  if (bad) st->bad = 1;
  const U32 tb = tqBlockSum (nb, sRed), tq = tqBlockSum (nq, sRed), te = tqBlockSum (ne, sRed);
  if (threadIdx.x == 0) { tileBases[blockIdx.x] = tb; tileQual[blockIdx.x] = tq; tileEnds[blockIdx.x] = te; }
}

/* Kq4: where the tiles' bases / quality counts / record ends go; the new totals */
__global__ __launch_bounds__ (1024)
void mgTextFastqOffsetKernel (const U32 *__restrict__ tileBases, const U32 *__restrict__ tileQual, const U32 *__restrict__ tileEnds, U64 nTiles,
                              U64 *__restrict__ offB, U64 *__restrict__ offQ, U64 *__restrict__ offE, TqState *st, U64 *hostCounts)
{
  __shared__ U64 totB, totQ, totE;
  for (U64 i = threadIdx.x ; i < nTiles ; i += 1024) { offB[i] = tileBases[i]; offQ[i] = tileQual[i]; offE[i] = tileEnds[i]; }
  __syncthreads ();
  txScanTiles<false> (offB, nTiles, st->accBases, &totB);
  __syncthreads ();
  txScanTiles<false> (offQ, nTiles, st->accQual, &totQ);
  __syncthreads ();
  txScanTiles<false> (offE, nTiles, st->accRecs, &totE);
  __syncthreads ();
  if (threadIdx.x == 0)
    { st->accBases = totB; st->accQual = totQ; st->accRecs = totE;
      hostCounts[0] = totB; hostCounts[1] = totE; hostCounts[2] = totQ; hostCounts[3] = st->nlCount; hostCounts[4] = st->bad;
    }
}

/* Kq5: the bases, and at every record's last newline the three running values (entries 1 .. of endB / endQ / endP; entry 0 is
 * what held when the accumulator was started) */
__global__ __launch_bounds__ (TX_THREADS)
void mgTextFastqEmitKernel (const unsigned char *__restrict__ text, U64 n, U64 textBase, U32 prevByte, const U64 *__restrict__ tileLine,
                            const U64 *__restrict__ offB, const U64 *__restrict__ offQ, const U64 *__restrict__ offE,
                            unsigned char *__restrict__ bases, U64 basesCap, U64 *__restrict__ endB, U64 *__restrict__ endQ, U64 *__restrict__ endP, U64 recCap,
                            U32 *__restrict__ overflow)
{
  __shared__ U32 sScan[TX_THREADS / 64];
  const U64 at = ((U64) blockIdx.x * TX_THREADS + threadIdx.x) * TX_PER;
  unsigned char b[16]; U32 prev = 0; U32 nl = 0;
  if (at < n)
    { txLoad (text, n, at, prevByte, b, &prev);
#pragma unroll
      for (int j = 0 ; j < 16 ; ++j) nl += (at + j < n && b[j] == '\n') ? 1u : 0u;
    }
  const U64 line = tileLine[blockIdx.x] + tqBlockExcl (nl, sScan);
  U32 nb = 0, nq = 0, ne = 0, bad = 0;
  if (at < n) tqWalk<false> (b, at, n, prev, line, 0, &nb, &nq, &ne, &bad, 0, 0, 0, 0, 0, 0, 0);
  const U64 myB = offB[blockIdx.x] + tqBlockExcl (nb, sScan);
  const U64 myQ = offQ[blockIdx.x] + tqBlockExcl (nq, sScan);
  const U64 myE = offE[blockIdx.x] + tqBlockExcl (ne, sScan) + 1;                     /* entry 0 is the accumulator's start */
  if (at >= n || (!nb && !ne)) return;
  if (myB + nb > basesCap || myE + ne > recCap) { *overflow = 1; return; }
  U32 x0 = 0, x1 = 0, x2 = 0, x3 = 0;
  tqWalk<true> (b, at, n, prev, line, textBase + at, &x0, &x1, &x2, &x3, bases, myB, myQ, endB, endQ, endP, myE);
}

/* Kq6: a record's sequence line and quality line are of one length (seqio.c:339), for the records completed in [first, last] */
__global__ void mgTextFastqCheckKernel (const U64 *__restrict__ endB, const U64 *__restrict__ endQ, U64 first, U64 last, TqState *st)
{
  const U64 r = first + (U64) blockIdx.x * blockDim.x + threadIdx.x;
  if (r > last) return;
  if (endB[r] - endB[r - 1] != endQ[r] - endQ[r - 1]) st->bad = 1;
}

/* ---------------------------------------------------------------------------------------- */
/* host side                                                                                  */

static inline size_t txAl (size_t n) { return (n + 255) & ~(size_t) 255; }
#include <time.h>
static double txNow (void) { struct timespec ts; clock_gettime (CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static bool txTiming (void) { return mgKnobs ()->textTiming == 1; }   /* dev knob */
struct TxClock { double reserve = 0, read = 0, wait = 0, flush = 0, ids = 0, sink = 0, t0 = 0; void lap (double &slot) { const double n = txNow (); slot += n - t0; t0 = n; } };

struct TxBufs {
  int dev = -1;
  size_t window = 0;
  unsigned char *hPin[2] = { 0, 0 }; hipEvent_t h2dDone[2];
  unsigned char *dText[2] = { 0, 0 };
  U64 *hCounts = 0;                    /* pinned: {accBases, accRecs} after the last window's K4 */
  U64 *hHdr = 0, *dHdr = 0; size_t hdrCap = 0;    /* pinned: where a window's record headers start (4e5 of them in a window of short reads: into pageable memory the copy took as long as extracting the ids) */
  TxState *dState = 0; U32 *dOverflow = 0;
  U64 *dTileEvent = 0, *dTileBaseOff = 0, *dTileStartOff = 0; U32 *dTileBases = 0, *dTileStarts = 0; size_t tilesCap = 0;
  U64 *dTileOffQ = 0; U32 *dTileQual = 0; TqState *dTq = 0;      /* FASTQ: the third counted quantity */
  U64 *dEndQ = 0, *dEndP = 0;                                    /* FASTQ: per completed record (dRecOff is the first of the three) */
  U64 *dRecPos = 0;                                              /* FASTA: file position of every record's '>' (for the record ids) */
  unsigned char *dBases = 0; size_t basesCap = 0;
  U64 *dRecOff = 0; size_t recCap = 0;
  U32 *dPacked = 0; size_t packedWords = 0;
  hipStream_t copy = 0;
  std::mutex lock;
  void release ()
  { for (int i = 0 ; i < 2 ; ++i) { if (hPin[i]) { (void) hipHostFree (hPin[i]); (void) hipEventDestroy (h2dDone[i]); } (void) hipFree (dText[i]); hPin[i] = 0; dText[i] = 0; }
    if (hCounts) (void) hipHostFree (hCounts);
    if (hHdr) (void) hipHostFree (hHdr);
    hHdr = 0; dHdr = 0; hdrCap = 0;
    (void) hipFree (dState); (void) hipFree (dOverflow); (void) hipFree (dTileEvent); (void) hipFree (dTileBaseOff); (void) hipFree (dTileStartOff);
    (void) hipFree (dTileBases); (void) hipFree (dTileStarts); (void) hipFree (dBases); (void) hipFree (dRecOff); (void) hipFree (dPacked);
    (void) hipFree (dTileOffQ); (void) hipFree (dTileQual); (void) hipFree (dTq); (void) hipFree (dEndQ); (void) hipFree (dEndP); (void) hipFree (dRecPos);
    dTileOffQ = 0; dTileQual = 0; dTq = 0; dEndQ = 0; dEndP = 0; dRecPos = 0;
    if (copy) (void) hipStreamDestroy (copy);
    hCounts = 0; dState = 0; dOverflow = 0; dTileEvent = dTileBaseOff = dTileStartOff = 0; dTileBases = dTileStarts = 0; dBases = 0; dRecOff = 0; dPacked = 0;
    tilesCap = basesCap = recCap = packedWords = window = 0; copy = 0; dev = -1;
  }
};
#define TX_MAXDEV 128
static TxBufs gTxs[TX_MAXDEV];                       /* by device: a process whose host threads read a file each on a GPU each parses them side by side */
extern "C" void mgTextReleaseBuffers (void)
{
  int before = -1; if (hipGetDevice (&before) != hipSuccess) { (void) hipGetLastError (); before = -1; }
  for (int dev = 0 ; dev < TX_MAXDEV ; ++dev)
    { std::lock_guard<std::mutex> g (gTxs[dev].lock);
      if (gTxs[dev].dev >= 0) { (void) hipSetDevice (gTxs[dev].dev); gTxs[dev].release (); }
    }
  if (before >= 0) (void) hipSetDevice (before);
}

/* text per window: 128 MiB (two pinned and two device buffers of that size are kept between calls; 64 MiB windows: 31.6 Gbp/s on a
   4 Gbp file, 256 MiB: 34.8), less for a file that is smaller */
static size_t txWindowBytes (size_t fileSize)
{
  const long wk = mgKnobs ()->textWindowKb;                  /* test knob: small windows put window and batch edges everywhere */
  long kb = wk != MG_KNOB_UNSET ? wk : 0;
  size_t w = kb > 0 ? (size_t) kb << 10 : (size_t) 128 << 20;
  if (kb <= 0) while (w > ((size_t) 1 << 20) && w / 2 >= fileSize) w /= 2;
  return (w + TX_TILE - 1) / TX_TILE * TX_TILE;
}
static U64 txBatchBases (U64 asked)
{
  const MgKnobs *kn = mgKnobs ();
  long mbp = kn->fileBatchMbp != MG_KNOB_UNSET ? kn->fileBatchMbp : 1024;
  if (mbp < 1) mbp = 1;
  U64 b = (U64) mbp * 1000000;
  if (asked && kn->fileBatchMbp == MG_KNOB_UNSET) b = asked;
  if (kn->fileBatchBases != MG_KNOB_UNSET && kn->fileBatchBases > 0) b = (U64) kn->fileBatchBases;        /* test knob (as in mg_seqio.c) */
  return b;
}

static int txReserveInner (TxBufs &t, size_t window, U64 basesNeed, U64 recsNeed);
static int txReserve (TxBufs &t, size_t window, U64 basesNeed, U64 recsNeed)
{
  const int rc = txReserveInner (t, window, basesNeed, recsNeed);
  if (rc) t.release ();      /* an allocation failed half way: give back what was made (release () tolerates null pointers) */
  return rc;
}
static int txReserveInner (TxBufs &t, size_t window, U64 basesNeed, U64 recsNeed)
{
  int dev = 0; if (hipGetDevice (&dev) != hipSuccess) return -1;
  if (t.dev >= 0 && (t.dev != dev || t.window < window)) t.release ();      /* (buffers made for a larger window serve a smaller one) */
  if (t.dev < 0)
    { for (int i = 0 ; i < 2 ; ++i)
        { if (hipHostMalloc ((void **) &t.hPin[i], window + 64, hipHostMallocDefault) != hipSuccess) return -1;
          if (hipEventCreateWithFlags (&t.h2dDone[i], hipEventDisableTiming) != hipSuccess) return -1;
          if (hipMalloc ((void **) &t.dText[i], window + 64) != hipSuccess) return -1;
        }
      if (hipHostMalloc ((void **) &t.hCounts, 64, hipHostMallocDefault) != hipSuccess) return -1;
      if (hipMalloc ((void **) &t.dState, sizeof (TxState)) != hipSuccess || hipMalloc ((void **) &t.dOverflow, 4) != hipSuccess) return -1;
      const size_t tiles = window / TX_TILE + 2;
      if (hipMalloc ((void **) &t.dTileEvent, tiles * 8) != hipSuccess || hipMalloc ((void **) &t.dTileBaseOff, tiles * 8) != hipSuccess
          || hipMalloc ((void **) &t.dTileStartOff, tiles * 8) != hipSuccess || hipMalloc ((void **) &t.dTileBases, tiles * 4) != hipSuccess
          || hipMalloc ((void **) &t.dTileStarts, tiles * 4) != hipSuccess
          || hipMalloc ((void **) &t.dTileOffQ, tiles * 8) != hipSuccess || hipMalloc ((void **) &t.dTileQual, tiles * 4) != hipSuccess
          || hipMalloc ((void **) &t.dTq, sizeof (TqState)) != hipSuccess) return -1;
      t.tilesCap = tiles;
      if (hipStreamCreateWithFlags (&t.copy, hipStreamNonBlocking) != hipSuccess) return -1;
      t.dev = dev; t.window = window;
    }
  if (basesNeed > t.basesCap)
    { unsigned char *nb = 0; size_t cap = (size_t) (basesNeed + basesNeed / 4 + (1 << 20)); if (cap < 2 * t.basesCap) cap = 2 * t.basesCap;
      if (hipMalloc ((void **) &nb, cap) != hipSuccess) return -1;
      if (t.dBases) { (void) hipMemcpy (nb, t.dBases, t.basesCap, hipMemcpyDeviceToDevice); (void) hipFree (t.dBases); }
      t.dBases = nb; t.basesCap = cap;
    }
  if (recsNeed > t.recCap)
    { size_t cap = (size_t) (recsNeed + recsNeed / 4 + 4096); if (cap < 2 * t.recCap) cap = 2 * t.recCap;
      U64 **arr[4] = { &t.dRecOff, &t.dEndQ, &t.dEndP, &t.dRecPos };
      for (int i = 0 ; i < 4 ; ++i)
        { U64 *nr = 0;
          if (hipMalloc ((void **) &nr, cap * 8) != hipSuccess) return -1;
          if (*arr[i]) { (void) hipMemcpy (nr, *arr[i], t.recCap * 8, hipMemcpyDeviceToDevice); (void) hipFree (*arr[i]); }
          *arr[i] = nr;
        }
      t.recCap = cap;
    }
  return 0;
}

/* [off, off + n) of the file into dst, by a team of threads (the copy out of the page cache is the cost) */
static bool txReadParallel (int fd, unsigned char *dst, size_t n, off_t off, int nThreads)
{
  if (nThreads < 1) nThreads = 1;
  const size_t slice = ((n + nThreads - 1) / nThreads + 4095) & ~(size_t) 4095;
  volatile int bad = 0;
  auto work = [&] (int t)
    { size_t a = slice * (size_t) t, b = a + slice < n ? a + slice : n;
      while (a < b)
        { ssize_t g = pread (fd, dst + a, b - a, off + (off_t) a);
          if (g <= 0) { bad = 1; return; }
          a += (size_t) g;
        }
    };
  std::vector<std::thread> th;
  int started = 1;
  try { for (int t = 1 ; t < nThreads && slice * (size_t) t < n ; ++t) { th.emplace_back (work, t); ++started; } }
  catch (...) { }
  work (0);
  for (auto &x : th) x.join ();
  for (int t = started ; t < nThreads && slice * (size_t) t < n ; ++t) work (t);      /* threads that could not be started */
  return !bad;
}

static int txHostThreads (void)
{
  const long pk = mgKnobs ()->parseThreads;
  long v = pk != MG_KNOB_UNSET ? pk : 0;
  if (v <= 0) v = mgCpuBudget ();
  if (v < 1) v = 1;
  if (v > 32) v = 32;
  return (int) v;
}

struct TxSink {                                            /* what is done with a batch of complete records */
  int (*fn) (void *ctx, const U32 *dPacked, U64 totalBases, const U64 *dReadOffsets, U32 nReads, const char *idBytes, const U64 *idOff, hipStream_t st);
  void *ctx;
  U64 batchBases = 0;                                      /* != 0: a batch is handed on once it holds this many bases (the knobs, which tests set, come first) */
  U64 batchRecs = 0;                                       /* != 0: ... and this many records -- or the default batch's bases whatever the records (long reads: few lines to format, and a batch's chaining costs its LONGEST read's serial walk: fewer, larger batches) */
  bool wantIds;                                            /* the records' ids (seqio.c:303-304: the header line after its '>' / '@' up to the first white space): id r = idBytes + idOff[r], 0-terminated */
};

/* The ids of the accumulator's records, in record order, copied out of the pinned text windows while they are there: the
 * kernels say where every header starts (FASTA: mgTextEmitKernel's recPos; FASTQ: the byte after a record's last newline,
 * endP + 1), the host copies the few bytes up to the first white space.  An id that runs on into the next window (or starts
 * at its first byte) is finished when that window is in. */
#include <ctype.h>
struct TxIds {
  std::vector<char> bytes; std::vector<U64> off;
  bool open = false; bool skipOne = false;                 /* the last id is not finished; its '@' is the next window's first byte */
  double tLens = 0;                 /* (dev timing) */
  void clear () { bytes.clear (); off.clear (); open = false; skipOne = false; }
  void feed (const unsigned char *p, size_t n)
  { if (skipOne) { if (!n) return; ++p; --n; skipOne = false; }
    size_t i = 0;
    while (i < n && !space (p[i])) ++i;                      /* (not isspace (): that one follows the process's locale; the reference's ids end where the C locale's white space is, seqio.c:303) */
    bytes.insert (bytes.end (), (const char *) p, (const char *) p + i);
    if (i < n) { bytes.push_back (0); open = false; }
  }
  /* a header whose first byte ('>' / '@') is byte `at` of the window (at == n: the next window's first byte) */
  void header (const unsigned char *win, size_t n, size_t at)
  { off.push_back ((U64) bytes.size ()); open = true;
    if (at >= n) { skipOne = true; return; }
    feed (win + at + 1, n - at - 1);
  }
  static inline bool space (unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); }      /* isspace in the C locale */
  /* cnt headers at window bytes at[0 .. cnt), every one of them with its id's end inside the window: lengths by a team of
     threads, places by a prefix over the threads' sums, copies by the team again (a window of short reads starts 4e5 records) */
  void headersTeam (const unsigned char *win, size_t n, const U64 *at, size_t cnt, int nThreads)
  { if (!cnt) return;
    if (nThreads > 16) nThreads = 16;
    if (cnt < 20000 || nThreads < 2) { for (size_t i = 0 ; i < cnt ; ++i) header (win, n, (size_t) at[i]); return; }
    std::vector<U32> len (cnt);
    std::vector<U64> sum ((size_t) nThreads + 1, 0);
    size_t base = 0, first = 0;
    pthread_barrier_t bar; pthread_barrier_init (&bar, 0, (unsigned) nThreads);
    /* one team for both passes (starting and joining fifteen threads costs as much as a pass): lengths; then member 0 makes the room; then copies */
    auto work = [&] (int t)
      { const size_t a = cnt * (size_t) t / nThreads, b = cnt * ((size_t) t + 1) / nThreads; U64 s = 0;
        for (size_t i = a ; i < b ; ++i)
          { if (i + 24 < b) __builtin_prefetch (win + at[i + 24] + 1);      /* a header per 300 bytes of a short-read file: every one is a cache miss */
            const unsigned char *p = win + at[i] + 1, *e = win + n; const unsigned char *q = p;
            while (q < e && !space (*q)) ++q;
            len[i] = (U32) (q - p); s += (U64) (q - p) + 1;
          }
        sum[(size_t) t + 1] = s;
        pthread_barrier_wait (&bar);
        if (t == 0)
          { for (int u = 0 ; u < nThreads ; ++u) sum[(size_t) u + 1] += sum[(size_t) u];
            base = bytes.size (); first = off.size ();
            bytes.resize (base + (size_t) sum[(size_t) nThreads]); off.resize (first + cnt);
          }
        pthread_barrier_wait (&bar);
        size_t o = base + (size_t) sum[(size_t) t];
        for (size_t i = a ; i < b ; ++i)
          { off[first + i] = (U64) o; memcpy (bytes.data () + o, win + at[i] + 1, len[i]); o += len[i]; bytes[o++] = 0; }
      };
    const double tA = txNow ();
    { std::vector<std::thread> th; for (int t = 1 ; t < nThreads ; ++t) th.emplace_back (work, t); work (0); for (auto &x : th) x.join (); }
    pthread_barrier_destroy (&bar);
    tLens += txNow () - tA;
    open = false;
  }
  /* the headers of a window: all but the last through the team (an id ends before the next header starts), the last one -- which may
     run on into the next window -- by header () */
  void headers (const unsigned char *win, size_t n, std::vector<U64> &at, int nThreads)
  { if (at.empty ()) return;
    const size_t cnt = at.size ();
    if (cnt > 1) headersTeam (win, n, at.data (), cnt - 1, nThreads);
    header (win, n, (size_t) at[cnt - 1]);
  }
  void dropFront (size_t nRec)                              /* the first nRec records were flushed */
  { if (!nRec) return;
    const U64 cut = nRec < off.size () ? off[nRec] : (U64) bytes.size ();
    bytes.erase (bytes.begin (), bytes.begin () + (ptrdiff_t) cut);
    off.erase (off.begin (), off.begin () + (ptrdiff_t) (nRec < off.size () ? nRec : off.size ()));
    for (auto &o : off) o -= cut;
  }
};

/* one batch: records [0, nRec) of the accumulator, bases [0, total); dRecOff[nRec] == total.  Returns 0 or an error */
static int txFlush (TxBufs &t, const TxSink &sink, U64 total, U64 nRec, hipStream_t st, TxIds *ids = 0)
{
  if (!nRec) return 0;
  if (sink.wantIds && (!ids || ids->off.size () < nRec || (ids->off.size () == nRec && ids->open)))
    { mgSetError ("device text parser: %llu records but %llu ids", (unsigned long long) nRec, (unsigned long long) (ids ? ids->off.size () : 0)); return -1; }
  const size_t nw = mgPackedWords (total);
  if (nw > t.packedWords)
    { (void) hipFree (t.dPacked); t.dPacked = 0; t.packedWords = 0;
      if (hipMalloc ((void **) &t.dPacked, (nw + nw / 4) * 4) != hipSuccess) return -1;
      t.packedWords = nw + nw / 4;
    }
  if (mgLaunchPack (t.dBases, total, t.dPacked, st)) return -1;
  return sink.fn (sink.ctx, t.dPacked, total, t.dRecOff, (U32) nRec, sink.wantIds ? ids->bytes.data () : (const char *) 0, sink.wantIds ? ids->off.data () : (const U64 *) 0, st);
}


/* where a window's record headers start, from the device to the host: a kernel writes them straight into page-locked host memory
   behind the emit kernel (hipMemcpy from the device took its turn behind the NEXT window's 128 MiB on its way to the device: 2 ms a
   window, as long as extracting the ids) */
__global__ void mgTextCopyOutKernel (const U64 *__restrict__ src, U64 *__restrict__ dstHost, U64 n)
{ for (U64 i = (U64) blockIdx.x * blockDim.x + threadIdx.x ; i < n ; i += (U64) gridDim.x * blockDim.x) dstHost[i] = src[i]; }
static bool txHeadersLaunch (TxBufs &t, const U64 *dSrc, size_t n, hipStream_t st)
{
  if (!n) return true;
  if (n > t.hdrCap)
    { if (t.hHdr) (void) hipHostFree (t.hHdr);
      t.hHdr = 0; t.hdrCap = 0; t.dHdr = 0;
      if (hipHostMalloc ((void **) &t.hHdr, (n + n / 4) * 8, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess
          || hipHostGetDevicePointer ((void **) &t.dHdr, t.hHdr, 0) != hipSuccess) return false;
      t.hdrCap = n + n / 4;
    }
  unsigned grid = (unsigned) ((n + 255) / 256); if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL (mgTextCopyOutKernel, dim3 (grid), dim3 (256), 0, st, dSrc, t.dHdr, (U64) n);
  return hipGetLastError () == hipSuccess;
}

static int txParseFastq (int fd, size_t fileSize, TxBufs &t, const TxSink &sink, U64 *nSeqOut, U64 *totLenOut, U64 *resumeOff, U64 *resumeLine);

/* the file through the device parser; every batch of complete records goes to sink.  Returns 0, -1 (error: mgLastError), or -2
 * (not a file this path takes: the caller uses the host parser).  *nSeqOut / *totLenOut: records and bases of the file. */
static int txParseFile (const char *filename, const TxSink &sink, U64 *nSeqOut, U64 *totLenOut, U64 *resumeOff = 0, U64 *resumeLine = 0)
{
  if (mgEnsureDevice ()) return -2;
  if (mgKnobs ()->textHost == 1) return -2;      /* test knob: the host parser */
  int fd = open (filename, O_RDONLY);
  if (fd < 0) return -2;
  struct stat sb;
  if (fstat (fd, &sb) || !S_ISREG (sb.st_mode) || sb.st_size < 2) { close (fd); return -2; }
  const size_t fileSize = (size_t) sb.st_size;
  unsigned char first = 0, lastc = 0;
  if (pread (fd, &first, 1, 0) != 1 || pread (fd, &lastc, 1, (off_t) fileSize - 1) != 1 || (first != '>' && first != '@') || lastc != '\n') { close (fd); return -2; }
  if (first == '>')                                       /* a FASTA file whose LAST line is a header: the reference reports the record as incomplete and does not
                                                             return it (seqio.c:213-217,314) -- left to the host parser, which says so with the line number */
    { unsigned char tail[4096]; const size_t tn = fileSize < sizeof (tail) ? fileSize : sizeof (tail);
      if (pread (fd, tail, tn, (off_t) (fileSize - tn)) != (ssize_t) tn) { close (fd); return -2; }
      size_t i = tn - 1;                                   /* the final newline */
      while (i > 0 && tail[i - 1] != '\n') --i;            /* start of the last line (or of the tail: a header longer than that is no record id line we want to judge here) */
      if (tail[i] == '>' && (i > 0 || tn == fileSize)) { close (fd); return -2; }
    }

  int curDev = 0;
  if (hipGetDevice (&curDev) != hipSuccess || curDev < 0 || curDev >= TX_MAXDEV) { (void) hipGetLastError (); close (fd); return -2; }      /* (no room kept for that device: the host parser) */
  TxBufs &t = gTxs[curDev];
  std::lock_guard<std::mutex> g (t.lock);
  if (first == '@')
    { const int rq = txParseFastq (fd, fileSize, t, sink, nSeqOut, totLenOut, resumeOff, resumeLine);
      close (fd);
      return rq;
    }
  const size_t window = txWindowBytes (fileSize);
  const U64 batch = txBatchBases (sink.batchBases), bigBatch = batch > txBatchBases (0) ? batch : txBatchBases (0);
  hipStream_t st = 0;
  int rc = -1;
  U64 nSeq = 0, totLen = 0;
  do {
    TxClock ck; ck.t0 = txNow ();
    if (txReserve (t, window, 1 << 20, 4096)) { mgSetError ("device text parser: allocation failed"); break; }      /* the accumulators grow to what the windows' counts ask for */
    ck.lap (ck.reserve);
    TxState init; init.lastEvent = 0; init.accBases = 0; init.accRecs = 0;
    if (hipMemcpy (t.dState, &init, sizeof (init), hipMemcpyHostToDevice) != hipSuccess || hipMemset (t.dOverflow, 0, 4) != hipSuccess) break;
    const int nThreads = txHostThreads ();
    TxIds ids; std::vector<U64> hdr;
    U64 accBases = 0, accRecs = 0;                          /* the accumulator as of the last synchronised window */
    size_t off = 0; int w = 0;
    U32 prevByte = '\n';
    size_t nCur = fileSize < window ? fileSize : window;
    ck.lap (ck.flush);
    if (!txReadParallel (fd, t.hPin[0], nCur, 0, nThreads)) { mgSetError ("device text parser: read failed"); break; }
    ck.lap (ck.read);
    if (hipMemcpyAsync (t.dText[0], t.hPin[0], nCur, hipMemcpyHostToDevice, t.copy) != hipSuccess || hipEventRecord (t.h2dDone[0], t.copy) != hipSuccess) break;
    bool failed = false;
    while (nCur)
      { const int cur = w & 1, oth = cur ^ 1;
        /* the window crosses the link and is parsed ... */
        const U64 nTiles = (nCur + TX_TILE - 1) / TX_TILE;
        if (hipStreamWaitEvent (st, t.h2dDone[cur], 0) != hipSuccess) { failed = true; break; }      /* the window's copy was started as soon as it was read (copy stream) */
        hipLaunchKernelGGL (mgTextEventKernel, dim3 ((unsigned) nTiles), dim3 (TX_THREADS), 0, st, t.dText[cur], (U64) nCur, (U64) off, prevByte, t.dTileEvent);
        hipLaunchKernelGGL (mgTextStateScanKernel, dim3 (1), dim3 (1024), 0, st, t.dTileEvent, nTiles, t.dState);
        hipLaunchKernelGGL (mgTextCountKernel, dim3 ((unsigned) nTiles), dim3 (TX_THREADS), 0, st, t.dText[cur], (U64) nCur, (U64) off, prevByte,
                            t.dTileEvent, t.dTileBases, t.dTileStarts);
        hipLaunchKernelGGL (mgTextOffsetScanKernel, dim3 (1), dim3 (1024), 0, st, t.dTileBases, t.dTileStarts, nTiles, t.dTileBaseOff, t.dTileStartOff, t.dState, t.hCounts);
        if (hipGetLastError () != hipSuccess) { failed = true; break; }
        const U32 prevOfWindow = prevByte;
        prevByte = t.hPin[cur][nCur - 1];
        /* ... while the host reads the next one into the other pinned buffer (whose last copy to the device must be over) */
        const size_t offNext = off + nCur;
        size_t nNext = fileSize - offNext < window ? fileSize - offNext : window;
        if (nNext)
          { if (w >= 1 && hipEventSynchronize (t.h2dDone[oth]) != hipSuccess) { failed = true; break; }
            ck.lap (ck.flush);
            if (!txReadParallel (fd, t.hPin[oth], nNext, (off_t) offNext, nThreads)) { mgSetError ("device text parser: read failed"); failed = true; break; }
            ck.lap (ck.read);
            /* across the link at once, beside this window's kernels (the other device buffer is free: its window's kernels were waited for) */
            if (hipMemcpyAsync (t.dText[oth], t.hPin[oth], nNext, hipMemcpyHostToDevice, t.copy) != hipSuccess
                || hipEventRecord (t.h2dDone[oth], t.copy) != hipSuccess) { failed = true; break; }
          }
        /* the counts are in: room for exactly what this window adds, then the bases and the offsets are written */
        if (hipStreamSynchronize (st) != hipSuccess) { failed = true; break; }
        const U64 recsBefore = accRecs;
        accBases = t.hCounts[0]; accRecs = t.hCounts[1];
        ck.lap (ck.wait);
        if (accBases + 64 > t.basesCap || accRecs + 2 > t.recCap)
          if (txReserve (t, window, accBases + 64, accRecs + 2)) { mgSetError ("device text parser: allocation failed"); failed = true; break; }
        hipLaunchKernelGGL (mgTextEmitKernel, dim3 ((unsigned) nTiles), dim3 (TX_THREADS), 0, st, t.dText[cur], (U64) nCur, (U64) off, prevOfWindow,
                            t.dTileEvent, t.dTileBaseOff, t.dTileStartOff, t.dBases, (U64) t.basesCap, t.dRecOff, (U64) t.recCap, t.dOverflow,
                            sink.wantIds ? t.dRecPos : (U64 *) 0);
        if (hipGetLastError () != hipSuccess) { failed = true; break; }
        if (sink.wantIds && !txHeadersLaunch (t, t.dRecPos + recsBefore, (size_t) (accRecs - recsBefore), st)) { failed = true; break; }
        if (hipStreamSynchronize (st) != hipSuccess) { failed = true; break; }
        if (sink.wantIds)                                  /* the ids of the records that start in this window, while its text is in the pinned buffer */
          { if (ids.open) ids.feed (t.hPin[cur], nCur);
            const U64 nNew = accRecs - recsBefore;
            hdr.assign (t.hHdr, t.hHdr + nNew);
            for (auto &h : hdr) h -= (U64) off;
            ids.headers (t.hPin[cur], nCur, hdr, nThreads);
          }
        const bool eof = !nNext;
        if (eof || accBases >= bigBatch || (accBases >= batch && accRecs >= sink.batchRecs))
          { /* complete records: all of them at the end of the file, otherwise all but the one still open */
            U64 nRec = eof ? accRecs : accRecs - 1, total = accBases;
            if (!eof && hipMemcpy (&total, t.dRecOff + nRec, 8, hipMemcpyDeviceToHost) != hipSuccess) { failed = true; break; }
            if (eof && hipMemcpy (t.dRecOff + nRec, &total, 8, hipMemcpyHostToDevice) != hipSuccess) { failed = true; break; }
            U32 ov = 0; if (hipMemcpy (&ov, t.dOverflow, 4, hipMemcpyDeviceToHost) != hipSuccess || ov) { mgSetError ("device text parser: accumulator overflow"); failed = true; break; }
            if (nRec)
              { if (txFlush (t, sink, total, nRec, st, &ids)) { failed = true; break; }
                nSeq += nRec; totLen += total;
                if (sink.wantIds) ids.dropFront ((size_t) nRec);
                /* the open record's bases move to the front; it becomes record 0 of the next batch */
                const U64 carry = accBases - total;
                if (!eof)
                  { txCarry (t.dBases, total, carry, st);
                    TxState ns; ns.lastEvent = 0; ns.accBases = carry; ns.accRecs = 1;
                    U64 zero = 0;
                    /* (lastEvent stays what it is on the device: only the two counters change) */
                    if (hipMemcpyAsync ((char *) t.dState + offsetof (TxState, accBases), &ns.accBases, 16, hipMemcpyHostToDevice, st) != hipSuccess
                        || hipMemcpyAsync (t.dRecOff, &zero, 8, hipMemcpyHostToDevice, st) != hipSuccess
                        || hipStreamSynchronize (st) != hipSuccess) { failed = true; break; }
                    accBases = carry; accRecs = 1;
                  }
              }
          }
        off = offNext; nCur = nNext; ++w;
      }
    ck.lap (ck.flush);
    if (txTiming ()) fprintf (stderr, "  [device text] FASTA: reserve %.3f s, read %.3f, wait for the device %.3f, emit + flush + rest %.3f\n", ck.reserve, ck.read, ck.wait, ck.flush);
    if (failed) { if (!mgLastError ()[0]) mgSetError ("device text parser: HIP failure (%s)", hipGetErrorString (hipGetLastError ())); break; }
    rc = 0;
  } while (0);
  close (fd);
  if (nSeqOut) *nSeqOut = nSeq;
  if (totLenOut) *totLenOut = totLen;
  return rc;
}

/* the FASTQ file through the device parser.  0 = the whole file; -1 = error; -3 = the text from byte *resumeOff on (a record start,
 * line *resumeLine) is left to the host parser: a rule of the format is broken somewhere after it, or the file ends in the middle of
 * a record -- the host parser then reports what the reference reports (seqio.c:213-217,326-339) */
static int txParseFastq (int fd, size_t fileSize, TxBufs &t, const TxSink &sink, U64 *nSeqOut, U64 *totLenOut, U64 *resumeOff, U64 *resumeLine)
{
  const size_t window = txWindowBytes (fileSize);
  const U64 batch = txBatchBases (sink.batchBases), bigBatch = batch > txBatchBases (0) ? batch : txBatchBases (0);
  hipStream_t st = 0;
  int rc = -1;
  U64 nSeq = 0, totLen = 0, resume = 0;
  do {
    TxClock ck; ck.t0 = txNow ();
    if (txReserve (t, window, 1 << 20, 4096)) { mgSetError ("device text parser: allocation failed"); break; }
    ck.lap (ck.reserve);
    TqState init; memset (&init, 0, sizeof (init));
    U64 zero = 0;
    if (hipMemcpy (t.dTq, &init, sizeof (init), hipMemcpyHostToDevice) != hipSuccess || hipMemset (t.dOverflow, 0, 4) != hipSuccess
        || hipMemcpy (t.dRecOff, &zero, 8, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy (t.dEndQ, &zero, 8, hipMemcpyHostToDevice) != hipSuccess) break;
    const int nThreads = txHostThreads ();
    TxIds ids; std::vector<U64> hdr;
    U64 accBases = 0, accRecs = 0, accQual = 0;
    size_t off = 0; int w = 0;
    U32 prevByte = '\n';
    size_t nCur = fileSize < window ? fileSize : window;
    ck.lap (ck.flush);
    if (!txReadParallel (fd, t.hPin[0], nCur, 0, nThreads)) { mgSetError ("device text parser: read failed"); break; }
    ck.lap (ck.read);
    if (hipMemcpyAsync (t.dText[0], t.hPin[0], nCur, hipMemcpyHostToDevice, t.copy) != hipSuccess || hipEventRecord (t.h2dDone[0], t.copy) != hipSuccess) break;
    bool failed = false, handOver = false;
    while (nCur)
      { const int cur = w & 1, oth = cur ^ 1;
        const U64 nTiles = (nCur + TX_TILE - 1) / TX_TILE;
        if (hipStreamWaitEvent (st, t.h2dDone[cur], 0) != hipSuccess) { failed = true; break; }      /* the window's copy was started as soon as it was read (copy stream) */
        hipLaunchKernelGGL (mgTextNewlineKernel, dim3 ((unsigned) nTiles), dim3 (TX_THREADS), 0, st, t.dText[cur], (U64) nCur, t.dTileEvent);
        hipLaunchKernelGGL (mgTextLineScanKernel, dim3 (1), dim3 (1024), 0, st, t.dTileEvent, nTiles, t.dTq);
        hipLaunchKernelGGL (mgTextFastqCountKernel, dim3 ((unsigned) nTiles), dim3 (TX_THREADS), 0, st, t.dText[cur], (U64) nCur, prevByte, t.dTileEvent,
                            t.dTileBases, t.dTileQual, t.dTileStarts, t.dTq);
        hipLaunchKernelGGL (mgTextFastqOffsetKernel, dim3 (1), dim3 (1024), 0, st, t.dTileBases, t.dTileQual, t.dTileStarts, nTiles,
                            t.dTileBaseOff, t.dTileOffQ, t.dTileStartOff, t.dTq, t.hCounts);
        if (hipGetLastError () != hipSuccess) { failed = true; break; }
        const U32 prevOfWindow = prevByte;
        prevByte = t.hPin[cur][nCur - 1];
        const size_t offNext = off + nCur;
        size_t nNext = fileSize - offNext < window ? fileSize - offNext : window;
        if (nNext)
          { if (w >= 1 && hipEventSynchronize (t.h2dDone[oth]) != hipSuccess) { failed = true; break; }
            ck.lap (ck.flush);
            if (!txReadParallel (fd, t.hPin[oth], nNext, (off_t) offNext, nThreads)) { mgSetError ("device text parser: read failed"); failed = true; break; }
            ck.lap (ck.read);
            /* across the link at once, beside this window's kernels (the other device buffer is free: its window's kernels were waited for) */
            if (hipMemcpyAsync (t.dText[oth], t.hPin[oth], nNext, hipMemcpyHostToDevice, t.copy) != hipSuccess
                || hipEventRecord (t.h2dDone[oth], t.copy) != hipSuccess) { failed = true; break; }
          }
        if (hipStreamSynchronize (st) != hipSuccess) { failed = true; break; }
        const U64 recsBefore = accRecs;
        accBases = t.hCounts[0]; accRecs = t.hCounts[1]; accQual = t.hCounts[2];
        ck.lap (ck.wait);
        if (accBases + 64 > t.basesCap || accRecs + 3 > t.recCap)
          if (txReserve (t, window, accBases + 64, accRecs + 3)) { mgSetError ("device text parser: allocation failed"); failed = true; break; }
        hipLaunchKernelGGL (mgTextFastqEmitKernel, dim3 ((unsigned) nTiles), dim3 (TX_THREADS), 0, st, t.dText[cur], (U64) nCur, (U64) off, prevOfWindow, t.dTileEvent,
                            t.dTileBaseOff, t.dTileOffQ, t.dTileStartOff, t.dBases, (U64) t.basesCap, t.dRecOff, t.dEndQ, t.dEndP, (U64) t.recCap, t.dOverflow);
        if (hipGetLastError () != hipSuccess) { failed = true; break; }
        if (sink.wantIds && !txHeadersLaunch (t, t.dEndP + recsBefore + 1, (size_t) (accRecs - recsBefore), st)) { failed = true; break; }
        if (hipStreamSynchronize (st) != hipSuccess) { failed = true; break; }
        const U64 nlCount = t.hCounts[3];
        bool bad = t.hCounts[4] != 0;
        if (sink.wantIds && !bad)                         /* a header starts at the file's first byte and after every record's last newline */
          { ck.lap (ck.flush);
            if (ids.open) ids.feed (t.hPin[cur], nCur);
            const U64 nNew = accRecs - recsBefore;
            hdr.assign (t.hHdr, t.hHdr + nNew);
            for (auto &h : hdr) h = h + 1 - (U64) off;     /* the byte after a record's last newline, in this window (or its end: the next window's first byte) */
            if (!hdr.empty () && hdr.back () + (U64) off >= (U64) fileSize) hdr.pop_back ();      /* the file's last record: no header follows */
            if (!off) hdr.insert (hdr.begin (), (U64) 0);
            ids.headers (t.hPin[cur], nCur, hdr, nThreads);
            ck.lap (ck.ids);
          }
        if (!bad && accRecs > recsBefore)                  /* the records this window completed: sequence and quality lines of one length? */
          { const U64 nNew = accRecs - recsBefore;
            hipLaunchKernelGGL (mgTextFastqCheckKernel, dim3 ((unsigned) ((nNew + 255) / 256)), dim3 (256), 0, st, t.dRecOff, t.dEndQ, recsBefore + 1, accRecs, t.dTq);
            TqState now;
            if (hipMemcpy (&now, t.dTq, sizeof (now), hipMemcpyDeviceToHost) != hipSuccess) { failed = true; break; }
            bad = now.bad != 0;
          }
        U32 ov = 0; if (hipMemcpy (&ov, t.dOverflow, 4, hipMemcpyDeviceToHost) != hipSuccess || ov) { mgSetError ("device text parser: accumulator overflow"); failed = true; break; }
        const bool eof = !nNext;
        U64 tail[3] = { 0, 0, 0 };                         /* bases, quality bytes, file position at the end of the last completed record */
        if (accRecs && (hipMemcpy (&tail[0], t.dRecOff + accRecs, 8, hipMemcpyDeviceToHost) != hipSuccess
                        || hipMemcpy (&tail[1], t.dEndQ + accRecs, 8, hipMemcpyDeviceToHost) != hipSuccess
                        || hipMemcpy (&tail[2], t.dEndP + accRecs, 8, hipMemcpyDeviceToHost) != hipSuccess)) { failed = true; break; }
        if (bad || (eof && ((nlCount & 3) || accBases != (accRecs ? tail[0] : 0) || accQual != (accRecs ? tail[1] : 0))))
          { handOver = true; break; }                      /* nothing of the accumulator has been added: the host parser starts at its first record */
        if ((eof || accBases >= bigBatch || (accBases >= batch && accRecs >= sink.batchRecs)) && accRecs)
          { ck.lap (ck.flush);
            if (txFlush (t, sink, tail[0], accRecs, st, &ids)) { failed = true; break; }
            ck.lap (ck.sink);
            nSeq += accRecs; totLen += tail[0];
            if (sink.wantIds) ids.dropFront ((size_t) accRecs);
            resume = tail[2] + 1;
            if (!eof)
              { const U64 carry = accBases - tail[0];
                txCarry (t.dBases, tail[0], carry, st);
                U64 counters[3] = { carry, accQual - tail[1], 0 };               /* accBases, accQual, accRecs */
                if (hipMemcpyAsync ((char *) t.dTq + offsetof (TqState, accBases), counters, 24, hipMemcpyHostToDevice, st) != hipSuccess
                    || hipStreamSynchronize (st) != hipSuccess) { failed = true; break; }
                accBases = carry; accQual -= tail[1]; accRecs = 0;
              }
          }
        off = offNext; nCur = nNext; ++w;
      }
    ck.lap (ck.flush);
    if (txTiming ()) fprintf (stderr, "  [device text] FASTQ: reserve %.3f s, read %.3f, wait for the device %.3f, record ids %.3f, pack + sink %.3f, emit + rest %.3f\n", ck.reserve, ck.read, ck.wait, ck.ids, ck.sink, ck.flush);
    if (txTiming ()) fprintf (stderr, "  [device text] ids: the team's two passes %.3f (%d threads)\n", ids.tLens, nThreads);
    if (failed) { if (!mgLastError ()[0]) mgSetError ("device text parser: HIP failure (%s)", hipGetErrorString (hipGetLastError ())); break; }
    rc = handOver ? -3 : 0;
  } while (0);
  if (nSeqOut) *nSeqOut = nSeq;
  if (totLenOut) *totLenOut = totLen;
  if (resumeOff) *resumeOff = resume;
  if (resumeLine) *resumeLine = 4 * nSeq + 1;
  return rc;
}

/* ---- sinks ---- */

struct TxAddCtx { Modset *ms; U64 totHash; };
static int txAddSink (void *v, const U32 *dPacked, U64 total, const U64 *dOff, U32 nReads, const char *, const U64 *, hipStream_t st)
{
  TxAddCtx *c = (TxAddCtx *) v;
  U64 nHash = 0;
  if (mgAddReadsDevice (c->ms, dPacked, total, dOff, nReads, &nHash, (void *) st) != MG_OK) return -1;
  c->totHash += nHash;
  return 0;
}

/* modutils.c:33-51 with the text parsed on the device.  0: done (counts filled in); -1: error; -2: not a file for this path */
extern "C" int mgAddSequenceFileDevice (Modset *ms, const char *filename, U64 *nSeq, U64 *totLen, U64 *totHash, U64 *resumeOff, U64 *resumeLine)
{
  TxAddCtx c; c.ms = ms; c.totHash = 0;
  TxSink sink; sink.fn = txAddSink; sink.ctx = &c; sink.wantIds = false;
  const int rc = txParseFile (filename, sink, nSeq, totLen, resumeOff, resumeLine);
  if (totHash) *totHash = c.totHash;
  return rc;
}

/* modmap's file entry points (mgReferenceFastaRead, mgQueryFile: modmap.c:93-134,188-281): every batch of complete records,
   device resident, with the records' ids, to a C callback.  0 / -1 / -2 / -3 as txParseFile */
struct TxCbCtx { MgTextBatchFn fn; void *ctx; };
static int txCbSink (void *v, const U32 *dPacked, U64 total, const U64 *dOff, U32 nReads, const char *idBytes, const U64 *idOff, hipStream_t st)
{ TxCbCtx *c = (TxCbCtx *) v; return c->fn (c->ctx, dPacked, total, dOff, nReads, idBytes, idOff, (void *) st); }
extern "C" int mgTextForEachBatchDevice (const char *filename, MgTextBatchFn fn, void *ctx, U64 batchBases, U64 batchRecs, U64 *nSeq, U64 *totLen, U64 *resumeOff, U64 *resumeLine)
{
  TxCbCtx c; c.fn = fn; c.ctx = ctx;
  TxSink sink; sink.fn = txCbSink; sink.ctx = &c; sink.wantIds = true; sink.batchBases = batchBases; sink.batchRecs = batchRecs;
  return txParseFile (filename, sink, nSeq, totLen, resumeOff, resumeLine);
}

/* test hook: the device parser's records as host arrays (bases 0..3 one per byte, offsets[nSeq + 1]), malloc()ed */
struct TxHostCtx { std::vector<unsigned char> bases; std::vector<int64_t> offs; };
static int txHostSink (void *v, const U32 *dPacked, U64 total, const U64 *dOff, U32 nReads, const char *idBytes, const U64 *idOff, hipStream_t st)
{
  TxHostCtx *c = (TxHostCtx *) v;
  std::vector<U64> o ((size_t) nReads + 1);
  if (hipMemcpy (o.data (), dOff, ((size_t) nReads + 1) * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  unsigned char *dB = 0;
  if (hipMalloc ((void **) &dB, total ? total : 16) != hipSuccess) return -1;
  int rc = -1;
  if (mgLaunchUnpack (dPacked, total, dB, st) == MG_OK && hipStreamSynchronize (st) == hipSuccess)
    { const size_t at = c->bases.size ();
      c->bases.resize (at + total);
      if (!total || hipMemcpy (c->bases.data () + at, dB, total, hipMemcpyDeviceToHost) == hipSuccess)
        { if (c->offs.empty ()) c->offs.push_back (0);
          for (U32 r = 1 ; r <= nReads ; ++r) c->offs.push_back ((int64_t) (at + o[r]));
          rc = 0;
        }
    }
  (void) hipFree (dB);
  return rc;
}
extern "C" int mgTextParseFileDevice (const char *filename, char **basesOut, int64_t **offsetsOut, int64_t *nSeqOut)
{
  TxHostCtx c;
  TxSink sink; sink.fn = txHostSink; sink.ctx = &c; sink.wantIds = false;
  U64 nSeq = 0, totLen = 0;
  const int rc = txParseFile (filename, sink, &nSeq, &totLen);
  if (rc) return rc == -3 ? -2 : rc;                       /* (FASTQ handed back to the host parser in the middle: nothing of it is returned here) */
  if (c.offs.empty ()) c.offs.push_back (0);
  char *b = (char *) malloc (c.bases.size () + 1); int64_t *o = (int64_t *) malloc (c.offs.size () * sizeof (int64_t));
  if (!b || !o) { free (b); free (o); return -1; }
  memcpy (b, c.bases.data (), c.bases.size ()); memcpy (o, c.offs.data (), c.offs.size () * sizeof (int64_t));
  *basesOut = b; *offsetsOut = o; *nSeqOut = (int64_t) nSeq;
  return 0;
}
