/* mg_synth.hip — deterministic synthetic reads generated directly in HBM.
 * Not part of the reference: the benchmark's inputs.  Counter-based, so the same data can be
 * regenerated on the host (modimizer_amd/synth.py) for parity checks.
 *   genome base g      = splitmix64 (seed ^ g*GOLD) >> 62
 *   read base          = genome[start+o]            (strand 0)
 *                        3 - genome[start+len-1-o]  (strand 1: reverse complement)
 *   substitution at global base ordinal q when u = splitmix64 (seedE ^ q*GOLD) < errRate*2^64:
 *                        base = (base + 1 + (u & 0xffff) % 3) & 3
 */
#include "mg_common.h"

#define MG_GOLD 0x9E3779B97F4A7C15ull

__host__ __device__ __forceinline__ U64 mgSplitmix64 (U64 x)
{
  x += MG_GOLD;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__global__ void mgSynthGenomeKernel (U32 *__restrict__ words, U64 nBases, U64 nWords, U64 seed)
{
  U64 i = (U64) blockIdx.x * blockDim.x + threadIdx.x;
  const U64 stride = (U64) gridDim.x * blockDim.x;
  for ( ; i < nWords ; i += stride)
    { U32 w = 0;
#pragma unroll
      for (int j = 0 ; j < 16 ; ++j)
        { U64 g = i * 16 + j;
          if (g < nBases) w |= (U32) (mgSplitmix64 (seed ^ (g * MG_GOLD)) >> 62) << (30 - 2 * j);
        }
      words[i] = w;
    }
}

__device__ __forceinline__ U32 mgBaseAt (const U32 *__restrict__ words, U64 g)
{ return (words[g >> 4] >> (30 - 2 * (g & 15))) & 3; }

__global__ void mgSynthReadsKernel (const U32 *__restrict__ genome, U64 genomeBases,
                                    const U64 *__restrict__ readStart, const U64 *__restrict__ readOff,
                                    const U8 *__restrict__ strand, U32 nReads, U64 totalBases, U64 nWords,
                                    U64 errThresh, U64 seed, U32 *__restrict__ out)
{
  U64 i = (U64) blockIdx.x * blockDim.x + threadIdx.x;
  const U64 stride = (U64) gridDim.x * blockDim.x;
  for ( ; i < nWords ; i += stride)
    { U32 w = 0;
      U64 q0 = i * 16;
      if (q0 < totalBases)
        { U32 lo = 0, hi = nReads - 1;             /* read containing q0 */
          while (lo < hi) { U32 mid = lo + (hi - lo + 1) / 2; if (readOff[mid] <= q0) lo = mid; else hi = mid - 1; }
          U32 r = lo;
          for (int j = 0 ; j < 16 ; ++j)
            { U64 q = q0 + j;
              if (q >= totalBases) break;
              while (readOff[r + 1] <= q) ++r;
              U64 o = q - readOff[r], len = readOff[r + 1] - readOff[r];
              U32 b = strand[r] ? 3 - mgBaseAt (genome, readStart[r] + len - 1 - o)
                                : mgBaseAt (genome, readStart[r] + o);
              U64 u = mgSplitmix64 (seed ^ (q * MG_GOLD));
              if (u < errThresh) b = (b + 1 + (U32) ((u & 0xffff) % 3)) & 3;
              w |= b << (30 - 2 * j);
            }
        }
      out[i] = w;
    }
}

MgStatus mgLaunchSynthGenome (U32 *dPacked, U64 nBases, U64 seed, hipStream_t st)
{
  U64 nWords = (U64) mgPackedWords (nBases);
  U64 blocks = (nWords + 255) / 256; if (blocks > 16384) blocks = 16384;
  MG_LAUNCH (MG_K_SYNTH_GENOME, st, mgSynthGenomeKernel, dim3 ((unsigned) blocks), dim3 (256), 0, st, dPacked, nBases, nWords, seed);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}

MgStatus mgLaunchSynthReads (const U32 *dGenome, U64 genomeBases, const U64 *dReadStart,
                             const U64 *dReadOffsets, const U8 *dStrand, U32 nReads, U64 totalBases,
                             double errRate, U64 seed, U32 *dOut, hipStream_t st)
{
  if (!nReads) return MG_OK;
  U64 nWords = (U64) mgPackedWords (totalBases);
  U64 thresh = errRate <= 0 ? 0 : (errRate >= 1 ? ~0ull : (U64) (errRate * 18446744073709551616.0));
  U64 blocks = (nWords + 255) / 256; if (blocks > 16384) blocks = 16384;
  MG_LAUNCH (MG_K_SYNTH_READS, st, mgSynthReadsKernel, dim3 ((unsigned) blocks), dim3 (256), 0, st,
                      dGenome, genomeBases, dReadStart, dReadOffsets, dStrand, nReads, totalBases, nWords, thresh, seed, dOut);
  MG_HIP (hipGetLastError ());
  return MG_OK;
}
