// dev microbenchmark: the WRITE pattern of a radix partition pass, with and without line-aligned reservations.
// 256 bins, each with its region and a cursor.  A workgroup of 1024 threads takes "sub-chunks" of 8192 elements: per bin a count
// (natural: 32 +- 8, what a counting sort of random digits gives; aligned: exactly 32), one returning atomicAdd on the bin's cursor,
// then the 8192 staged elements leave in bin order, consecutive lanes writing consecutive places of a run -- as mgPartScatterKernel
// does.  The input is read in order (8 bytes per element) so that the traffic is that of a pass.  Rates: bytes read + written.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned long long u64;
typedef unsigned int u32;
#define SUB 8192
#define BINS 256
__device__ __forceinline__ u64 mix (u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

template <bool ALIGNED>
__global__ __launch_bounds__ (1024) void part (const u64 *__restrict__ in, u64 *__restrict__ out, u64 nSub, const u64 *__restrict__ binStart, u64 *cursor)
{
  __shared__ u32 sCnt[BINS], sOff[BINS + 1];
  __shared__ u64 sBase[BINS];
  __shared__ unsigned char sBin[SUB];
  const int tid = threadIdx.x;
  for (u64 s = blockIdx.x ; s < nSub ; s += gridDim.x)
    { u64 v[8];
#pragma unroll
      for (int j = 0 ; j < 8 ; ++j) v[j] = in[s * SUB + (u64) j * 1024 + tid];
      if (tid < BINS)
        { u32 c = 32;
          if (!ALIGNED) c = 24 + (u32) (mix (s * BINS + tid) % 17);          /* mean 32 */
          sCnt[tid] = c;
        }
      __syncthreads ();
      if (tid == 0) { u32 a = 0; for (int b = 0 ; b < BINS ; ++b) { sOff[b] = a; a += sCnt[b]; } sOff[BINS] = a; }      /* (a serial scan: not what is measured... kept short) */
      if (tid < BINS) sBase[tid] = binStart[tid] + atomicAdd (&cursor[tid * 16], (u64) sCnt[tid]);
      __syncthreads ();
      const u32 total = sOff[BINS] < SUB ? sOff[BINS] : SUB;
      if (tid < BINS) for (u32 p = sOff[tid] ; p < sOff[tid + 1] && p < SUB ; ++p) sBin[p] = (unsigned char) tid;
      __syncthreads ();
#pragma unroll
      for (int j = 0 ; j < 8 ; ++j)
        { const u32 p = (u32) j * 1024 + tid;
          if (p < total) { const u32 b = sBin[p]; out[sBase[b] + (p - sOff[b])] = v[j]; }
        }
      __syncthreads ();
    }
}

template <bool ALIGNED> static float run (const u64 *in, u64 *out, u64 nSub, const u64 *binStart, u64 *cursor, int grid)
{
  hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
  float best = 1e9f;
  for (int rep = 0 ; rep < 4 ; ++rep)
    { hipMemset (cursor, 0, BINS * 16 * 8);
      hipDeviceSynchronize ();
      hipEventRecord (e0);
      hipLaunchKernelGGL ((part<ALIGNED>), dim3 (grid), dim3 (1024), 0, 0, in, out, nSub, binStart, cursor);
      hipEventRecord (e1); hipEventSynchronize (e1);
      float ms; hipEventElapsedTime (&ms, e0, e1); if (rep && ms < best) best = ms;
    }
  return best;
}

int main ()
{
  const u64 n = (u64) 156 << 20, nSub = n / SUB;                 /* about config 2's modimizers */
  const u64 region = (n / BINS + n / BINS / 4 + 4096) & ~(u64) 15;
  u64 *in, *out, *binStart, *cursor;
  if (hipMalloc (&in, n * 8) != hipSuccess || hipMalloc (&out, region * BINS * 8) != hipSuccess || hipMalloc (&binStart, BINS * 8) != hipSuccess
      || hipMalloc (&cursor, BINS * 16 * 8) != hipSuccess) { printf ("alloc failed\n"); return 1; }
  hipMemset (in, 1, n * 8);
  u64 h[BINS];
  for (int skew = 0 ; skew < 2 ; ++skew)
    { for (int b = 0 ; b < BINS ; ++b) h[b] = region * b + (skew ? (u64) (b * 7 + 3) % 16 : 0);      /* bin regions start on a line, or anywhere */
      hipMemcpy (binStart, h, sizeof (h), hipMemcpyHostToDevice);
      for (int grid : { 256, 1024 })
        { const float a = run<true> (in, out, nSub, binStart, cursor, grid), b = run<false> (in, out, nSub, binStart, cursor, grid);
          printf ("regions start %s, grid %4d x 1024: every reservation 32 elements %7.3f ms %5.2f TB/s | counts 24..40 %7.3f ms %5.2f TB/s\n",
                  skew ? "anywhere " : "on a line", grid, a, 2 * n * 8 / a / 1e9, b, 2 * n * 8 / b / 1e9);
        }
    }
  return 0;
}
