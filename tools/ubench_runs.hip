// dev microbenchmark: what a partition pass can reach.  A stream of 8-byte elements is read in order and written out as RUNS of
// R elements to pseudo-random places (a bijection of the run index), the runs starting at any 8-byte offset (as a radix
// partition's runs do: the bins' cursors advance by counts), against the plain in-order copy of the same bytes.
// Rates count bytes read + bytes written.   hipcc -O3 --offload-arch=gfx950 tools/ubench_runs.hip -o ubench_runs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned long long u64;

template <int UNR, bool NT>
__global__ __launch_bounds__ (256) void copyRuns (const u64 *__restrict__ in, u64 *__restrict__ out, u64 n, int runShift, u64 runMask, u64 mult, u64 skew)
{
  // thread t of the grid takes elements t, t + T, ... (T = grid size): a wave covers 64 consecutive elements
  const u64 T = (u64) gridDim.x * blockDim.x;
  for (u64 i0 = (u64) blockIdx.x * blockDim.x + threadIdx.x ; i0 < n ; i0 += T * UNR)
    { u64 v[UNR];
#pragma unroll
      for (int j = 0 ; j < UNR ; ++j) { const u64 i = i0 + (u64) j * T; v[j] = i < n ? in[i] : 0; }
#pragma unroll
      for (int j = 0 ; j < UNR ; ++j)
        { const u64 i = i0 + (u64) j * T;
          if (i >= n) continue;
          const u64 run = i >> runShift, within = i & (((u64) 1 << runShift) - 1);
          const u64 to = (((run * mult) & runMask) << runShift) + within + skew;          /* mult odd: a bijection of the runs */
          if (NT) __builtin_nontemporal_store (v[j], out + to); else out[to] = v[j];
        }
    }
}

template <int UNR, bool NT> static float timeIt (const u64 *in, u64 *out, u64 n, int runShift, u64 mult, u64 skew, int grid)
{
  const u64 runMask = (n >> runShift) - 1;
  hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
  hipLaunchKernelGGL ((copyRuns<UNR, NT>), dim3 (grid), dim3 (256), 0, 0, in, out, n, runShift, runMask, mult, skew);
  hipDeviceSynchronize ();
  float best = 1e9f;
  for (int rep = 0 ; rep < 3 ; ++rep)
    { hipEventRecord (e0);
      hipLaunchKernelGGL ((copyRuns<UNR, NT>), dim3 (grid), dim3 (256), 0, 0, in, out, n, runShift, runMask, mult, skew);
      hipEventRecord (e1); hipEventSynchronize (e1);
      float ms; hipEventElapsedTime (&ms, e0, e1); if (ms < best) best = ms;
    }
  return best;
}

int main ()
{
  const u64 n = (u64) 1 << 28;                    /* 2 GiB in, 2 GiB out */
  u64 *in, *out;
  if (hipMalloc (&in, n * 8) != hipSuccess || hipMalloc (&out, (n + 64) * 8) != hipSuccess) { printf ("alloc failed\n"); return 1; }
  hipMemset (in, 1, n * 8); hipMemset (out, 0, (n + 64) * 8);
  printf ("copy of %.1f GiB of 8-byte elements (bytes read + written per second), MI355X\n", n * 8.0 / (1 << 30));
  for (int grid : { 2048, 8192 })
    { float ms = timeIt<8, false> (in, out, n, 20, 1, 0, grid);
      printf ("  in order, grid %5d x 256, 8 loads in flight:            %7.3f ms  %6.2f TB/s\n", grid, ms, 2 * n * 8 / ms / 1e9);
      ms = timeIt<8, true> (in, out, n, 20, 1, 0, grid);
      printf ("  in order, grid %5d x 256, non-temporal stores:          %7.3f ms  %6.2f TB/s\n", grid, ms, 2 * n * 8 / ms / 1e9);
    }
  { hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
    hipMemcpyAsync (out, in, n * 8, hipMemcpyDeviceToDevice, 0); hipDeviceSynchronize ();
    hipEventRecord (e0); hipMemcpyAsync (out, in, n * 8, hipMemcpyDeviceToDevice, 0); hipEventRecord (e1); hipEventSynchronize (e1);
    float ms; hipEventElapsedTime (&ms, e0, e1);
    printf ("  hipMemcpyAsync device to device:                            %7.3f ms  %6.2f TB/s\n", ms, 2 * n * 8 / ms / 1e9);
    hipEventRecord (e0); hipMemsetAsync (out, 0, n * 8, 0); hipEventRecord (e1); hipEventSynchronize (e1);
    hipEventElapsedTime (&ms, e0, e1);
    printf ("  hipMemsetAsync (write only):                                %7.3f ms  %6.2f TB/s written\n", ms, n * 8 / ms / 1e9);
  }
  const u64 mult = 0x9E3779B97F4A7C15ull | 1;
  for (int runShift : { 3, 4, 5, 6, 7, 9 })
    for (u64 skew : { (u64) 0, (u64) 5 })
      { float ms = timeIt<8, false> (in, out, n, runShift, mult, skew, 8192);
        float msn = timeIt<8, true> (in, out, n, runShift, mult, skew, 8192);
        printf ("  runs of %4d elements (%5d B) to random places, start %s:  %7.3f ms  %6.2f TB/s   (non-temporal stores %6.2f)\n",
                1 << runShift, 8 << runShift, skew ? "any 8 B  " : "run-sized", ms, 2 * n * 8 / ms / 1e9, 2 * n * 8 / msn / 1e9);
      }
  return 0;
}
