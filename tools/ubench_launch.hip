// dev microbenchmark: the floor of a one-launch-per-call facade: host launches a 1-workgroup kernel that writes a sequence
// number into pinned host memory and polls it (as mgIterScan does) -- and the same with hipStreamSynchronize instead.
// build: hipcc -O2 --offload-arch=gfx950 tools/ubench_launch.hip -o tools/ubench_launch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <time.h>
#include <immintrin.h>
struct Big { unsigned long long a[32]; };
__global__ void flagKernel (volatile unsigned long long *flag, unsigned long long seq, Big b)
{ if (threadIdx.x == 0) { __threadfence_system (); __hip_atomic_store ((unsigned long long *) flag, seq + (b.a[0] & 0), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); } }
__global__ void readKernel (const unsigned *in, volatile unsigned long long *flag, unsigned long long seq, Big b)
{ unsigned v = in[threadIdx.x]; if (threadIdx.x == 0) { flag[1] = v; __threadfence_system (); __hip_atomic_store ((unsigned long long *) flag, seq + (b.a[0] & 0), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); } }
static double now () { timespec t; clock_gettime (CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main ()
{
  unsigned long long *h, *d; unsigned *hin, *din;
  hipHostMalloc ((void **) &h, 64, hipHostMallocMapped | hipHostMallocCoherent); hipHostGetDevicePointer ((void **) &d, h, 0);
  hipHostMalloc ((void **) &hin, 4096, hipHostMallocMapped | hipHostMallocCoherent); hipHostGetDevicePointer ((void **) &din, hin, 0);
  hipStream_t st; hipStreamCreateWithFlags (&st, hipStreamNonBlocking);
  Big b; for (int i = 0 ; i < 32 ; ++i) b.a[i] = 0;
  const int N = 20000;
  unsigned long long seq = 0;
  for (int mode = 0 ; mode < 3 ; ++mode)
    { for (int w = 0 ; w < 100 ; ++w) { ++seq; hipLaunchKernelGGL (flagKernel, dim3 (1), dim3 (512), 0, st, d, seq, b); hipStreamSynchronize (st); }
      double t = now ();
      for (int i = 0 ; i < N ; ++i)
        { ++seq;
          if (mode == 2) hipLaunchKernelGGL (readKernel, dim3 (1), dim3 (512), 0, st, din, d, seq, b);
          else hipLaunchKernelGGL (flagKernel, dim3 (1), dim3 (512), 0, st, d, seq, b);
          if (mode == 1) hipStreamSynchronize (st);
          else while (*(volatile unsigned long long *) h != seq) _mm_pause ();
        }
      printf ("%s: %.2f us per launch + completion\n", mode == 0 ? "flag kernel, host polls pinned flag" : mode == 1 ? "flag kernel, hipStreamSynchronize" : "kernel reads pinned host memory first, host polls", (now () - t) / N * 1e6);
    }
  return 0;
}
