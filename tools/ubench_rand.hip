// dev microbenchmark: random load/store throughput versus footprint (L2-resident .. MALL .. HBM)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ uint64_t mix (uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
template <int OP, int UNR>
__global__ void k (unsigned char *t, uint64_t mask, uint64_t n, uint64_t seed, unsigned *sink)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t) gridDim.x * blockDim.x;
  unsigned acc = 0;
  for (; i < n; i += stride * UNR)
    { uint64_t s[UNR];
#pragma unroll
      for (int j = 0; j < UNR; ++j) s[j] = mix (i + j * stride + seed) & mask;
#pragma unroll
      for (int j = 0; j < UNR; ++j)
        { if (OP == 0) acc += *(const unsigned *) (t + (s[j] & ~3ull));
          else if (OP == 1) { uint4 v = *(const uint4 *) (t + (s[j] & ~15ull)); acc += v.x ^ v.w; }
          else if (OP == 2) t[s[j]] = 1;
          else if (OP == 3) *(unsigned *) (t + (s[j] & ~3ull)) = (unsigned) i;
          else if (OP == 4) { typedef unsigned v4u __attribute__ ((ext_vector_type (4))); v4u v = __builtin_nontemporal_load ((const v4u *) (t + (s[j] & ~15ull))); acc += v.x ^ v.w; }
        }
    }
  if (acc == 0x12345678u) *sink = acc;
}
template <int OP, int UNR> void run (const char *name, unsigned char *t, uint64_t bytes, uint64_t n, unsigned *sink)
{
  hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
  hipLaunchKernelGGL ((k<OP, UNR>), dim3 (8192), dim3 (256), 0, 0, t, bytes - 1, n, 11ull, sink);
  hipDeviceSynchronize ();
  hipEventRecord (e0);
  hipLaunchKernelGGL ((k<OP, UNR>), dim3 (8192), dim3 (256), 0, 0, t, bytes - 1, n, 77ull, sink);
  hipEventRecord (e1); hipEventSynchronize (e1);
  float ms; hipEventElapsedTime (&ms, e0, e1);
  printf ("  %-22s %7.3f ms %7.1f G/s", name, ms, n / ms / 1e6);
}
int main ()
{
  unsigned *sink; hipMalloc (&sink, 4);
  uint64_t n = 1ull << 27;
  unsigned char *t; hipMalloc (&t, 1ull << 31); hipMemset (t, 0, 1ull << 31);
  for (int lg = 19; lg <= 31; ++lg)
    { uint64_t bytes = 1ull << lg;
      printf ("%6llu KB:", (unsigned long long) (bytes >> 10));
      run<0, 1> ("ld4", t, bytes, n, sink); run<1, 1> ("ld16", t, bytes, n, sink); run<1, 4> ("ld16x4", t, bytes, n, sink);
      run<4, 4> ("ld16nt x4", t, bytes, n, sink);
      run<2, 1> ("st1", t, bytes, n, sink); run<3, 1> ("st4", t, bytes, n, sink);
      printf ("\n");
    }
  return 0;
}
