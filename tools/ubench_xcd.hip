// dev microbenchmark: random 16-byte loads from a table when every workgroup only touches the eighth of the
// table that belongs to "its" XCD (blockIdx % 8), with a sequential 4-byte read + 4-byte write per lookup on
// the side (the shape of a rank fetch).  Is an XCD-local slice L2-resident?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ uint64_t mix (uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
// MODE 0: whole table; 1: slice = blockIdx % 8; 2: slice = (blockIdx / 8) % 8 (a wrong guess, for contrast)
template <int MODE, int NT>
__global__ __launch_bounds__ (1024) void k (const unsigned char *t, uint64_t tableBytes, const unsigned *in, unsigned *out, uint64_t n, uint64_t seed)
{
  const uint64_t sliceBytes = tableBytes / 8;
  const unsigned slice = MODE == 1 ? blockIdx.x % 8 : (blockIdx.x / 8) % 8;
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t) gridDim.x * blockDim.x;
  for (; i < n; i += stride * 4)
    { unsigned o[4]; uint4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { uint64_t e = i + j * stride; o[j] = e < n ? (NT ? __builtin_nontemporal_load (in + e) : in[e]) : 0; }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        { uint64_t r = mix (o[j] + seed);
          uint64_t at = MODE == 0 ? (r & (tableBytes - 1)) : slice * sliceBytes + (r % sliceBytes);
          v[j] = *(const uint4 *) (t + (at & ~15ull));
        }
#pragma unroll
      for (int j = 0; j < 4; ++j) { uint64_t e = i + j * stride; if (e < n) { unsigned res = v[j].x + v[j].z; if (NT) __builtin_nontemporal_store (res, out + e); else out[e] = res; } }
    }
}
template <int MODE, int NT> void run (const char *name, unsigned char *t, uint64_t bytes, unsigned *in, unsigned *out, uint64_t n, unsigned grid)
{
  hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
  hipLaunchKernelGGL ((k<MODE, NT>), dim3 (grid), dim3 (1024), 0, 0, t, bytes, in, out, n, 11ull);
  hipDeviceSynchronize ();
  hipEventRecord (e0);
  hipLaunchKernelGGL ((k<MODE, NT>), dim3 (grid), dim3 (1024), 0, 0, t, bytes, in, out, n, 77ull);
  hipEventRecord (e1); hipEventSynchronize (e1);
  float ms; hipEventElapsedTime (&ms, e0, e1);
  printf ("  %-16s %6.3f ms %6.1f G/s", name, ms, n / ms / 1e6);
}
int main ()
{
  uint64_t n = 156000000ull;
  unsigned char *t; hipMalloc (&t, 1ull << 27); hipMemset (t, 1, 1ull << 27);
  unsigned *in, *out; hipMalloc (&in, n * 4); hipMalloc (&out, n * 4);
  hipMemset (in, 0, n * 4);
  { // fill in[] with distinct values
    unsigned *h = (unsigned *) malloc (n * 4); for (uint64_t i = 0; i < n; ++i) h[i] = (unsigned) (i * 2654435761u); hipMemcpy (in, h, n * 4, hipMemcpyHostToDevice); free (h); }
  for (uint64_t mb : { 16, 24, 32, 40, 64 })
    for (unsigned grid : { 2048u, 8192u })
      { uint64_t bytes = mb << 20;
        printf ("%3llu MB grid %5u:", (unsigned long long) mb, grid);
        run<0, 0> ("whole", t, bytes, in, out, n, grid);
        run<1, 0> ("slice b%8", t, bytes, in, out, n, grid);
        run<1, 1> ("slice b%8 nt", t, bytes, in, out, n, grid);
        run<2, 0> ("slice (b/8)%8", t, bytes, in, out, n, grid);
        printf ("\n");
      }
  return 0;
}
