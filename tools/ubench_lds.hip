// tools/ubench_lds.hip — LDS atomic throughput with random addresses (round 6): what does a probe of the bucket kernels' claim loop cost the CU?
// One 1024-thread workgroup per CU x 2 (as the bucket kernels run), R slots, every thread ITERS dependent operations on pseudo-random slots.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define R 3200
#define ITERS 4096
template <int OP>
__global__ __launch_bounds__(1024) void k (uint32_t *out, uint32_t seed)
{
  __shared__ unsigned long long s64[R];
  __shared__ uint32_t s32[R];
  for (int i = threadIdx.x; i < R; i += 1024) { s64[i] = 0; s32[i] = 0; }
  __syncthreads ();
  uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + seed, acc = 0;
  for (int it = 0; it < ITERS; ++it)
    { x = x * 1664525u + 1013904223u;
      const uint32_t at = __umulhi (x, R);
      if (OP == 0) acc += (uint32_t) atomicCAS (&s64[at], 0ull, (unsigned long long) x | 1);           // ds_cmpst_rtn_b64, result used (dependent)
      else if (OP == 1) acc += atomicCAS (&s32[at], 0u, x | 1);                                            // ds_cmpst_rtn_b32
      else if (OP == 2) acc += (uint32_t) s64[at];                                                         // ds_read_b64
      else if (OP == 3) acc += s32[at];                                                                    // ds_read_b32
      else if (OP == 4) acc += atomicMax (&s32[at], x);                                                    // ds_max_rtn_u32
      else if (OP == 5) atomicAdd (&s32[at], 1u);                                                          // ds_add_u32 (no return)
      else if (OP == 6) atomicMax (&s32[at], x);                                                           // ds_max_u32 (no return)
      else if (OP == 7) { acc += (uint32_t) atomicCAS (&s64[at], 0ull, (unsigned long long) x | 1); atomicMax (&s32[at], x); atomicAdd (&s32[(at + 1) % R], 1u); }   // a dedup probe: cas64 + max + add
      else if (OP == 8) { acc += atomicCAS (&s32[at], 0u, x | 1); atomicMax (&s32[(at + 7) % R], x); atomicAdd (&s32[(at + 1) % R], 1u); }                          // the same with a 32-bit key
      if (OP <= 4 || OP >= 7) x += acc & 1;      // the next address depends on the result: a chain, as in the claim loop
    }
  out[blockIdx.x * 1024 + threadIdx.x] = acc + (uint32_t) s64[threadIdx.x % R] + s32[threadIdx.x % R];
}
template <int OP> void run (const char *name, uint32_t *d, int ops = 1)
{
  const int blocks = 256 * 2;
  hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
  hipLaunchKernelGGL (k<OP>, dim3 (blocks), dim3 (1024), 0, 0, d, 1u);
  hipDeviceSynchronize ();
  hipEventRecord (e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL (k<OP>, dim3 (blocks), dim3 (1024), 0, 0, d, 2u + r);
  hipEventRecord (e1); hipEventSynchronize (e1);
  float ms; hipEventElapsedTime (&ms, e0, e1); ms /= 3;
  const double lane_ops = (double) blocks * 1024 * ITERS * ops;
  printf ("%-34s %8.3f ms  %7.1f G lane-ops/s  = %.3f per cycle per CU at 2.1 GHz (%d LDS ops a turn)\n", name, ms, lane_ops / ms / 1e6, lane_ops / ms / 1e6 / 256 / 2.1, ops);
}
int main ()
{
  uint32_t *d; hipMalloc (&d, 512 * 1024 * 4);
  run<3> ("ds_read_b32 (warm)", d); run<3> ("ds_read_b32", d); run<2> ("ds_read_b64", d); run<1> ("ds_cmpst_rtn_b32", d); run<0> ("ds_cmpst_rtn_b64", d);
  run<4> ("ds_max_rtn_u32", d); run<6> ("ds_max_u32", d); run<5> ("ds_add_u32", d); run<7> ("cas64 + max + add", d, 3); run<8> ("cas32 + max + add", d, 3);
  return 0;
}
