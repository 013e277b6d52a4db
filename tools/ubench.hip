// tools/ubench.hip — integer instruction throughput on gfx950 (dev helper, not part of the library)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITERS 32768
template <int OP>
__global__ __launch_bounds__(256) void k (uint32_t *out, uint32_t s0, uint32_t s1)
{
  uint32_t a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u + s0;
  uint32_t f = s1 | 1;
  for (int it = 0; it < ITERS; ++it)
    {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        {
          if (OP == 0) asm volatile ("v_mul_lo_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f));
          else if (OP == 1) asm volatile ("v_mul_u32_u24 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f));
          else if (OP == 2) asm volatile ("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(f), "v"(a[i]));
          else if (OP == 3) asm volatile ("v_dot2_u32_u16 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(f), "v"(a[i]));
          else if (OP == 4) asm volatile ("v_alignbit_b32 %0, %1, %2, 6" : "=v"(a[i]) : "v"(a[i]), "v"(f));
          else if (OP == 5) asm volatile ("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f));
          else if (OP == 6) { uint64_t r; asm volatile ("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(a[i]), "v"(f), "v"((uint64_t) a[i]) : "vcc"); a[i] = (uint32_t) r ^ (uint32_t) (r >> 32); }
          else if (OP == 7) asm volatile ("v_mul_hi_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f));
          else if (OP == 8) asm volatile ("v_add_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f));
          else if (OP == 9) asm volatile ("v_min_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f));
          else if (OP == 10) asm volatile ("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f));
          else if (OP == 11) asm volatile ("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(f), "v"(a[i]));
          else if (OP == 12) asm volatile ("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(f), "v"(a[i]));
          else if (OP == 13) asm volatile ("v_dot4_u32_u8 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(f), "v"(a[i]));
        }
    }
  uint32_t r = 0; for (int i = 0; i < 8; ++i) r ^= a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP> void run (const char *name, uint32_t *d)
{
  const int blocks = 256 * 8;
  hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
  hipLaunchKernelGGL (k<OP>, dim3 (blocks), dim3 (256), 0, 0, d, 1u, 0x9e3779b9u);
  hipDeviceSynchronize ();
  hipEventRecord (e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL (k<OP>, dim3 (blocks), dim3 (256), 0, 0, d, 1u, 0x9e3779b9u);
  hipEventRecord (e1); hipEventSynchronize (e1);
  float ms; hipEventElapsedTime (&ms, e0, e1); ms /= 3;
  double ops = (double) blocks * 256 * ITERS * 8;
  printf ("%-18s %8.3f ms  %7.2f T lane-ops/s\n", name, ms, ops / ms / 1e9);
}
int main ()
{
  uint32_t *d; hipMalloc (&d, 256 * 8 * 256 * 4);
  run<8> ("v_add_u32", d); run<0> ("v_mul_lo_u32", d); run<7> ("v_mul_hi_u32", d); run<1> ("v_mul_u32_u24", d);
  run<12> ("v_mad_u32_u24", d); run<2> ("v_mad_u32_u16", d); run<3> ("v_dot2_u32_u16", d); run<13> ("v_dot4_u32_u8", d);
  run<4> ("v_alignbit_b32", d); run<5> ("v_lshl_add_u32", d); run<6> ("v_mad_u64_u32", d); run<9> ("v_min_u32", d);
  run<10> ("v_pk_mul_lo_u16", d); run<11> ("v_pk_mad_u16", d);
  return 0;
}
