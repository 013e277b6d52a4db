// tools/ubench2.hip — issue rates of more VALU instructions on gfx950 (dev helper): which ones run above the
// 37.6 T lane-ops/s that v_mul_lo_u32 / v_alignbit_b32 / v_min_u32 reach?
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
  float fa[8]; for (int i = 0; i < 8; ++i) fa[i] = (float) a[i];
  for (int it = 0; it < ITERS; ++it)
    {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        {
          if (OP == 0) asm volatile ("v_add_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f));
          else if (OP == 1) asm volatile ("v_and_b32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f));
          else if (OP == 2) asm volatile ("v_xor_b32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f));
          else if (OP == 3) asm volatile ("v_lshlrev_b32 %0, 3, %1" : "=v"(a[i]) : "v"(a[i]));
          else if (OP == 4) asm volatile ("v_bfe_u32 %0, %1, 3, 20" : "=v"(a[i]) : "v"(a[i]));
          else if (OP == 5) asm volatile ("v_cmp_gt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(a[i]) : "v"(f), "v"(a[i]) : "vcc");
          else if (OP == 6) asm volatile ("v_add3_u32 %0, %1, %2, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f));
          else if (OP == 7) asm volatile ("v_fma_f32 %0, %1, %2, %1" : "=v"(fa[i]) : "v"(fa[i]), "v"(1.0001f));
          else if (OP == 8) asm volatile ("v_sub_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f));
          else if (OP == 9) asm volatile ("v_max_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f));
          else if (OP == 10) asm volatile ("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[i]), "v"(f) : "vcc");
          else if (OP == 11) asm volatile ("v_mul_lo_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f));
          else if (OP == 12) asm volatile ("v_alignbit_b32 %0, %1, %2, 6" : "=v"(a[i]) : "v"(a[i]), "v"(f));
          else if (OP == 13) asm volatile ("v_or_b32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f));
          else if (OP == 14) asm volatile ("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
          else if (OP == 15) asm volatile ("v_perm_b32 %0, %1, %2, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f));
          else if (OP == 16) asm volatile ("v_add_co_u32 %0, s[20:21], %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f) : "s20", "s21");      /* carry out to an SGPR pair */
          else if (OP == 17) asm volatile ("v_add_co_u32 %0, vcc, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f) : "vcc");
          else if (OP == 18) asm volatile ("v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(a[i]) : : "vcc");
          else if (OP == 19) asm volatile ("v_cmp_gt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(f) : "vcc");
          else if (OP == 20) asm volatile ("v_cmp_gt_u32 s[20:21], %0, %1" : : "v"(a[i]), "v"(f) : "s20", "s21");
          else if (OP == 21) asm volatile ("v_add_co_u32 %0, s[20:21], %1, %2\n\tv_add_co_u32 %0, s[22:23], %0, %2\n\ts_and_b64 vcc, s[20:21], s[22:23]\n\tv_addc_co_u32 %3, vcc, %3, %3, vcc"
                                           : "=&v"(a[i]), "+v"(a[i]) , "+v"(f), "+v"(a[(i + 1) & 7]) : : "s20", "s21", "s22", "s23", "vcc");   /* the proposed threshold test: 3 VALU + 1 SALU */
          else if (OP == 22) asm volatile ("v_min_u32 %0, %1, %2\n\tv_cmp_gt_u32 vcc, %2, %0\n\tv_addc_co_u32 %3, vcc, %3, %3, vcc" : "=&v"(a[i]) : "v"(a[i]), "v"(f), "v"(a[(i + 1) & 7]) : "vcc");   /* today's: min, cmp, addc */
          else if (OP == 23) asm volatile ("v_mul_u32_u24 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f));
          else if (OP == 24) asm volatile ("v_mad_u32_u24 %0, %1, %2, %1" : "=v"(a[i]) : "v"(a[i]), "v"(f));
          else if (OP == 25) asm volatile ("v_min3_u32 %0, %1, %2, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f));
          else if (OP == 26) asm volatile ("v_sub_co_u32 %0, vcc, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f) : "vcc");
          else if (OP == 27) asm volatile ("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f));
          else if (OP == 28) asm volatile ("v_xad_u32 %0, %1, %2, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f));
          else if (OP == 29) asm volatile ("v_mul_hi_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f));
        }
    }
  uint32_t r = 0; for (int i = 0; i < 8; ++i) r ^= a[i] ^ (uint32_t) fa[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP> void run (const char *name, uint32_t *d, int perInst = 1)
{
  const int blocks = 256 * 8;
  hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
  hipLaunchKernelGGL (k<OP>, dim3 (blocks), dim3 (256), 0, 0, d, 1u, 0x9e3779b9u);
  hipDeviceSynchronize ();
  hipEventRecord (e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL (k<OP>, dim3 (blocks), dim3 (256), 0, 0, d, 1u, 0x9e3779b9u);
  hipEventRecord (e1); hipEventSynchronize (e1);
  float ms; hipEventElapsedTime (&ms, e0, e1); ms /= 3;
  double ops = (double) blocks * 256 * ITERS * 8 * perInst;
  printf ("%-22s %8.3f ms  %7.2f T lane-ops/s\n", name, ms, ops / ms / 1e9);
}
int main ()
{
  uint32_t *d; hipMalloc (&d, 256 * 8 * 256 * 4);
  run<11> ("v_mul_lo_u32 (warm)", d); run<0> ("v_add_u32", d); run<8> ("v_sub_u32", d); run<1> ("v_and_b32", d); run<13> ("v_or_b32", d); run<2> ("v_xor_b32", d);
  run<3> ("v_lshlrev_b32", d); run<4> ("v_bfe_u32", d); run<5> ("v_cmp+v_addc (2 inst)", d, 2); run<6> ("v_add3_u32", d); run<9> ("v_max_u32", d);
  run<16> ("v_add_co_u32 -> sgpr", d); run<17> ("v_add_co_u32 -> vcc", d); run<26> ("v_sub_co_u32 -> vcc", d); run<18> ("v_addc_co_u32", d); run<19> ("v_cmp_gt_u32 -> vcc", d); run<20> ("v_cmp_gt_u32 -> sgpr", d);
  run<21> ("2 add_co + s_and + addc", d, 3); run<22> ("min + cmp + addc", d, 3); run<23> ("v_mul_u32_u24", d); run<24> ("v_mad_u32_u24", d); run<25> ("v_min3_u32", d); run<27> ("v_lshl_add_u32", d); run<28> ("v_xad_u32", d); run<29> ("v_mul_hi_u32", d);
  run<10> ("v_cndmask_b32", d); run<14> ("v_mov_b32", d); run<15> ("v_perm_b32", d); run<12> ("v_alignbit_b32", d); run<11> ("v_mul_lo_u32", d); run<7> ("v_fma_f32", d); run<0> ("v_add_u32 (again)", d);
  return 0;
}
