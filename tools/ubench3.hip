// tools/ubench3.hip — issue rates of the instructions of the exact-mode scan's phase A that tools/ubench2.hip did not price (round 6, VERDICT r5
// item 7): selects (v_cndmask with its mask in VCC / in an SGPR pair), 64-bit shifts / compares / adds, v_bfrev, v_bitop3 -- and a select written as
// arithmetic.  Same harness: 8 independent chains per lane, 8 waves per SIMD, every CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITERS 32768
template <int OP>
__global__ __launch_bounds__(256) void k (uint32_t *out, uint32_t s0, uint32_t s1)
{
  uint32_t a[8]; uint64_t q[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 2654435761u + i * 40503u + s0; q[i] = ((uint64_t) a[i] << 32) | (a[i] * 31u); }
  uint32_t f = s1 | 1;
  for (int it = 0; it < ITERS; ++it)
    {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        {
          if (OP == 0) asm volatile ("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[i]), "v"(f) : "vcc");
          else if (OP == 1) asm volatile ("v_cndmask_b32 %0, %1, %2, s[20:21]" : "=v"(a[i]) : "v"(a[i]), "v"(f) : "s20", "s21");
          else if (OP == 2) asm volatile ("v_cmp_gt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[i]), "v"(f) : "vcc");          /* 2 inst */
          else if (OP == 3) asm volatile ("v_lshlrev_b64 %0, 3, %1" : "=v"(q[i]) : "v"(q[i]));
          else if (OP == 4) asm volatile ("v_lshrrev_b64 %0, 3, %1" : "=v"(q[i]) : "v"(q[i]));
          else if (OP == 5) asm volatile ("v_cmp_lt_u64 vcc, %0, %1" : : "v"(q[i]), "v"(q[(i + 1) & 7]) : "vcc");
          else if (OP == 6) asm volatile ("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(q[i]) : "v"(q[i]), "v"(q[(i + 1) & 7]));
          else if (OP == 7) asm volatile ("v_bfrev_b32 %0, %1" : "=v"(a[i]) : "v"(a[i]));
          else if (OP == 8) asm volatile ("v_bitop3_b32 %0, %1, %2, %2 bitop3:0x96" : "=v"(a[i]) : "v"(a[i]), "v"(f));
          else if (OP == 9) asm volatile ("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(a[i]), "v"(f) : "vcc");
          else if (OP == 10) asm volatile ("v_cmp_lt_u64 vcc, %2, %3\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(f), "v"(q[i]), "v"(q[(i + 1) & 7]) : "vcc");   /* 64-bit compare + one select */
          else if (OP == 11) asm volatile ("v_sub_u32 %0, 0, %1\n\tv_and_b32 %0, %0, %2\n\tv_xor_b32 %0, %0, %1" : "=&v"(a[i]) : "v"(a[i]), "v"(f));      /* a select as arithmetic: 3 inst */
          else if (OP == 12) asm volatile ("v_cmp_eq_u32 vcc, %0, %1" : : "v"(a[i]), "v"(f) : "vcc");
          else if (OP == 13) asm volatile ("v_and_or_b32 %0, %1, %2, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f));
          else if (OP == 14) asm volatile ("v_mul_lo_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(f));
          else if (OP == 15) asm volatile ("v_ashrrev_i32 %0, 31, %1" : "=v"(a[i]) : "v"(a[i]));
          else if (OP == 16) asm volatile ("v_subb_co_u32 %0, vcc, %1, %1, vcc" : "=v"(a[i]) : "v"(a[i]) : "vcc");      /* mask = -carry */
        }
    }
  uint32_t r = 0; for (int i = 0; i < 8; ++i) r ^= a[i] ^ (uint32_t) q[i] ^ (uint32_t) (q[i] >> 32);
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
  printf ("%-34s %8.3f ms  %7.2f T lane-ops/s  (%d inst)\n", name, ms, ops / ms / 1e9, perInst);
}
int main ()
{
  uint32_t *d; hipMalloc (&d, 256 * 8 * 256 * 4);
  run<14> ("v_mul_lo_u32 (warm)", d); run<14> ("v_mul_lo_u32", d);
  run<0> ("v_cndmask_b32 (vcc)", d); run<1> ("v_cndmask_b32 (sgpr pair)", d); run<2> ("v_cmp_gt_u32 + v_cndmask", d, 2);
  run<3> ("v_lshlrev_b64", d); run<4> ("v_lshrrev_b64", d); run<5> ("v_cmp_lt_u64", d); run<6> ("v_lshl_add_u64", d);
  run<7> ("v_bfrev_b32", d); run<8> ("v_bitop3_b32", d); run<9> ("v_mad_u64_u32", d); run<10> ("v_cmp_lt_u64 + v_cndmask", d, 2);
  run<11> ("select as sub/and/xor", d, 3); run<12> ("v_cmp_eq_u32", d); run<13> ("v_and_or_b32", d); run<15> ("v_ashrrev_i32", d); run<16> ("v_subb_co_u32", d);
  return 0;
}
