// tools/ubench_mem.hip — random access / atomic rates vs footprint on MI355X (dev helper)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ uint64_t mix (uint64_t x)
{ x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
struct Slot { unsigned long long key; unsigned ord, cnt; };
template <int OP>
__global__ void k (Slot *t, uint64_t mask, uint64_t n, uint64_t seed, unsigned *sink)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t) gridDim.x * blockDim.x;
  unsigned acc = 0;
  for (; i < n; i += stride)
    { uint64_t s = mix (i + seed) & mask;
      if (OP == 0) { acc += t[s].ord; }                                   // 4-byte random load
      else if (OP == 1) { uint4 v = *(const uint4 *) &t[s]; acc += v.x ^ v.w; }     // 16-byte random load
      else if (OP == 2) atomicAdd (&t[s].cnt, 1u);                        // no-return add
      else if (OP == 3) acc += atomicMax (&t[s].ord, (unsigned) i);       // returning max
      else if (OP == 4) acc += (unsigned) atomicCAS (&t[s].key, 0ull, (unsigned long long) (i + 1));
      else if (OP == 5) { acc += (unsigned) atomicCAS (&t[s].key, 0ull, (unsigned long long) (i + 1)); acc += atomicMax (&t[s].ord, (unsigned) i); atomicAdd (&t[s].cnt, 1u); }
      else if (OP == 6) { t[s].cnt = (unsigned) i; }                      // 4-byte random store
      else if (OP == 7) { unsigned long long kk = t[s].key; if (kk == 0) kk = atomicCAS (&t[s].key, 0ull, (unsigned long long) (i + 1)); unsigned v = t[s].ord; if (v < (unsigned) i) atomicMax (&t[s].ord, (unsigned) i); atomicAdd (&t[s].cnt, 1u); acc += (unsigned) kk; } // current insert pattern
      else if (OP == 8) { acc += __hip_atomic_fetch_add (&t[s].cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } // L2-scope add
    }
  if (acc == 0x12345678u) *sink = acc;
}
template <int OP> void run (const char *name, Slot *t, uint64_t slots, uint64_t n, unsigned *sink)
{
  hipMemset (t, 0, slots * sizeof (Slot));
  hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
  hipDeviceSynchronize ();
  hipEventRecord (e0);
  hipLaunchKernelGGL (k<OP>, dim3 (16384), dim3 (256), 0, 0, t, slots - 1, n, 77ull, sink);
  hipEventRecord (e1); hipEventSynchronize (e1);
  float ms; hipEventElapsedTime (&ms, e0, e1);
  printf ("  %-28s %8.3f ms  %6.2f G ops/s\n", name, ms, n / ms / 1e6);
}
int main ()
{
  unsigned *sink; hipMalloc (&sink, 4);
  uint64_t n = 150000000ull;
  for (int lg = 22; lg <= 29; lg += (lg < 26 ? 2 : 1))
    { uint64_t slots = 1ull << lg;
      Slot *t; if (hipMalloc (&t, slots * sizeof (Slot)) != hipSuccess) { printf ("alloc fail\n"); return 1; }
      printf ("table %llu MB (n = %llu accesses)\n", (unsigned long long) (slots * 16 >> 20), (unsigned long long) n);
      run<0> ("load 4B", t, slots, n, sink); run<1> ("load 16B", t, slots, n, sink); run<6> ("store 4B", t, slots, n, sink);
      run<2> ("atomicAdd (no return)", t, slots, n, sink); run<8> ("atomicAdd wg-scope returning", t, slots, n, sink);
      run<3> ("atomicMax returning", t, slots, n, sink);
      run<4> ("atomicCAS 64", t, slots, n, sink); run<5> ("CAS+Max+Add same slot", t, slots, n, sink); run<7> ("load,CAS,load,Max,Add", t, slots, n, sink);
      hipFree (t);
    }
  return 0;
}
