// dev: does a small device-to-host copy wait behind a large host-to-device copy that is in flight on another (non-blocking) stream?
// hipcc --offload-arch=gfx950 -O2 -o /tmp/ubench_copyq tools/ubench_copyq.hip && /tmp/ubench_copyq
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <chrono>
static double now () { return std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now ().time_since_epoch ()).count (); }
__global__ void copyOut (const uint64_t *src, uint64_t *dstHost, int n) { for (int i = threadIdx.x ; i < n ; i += blockDim.x) dstHost[i] = src[i]; }
int main ()
{ const size_t big = 128u << 20;
  unsigned char *hBig, *dBig; uint64_t *hSmall, *dSmall, *hMapped, *dMapped;
  hipHostMalloc ((void **) &hBig, big, hipHostMallocDefault); hipMalloc ((void **) &dBig, big);
  hipHostMalloc ((void **) &hSmall, 4096, hipHostMallocDefault); hipMalloc ((void **) &dSmall, 4096); hipMemset (dSmall, 1, 4096);
  hipHostMalloc ((void **) &hMapped, 4096, hipHostMallocMapped); hipHostGetDevicePointer ((void **) &dMapped, hMapped, 0);
  hipStream_t copy, other; hipStreamCreateWithFlags (&copy, hipStreamNonBlocking); hipStreamCreateWithFlags (&other, hipStreamNonBlocking);
  uint64_t *pageable = (uint64_t *) malloc (4096);
  for (int variant = 0 ; variant < 5 ; ++variant)
    for (int rep = 0 ; rep < 3 ; ++rep)
      { hipDeviceSynchronize ();
        double t0 = now ();
        hipMemcpyAsync (dBig, hBig, big, hipMemcpyHostToDevice, copy);
        double t1 = now ();
        const char *what = "";
        switch (variant)
          { case 0: what = "hipMemcpy D2H 8 B to pinned (null stream)"; hipMemcpy (hSmall, dSmall, 8, hipMemcpyDeviceToHost); break;
            case 1: what = "hipMemcpy D2H 8 B to pageable (null stream)"; hipMemcpy (pageable, dSmall, 8, hipMemcpyDeviceToHost); break;
            case 2: what = "hipMemcpyAsync D2H 8 B on another non-blocking stream + sync"; hipMemcpyAsync (hSmall, dSmall, 8, hipMemcpyDeviceToHost, other); hipStreamSynchronize (other); break;
            case 3: what = "kernel writes 8 B to mapped host memory (null stream) + sync"; hipLaunchKernelGGL (copyOut, dim3 (1), dim3 (64), 0, 0, dSmall, dMapped, 1); hipStreamSynchronize (0); break;
            case 4: what = "hipMemcpy H2D 8 B from pinned (null stream)"; hipMemcpy (dSmall, hSmall, 8, hipMemcpyHostToDevice); break;
          }
        double t2 = now ();
        hipStreamSynchronize (copy);
        double t3 = now ();
        printf ("%-66s: issue big %.3f ms, small op %.3f ms, big done after %.3f ms\n", what, t1 - t0, t2 - t1, t3 - t0);
      }
  return 0; }
