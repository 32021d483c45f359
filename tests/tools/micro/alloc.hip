// Ad-hoc: what a context's device memory costs to get (round 5: the file splitter/muxer waits 0.3-0.7 s for its first context)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
  double t0 = now();
  hipFree(nullptr);
  printf("runtime initialisation %.3f s\n", now() - t0);
  const size_t sizes[] = { (size_t)1 << 30, (size_t)4 << 30, (size_t)14 << 30, (size_t)14 << 30, (size_t)60 << 30 };
  for (size_t sz : sizes) {
    void *p = nullptr;
    t0 = now();
    hipError_t e = hipMalloc(&p, sz);
    double t1 = now();
    hipFree(p);
    printf("hipMalloc %5.1f GB: %.3f s (%s), hipFree %.3f s\n", sz / 1073741824.0, t1 - t0, hipGetErrorString(e), now() - t1);
  }
  {
    void *p[8];
    t0 = now();
    for (int i = 0; i < 8; i++) hipMalloc(&p[i], (size_t)14 << 27);
    printf("8 x hipMalloc 1.75 GB: %.3f s\n", now() - t0);
    for (int i = 0; i < 8; i++) hipFree(p[i]);
  }
  hipStream_t q;
  hipStreamCreate(&q);
  for (int rep = 0; rep < 2; rep++) {
    void *p = nullptr;
    t0 = now();
    hipError_t e = hipMallocAsync(&p, (size_t)14 << 30, q);
    hipStreamSynchronize(q);
    double t1 = now();
    hipFreeAsync(p, q);
    hipStreamSynchronize(q);
    printf("hipMallocAsync 14 GB: %.3f s (%s), free %.3f s\n", t1 - t0, hipGetErrorString(e), now() - t1);
  }
  for (size_t sz : { (size_t)256 << 20, (size_t)1 << 30 }) {
    void *h = nullptr;
    t0 = now();
    hipHostMalloc(&h, sz, hipHostMallocPortable);
    double t1 = now();
    hipHostFree(h);
    printf("hipHostMalloc %.2f GB: %.3f s, free %.3f s\n", sz / 1073741824.0, t1 - t0, now() - t1);
  }
  return 0;
}
