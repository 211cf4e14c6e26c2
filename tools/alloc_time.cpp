// how long do hipMalloc / hipFree of engine-sized buffers take, fresh and after a free?  (hipcc alloc_time.cpp -o alloc_time)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipFree(0);
  const size_t sizes[] = {2304u << 20, 192u << 20, 96u << 20, 1024u << 20, 48u << 20, 288u << 20, 8u << 20, 24u << 20};
  for (int rep = 0; rep < 4; ++rep) {
    std::vector<void *> p;
    double t0 = now();
    for (size_t s : sizes) { void *q; hipMalloc(&q, s); p.push_back(q); }
    double t1 = now();
    hipMemset(p[0], 0, sizes[0]); hipDeviceSynchronize();
    double t2 = now();
    for (void *q : p) hipFree(q);
    double t3 = now();
    printf("hipMalloc rep %d: malloc %.2f ms, memset 2.3 GB %.2f ms, free %.2f ms\n", rep, t1 - t0, t2 - t1, t3 - t2);
  }
  hipStream_t st; hipStreamCreate(&st);
  hipMemPool_t pool; hipDeviceGetDefaultMemPool(&pool, 0);
  uint64_t thr = ~0ull; hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
  for (int rep = 0; rep < 4; ++rep) {
    std::vector<void *> p;
    double t0 = now();
    for (size_t s : sizes) { void *q = nullptr; hipError_t e = hipMallocAsync(&q, s, st); if (e != hipSuccess) printf("err %s\n", hipGetErrorString(e)); p.push_back(q); }
    hipStreamSynchronize(st);
    double t1 = now();
    hipMemsetAsync(p[0], 0, sizes[0], st); hipStreamSynchronize(st);
    double t2 = now();
    for (void *q : p) hipFreeAsync(q, st);
    hipStreamSynchronize(st);
    double t3 = now();
    printf("hipMallocAsync rep %d: malloc %.2f ms, memset %.2f ms, free %.2f ms\n", rep, t1 - t0, t2 - t1, t3 - t2);
  }
  return 0;
}
