#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const size_t gb = argc > 1 ? atoi(argv[1]) : 38;
  for (int rep = 0; rep < 6; ++rep) {
    void* p = nullptr;
    double t0 = now();
    hipError_t e = hipMalloc(&p, gb << 30);
    double t1 = now();
    hipMemsetAsync(p, 0, 1 << 20, 0); hipDeviceSynchronize();
    double t2 = now();
    hipFree(p);
    double t3 = now();
    printf("rep %d: malloc %zu GiB %.3f s (%s), touch %.3f, free %.3f\n", rep, gb, t1 - t0, hipGetErrorString(e), t2 - t1, t3 - t2);
    if (rep == 2) {  // burn some CPU on threads, as the PAF formatter does
      std::vector<std::thread> th;
      for (int t = 0; t < 4; ++t) th.emplace_back([] { volatile double x = 0; for (long i = 0; i < 200000000; ++i) x += i; });
      for (auto& x : th) x.join();
    }
  }
  return 0;
}
