// ldsoob.hip -- what does an LDS read outside the workgroup's allocation return on gfx950?  (round 3: could the extension
// probes of cells that are not part of the wavefront skip their position clamps?)  One wave, 1 KB of LDS filled with a
// pattern; lane i reads a dword at byte address addr[i] (inside, just past the allocation, past 64 KB, past 160 KB, "negative").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void probe(const unsigned* addr, unsigned* out, int n) {
  extern __shared__ unsigned lds[];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 0xA0000000u + i;
  __syncthreads();
  if ((int)threadIdx.x < n) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr[threadIdx.x]) : "memory");
    out[threadIdx.x] = v;
  }
}
int main() {
  std::vector<unsigned> a = {0u, 4u, 1020u, 1024u, 2048u, 65532u, 65536u, 100000u, 163836u, 163840u, 1u << 20, 0x7FFFFFFCu, 0xFFFFFFFCu, 0xFFFFF000u};
  unsigned *da, *dout;
  hipMalloc(&da, a.size() * 4); hipMalloc(&dout, a.size() * 4);
  hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
  hipMemset(dout, 0xEE, a.size() * 4);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 1024, 0, da, dout, (int)a.size());
  hipError_t e = hipDeviceSynchronize();
  printf("sync: %s\n", hipGetErrorString(e));
  std::vector<unsigned> o(a.size());
  hipMemcpy(o.data(), dout, a.size() * 4, hipMemcpyDeviceToHost);
  for (size_t i = 0; i < a.size(); ++i) printf("addr 0x%08x -> 0x%08x\n", a[i], o[i]);
  return 0;
}
