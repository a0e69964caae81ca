// Memory-pattern ceiling: the row traffic of the step kernel (7 x 512-B loads + 5 x 512-B stores per
// window, rows of a 32-deep ring per workgroup arena) with no algorithmic work at all.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__global__ __launch_bounds__(64, 4) void k(unsigned char* arena, size_t slot_stride, int wcap, int steps, int width, int waitmode, unsigned long long* sink) {
  constexpr int AM_LD = AUX_MLOAD, AM_ST = AUX_MSTORE, AI_ST = AUX_IDSTORE, AI_LD = AUX_IDLOAD;
  const int lane = threadIdx.x;
  unsigned char* base = arena + (size_t)blockIdx.x * slot_stride;
  rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)slot_stride, 0x00020000);
  const int rowb = wcap * 2;
  auto off = [&](int dir, int comp, int score) { const int c2 = comp == 0 ? 0 : (comp == 1 ? 1 : 3); return ((dir * 5 + c2) * 32 + (score & 31) * (comp == 0 ? 1 : 2)) * rowb; };
  unsigned acc = 0;
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  for (int s = 32; s < 32 + steps; ++s) {
    for (int dir = 0; dir < 2; ++dir) {
      const int lo = wcap / 2 - width / 2 + (s & 7) * 4;  // drifting start column
      const int sMx = off(dir, 0, s - 5), sO1 = off(dir, 0, s - 10), sO2 = off(dir, 0, s - 25);
      const int sI1 = off(dir, 1, s - 2), sD1 = off(dir, 3, s - 2), sI2 = off(dir, 2, s - 1), sD2 = off(dir, 4, s - 1);
      const int tM = off(dir, 0, s), tI1 = off(dir, 1, s), tD1 = off(dir, 3, s), tI2 = off(dir, 2, s), tD2 = off(dir, 4, s);
      for (int cb = lo; cb < lo + width; cb += 248) {
        const int voff = (cb + lane * 4) * 2, voff4 = (cb + lane * 4) * 4;
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, sMx, 0);
        u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, sO1, 0);
        u32x2 c = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, sO2, 0);
        u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rs, voff4, sI1, 2);   // {I1,D1} interleaved row (rows of 2x width: I1 and D1 slots adjacent)
        u32x4 f = __builtin_amdgcn_raw_buffer_load_b128(rs, voff4, sI2, 2);   // {I2,D2}
        u32x2 m = a + b + c;
        u32x4 id1 = d + f, id2 = f; id2[0] += b[0]; id1[1] += c[1];
        acc += m[0] ^ m[1];
        __builtin_amdgcn_raw_buffer_store_b128(id1, rs, voff4, tI1, 0);
        __builtin_amdgcn_raw_buffer_store_b128(id2, rs, voff4, tI2, 0);
        __builtin_amdgcn_raw_buffer_store_b64(m, rs, voff, tM, 0);
      }
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
int main(int argc, char** argv) {
  const int nslots = argc > 1 ? atoi(argv[1]) : 4096;
  const int width = argc > 2 ? atoi(argv[2]) : 1400;
  const int steps = argc > 3 ? atoi(argv[3]) : 400;
  const int waitmode = argc > 4 ? atoi(argv[4]) : 0;
  const int wcap = 20480;
  const size_t slot_stride = (size_t)2 * 5 * 32 * wcap * 2;
  unsigned char* arena; unsigned long long* sink;
  CHECK(hipMalloc(&arena, slot_stride * nslots)); CHECK(hipMalloc(&sink, 8));
  CHECK(hipMemset(arena, 1, slot_stride * nslots));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(nslots), dim3(64), 0, 0, arena, slot_stride, wcap, steps, width, waitmode, sink);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double windows = (double)nslots * steps * 2 * ((width + 247) / 248);
    const double bytes = windows * 12 * 512;
    printf("slots %d width %d steps %d alu %d: %.1f ms  %.3e windows/s  %.2f TB/s (7 loads + 5 stores of 512 B per window)\n", nslots, width, steps, waitmode, ms, windows / (ms * 1e-3), bytes / (ms * 1e-3) / 1e12);
  }
  return 0;
}
