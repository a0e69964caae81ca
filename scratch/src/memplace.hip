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
  auto off = [&](int dir, int comp, int score) { return ((dir * 5 + comp) * 32 + (score & 31)) * rowb; };
  unsigned acc = 0;
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  for (int s = 32; s < 32 + steps; ++s) {
    for (int dir = 0; dir < 2; ++dir) {
      const int lo = wcap / 2 - width / 2 + (s & 7) * 4;  // drifting start column
      const int sMx = off(dir, 0, s - 5), sO1 = off(dir, 0, s - 10), sO2 = off(dir, 0, s - 25);
      const int sI1 = off(dir, 1, s - 2), sD1 = off(dir, 3, s - 2), sI2 = off(dir, 2, s - 1), sD2 = off(dir, 4, s - 1);
      const int tM = off(dir, 0, s), tI1 = off(dir, 1, s), tD1 = off(dir, 3, s), tI2 = off(dir, 2, s), tD2 = off(dir, 4, s);
      for (int cb = lo; cb < lo + width; cb += 248) {
        const int voff = (cb + lane * 4) * 2;
        u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, sMx, AM_LD);
        u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, sO1, AM_LD);
        u32x2 c = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, sO2, AUX_MDEEP);
        u32x2 d = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, sI1, AI_LD);
        u32x2 e = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, sD1, AI_LD);
        u32x2 f = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, sI2, AI_LD);
        u32x2 g = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, sD2, AI_LD);
        u32x2 m = a + b + c, i1 = b + d, d1 = b + e, i2 = c + f, d2 = c + g;
        acc += m[0] ^ m[1];
        __builtin_amdgcn_raw_buffer_store_b64(i1, rs, voff, tI1, AI_ST);
        __builtin_amdgcn_raw_buffer_store_b64(d1, rs, voff, tD1, AI_ST);
        __builtin_amdgcn_raw_buffer_store_b64(i2, rs, voff, tI2, AI_ST);
        __builtin_amdgcn_raw_buffer_store_b64(d2, rs, voff, tD2, AI_ST);
        if (waitmode) {  // mimic the dependent ALU/LDS phase between the I/D stores and the M store
          for (int t = 0; t < waitmode; ++t) { m[0] = m[0] * 1664525u + 1013904223u; m[1] ^= m[0] >> 3; }
        }
        __builtin_amdgcn_raw_buffer_store_b64(m, rs, voff, tM, AM_ST);
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
  const int wcap = 10240;
  const size_t slot_stride = (size_t)2 * 5 * 32 * wcap * 2;
  unsigned long long* sink;
  CHECK(hipMalloc(&sink, 8));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  unsigned char* arenas[6];
  for (int a = 0; a < 6; ++a) {   // six arenas alive at once: is the rate a property of where an arena lies?
    CHECK(hipMalloc(&arenas[a], slot_stride * nslots));
    CHECK(hipMemset(arenas[a], 1, slot_stride * nslots));
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k, dim3(nslots), dim3(64), 0, 0, arenas[a], slot_stride, wcap, steps, width, waitmode, sink);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    printf("arena %d at %p: %.2f ms\n", a, (void*)arenas[a], best);
  }
  return 0;
}
