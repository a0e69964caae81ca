// FETCH_SIZE / WRITE_SIZE calibration for the step kernel's access widths.
// Moves a KNOWN number of bytes with exactly the instructions the alignment kernel uses for its
// wavefront rows -- raw_buffer_load_b64 (8 B per lane, default and nt policy), raw_buffer_store_b64,
// and the 16 B per lane forms used with 32-bit rows -- so that bytes_moved / (counter * 1024) can be
// read off a `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` pass of this binary.
// Each mode is its own kernel (name = mode) so the counter CSV separates them.  The footprint (default
// 16 GiB) is far beyond L2 + Infinity Cache, every byte is touched exactly once per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// every workgroup (one wave) owns a contiguous slab; a "row" is 512 B (b64) or 1024 B (b128) per wave access
template <int AUX>
__global__ __launch_bounds__(64, 4) void cal_load_b64(unsigned char* base, size_t slab, unsigned long long* sink) {
  rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base + (size_t)blockIdx.x * slab, 0, (int)slab, 0x00020000);
  unsigned acc = 0;
  for (int off = 0; off < (int)slab; off += 512 * 4) {
    u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(rs, threadIdx.x * 8, off, AUX);
    u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(rs, threadIdx.x * 8, off + 512, AUX);
    u32x2 c = __builtin_amdgcn_raw_buffer_load_b64(rs, threadIdx.x * 8, off + 1024, AUX);
    u32x2 d = __builtin_amdgcn_raw_buffer_load_b64(rs, threadIdx.x * 8, off + 1536, AUX);
    acc += a[0] ^ a[1] ^ b[0] ^ b[1] ^ c[0] ^ c[1] ^ d[0] ^ d[1];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
template <int AUX>
__global__ __launch_bounds__(64, 4) void cal_load_b128(unsigned char* base, size_t slab, unsigned long long* sink) {
  rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base + (size_t)blockIdx.x * slab, 0, (int)slab, 0x00020000);
  unsigned acc = 0;
  for (int off = 0; off < (int)slab; off += 1024 * 2) {
    u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, threadIdx.x * 16, off, AUX);
    u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, threadIdx.x * 16, off + 1024, AUX);
    acc += a[0] ^ a[1] ^ a[2] ^ a[3] ^ b[0] ^ b[1] ^ b[2] ^ b[3];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(64, 4) void cal_store_b64(unsigned char* base, size_t slab, unsigned v) {
  rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base + (size_t)blockIdx.x * slab, 0, (int)slab, 0x00020000);
  u32x2 w;
  w[0] = v + threadIdx.x;
  w[1] = v ^ blockIdx.x;
  for (int off = 0; off < (int)slab; off += 512) __builtin_amdgcn_raw_buffer_store_b64(w, rs, threadIdx.x * 8, off, 0);
}
__global__ __launch_bounds__(64, 4) void cal_store_b128(unsigned char* base, size_t slab, unsigned v) {
  rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base + (size_t)blockIdx.x * slab, 0, (int)slab, 0x00020000);
  u32x4 w;
  w[0] = v + threadIdx.x;
  w[1] = v ^ blockIdx.x;
  w[2] = v;
  w[3] = ~v;
  for (int off = 0; off < (int)slab; off += 1024) __builtin_amdgcn_raw_buffer_store_b128(w, rs, threadIdx.x * 16, off, 0);
}
// the step kernel's own pattern: per 512-B window seven row loads (three default policy, four nt) and five
// stores, rows of a 32-deep ring per workgroup; bytes known exactly = windows * 12 * 512
__global__ __launch_bounds__(64, 4) void cal_step_pattern(unsigned char* arena, size_t slot_stride, int wcap, int steps, int width, unsigned long long* sink) {
  const int lane = threadIdx.x;
  rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(arena + (size_t)blockIdx.x * slot_stride, 0, (int)slot_stride, 0x00020000);
  const int rowb = wcap * 2;
  auto off = [&](int dir, int comp, int score) { return ((dir * 5 + comp) * 32 + (score & 31)) * rowb; };
  unsigned acc = 0;
  for (int s = 32; s < 32 + steps; ++s) {
    for (int dir = 0; dir < 2; ++dir) {
      const int lo = wcap / 2 - width / 2;
      for (int cb = lo; cb < lo + width; cb += 256) {  // disjoint windows: every byte of a row is read once per step
        const int voff = (cb + lane * 4) * 2;
        u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, off(dir, 0, s - 5), 0);
        u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, off(dir, 0, s - 10), 0);
        u32x2 c = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, off(dir, 0, s - 25), 0);
        u32x2 d = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, off(dir, 1, s - 2), 2);
        u32x2 e = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, off(dir, 3, s - 2), 2);
        u32x2 f = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, off(dir, 2, s - 1), 2);
        u32x2 g = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, off(dir, 4, s - 1), 2);
        u32x2 m = a + b + c;
        acc += m[0] ^ m[1];
        __builtin_amdgcn_raw_buffer_store_b64(b + d, rs, voff, off(dir, 1, s), 0);
        __builtin_amdgcn_raw_buffer_store_b64(b + e, rs, voff, off(dir, 3, s), 0);
        __builtin_amdgcn_raw_buffer_store_b64(c + f, rs, voff, off(dir, 2, s), 0);
        __builtin_amdgcn_raw_buffer_store_b64(c + g, rs, voff, off(dir, 4, s), 0);
        __builtin_amdgcn_raw_buffer_store_b64(m, rs, voff, off(dir, 0, s), 0);
      }
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
  const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 16;
  const int nwg = 4096;
  const size_t slab = (gib << 30) / nwg;  // 4 MiB per workgroup at 16 GiB
  unsigned char* buf; unsigned long long* sink;
  CHECK(hipMalloc(&buf, slab * nwg)); CHECK(hipMalloc(&sink, 8));
  CHECK(hipMemset(buf, 1, slab * nwg));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const double bytes = (double)slab * nwg;
  auto timeit = [&](const char* name, auto fn, double b) {
    CHECK(hipEventRecord(e0)); fn(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("{\"kernel\": \"%s\", \"bytes\": %.0f, \"ms\": %.3f, \"TBps\": %.3f}\n", name, b, ms, b / (ms * 1e-3) / 1e12);
  };
  timeit("cal_load_b64<0>", [&] { hipLaunchKernelGGL(cal_load_b64<0>, dim3(nwg), dim3(64), 0, 0, buf, slab, sink); }, bytes);
  timeit("cal_load_b64<2>", [&] { hipLaunchKernelGGL(cal_load_b64<2>, dim3(nwg), dim3(64), 0, 0, buf, slab, sink); }, bytes);
  timeit("cal_load_b128<0>", [&] { hipLaunchKernelGGL(cal_load_b128<0>, dim3(nwg), dim3(64), 0, 0, buf, slab, sink); }, bytes);
  timeit("cal_store_b64", [&] { hipLaunchKernelGGL(cal_store_b64, dim3(nwg), dim3(64), 0, 0, buf, slab, 7u); }, bytes);
  timeit("cal_store_b128", [&] { hipLaunchKernelGGL(cal_store_b128, dim3(nwg), dim3(64), 0, 0, buf, slab, 7u); }, bytes);
  // step pattern: slot = 2 dirs * 5 comps * 32 rows * wcap * 2 B; must fit the slab
  const int wcap = 4096;  // 8 KiB rows -> 2.5 MiB per slot
  const size_t slot = (size_t)2 * 5 * 32 * wcap * 2;
  if (slot <= slab) {
    const int steps = 200, width = 2816;  // 11 windows
    const double windows = (double)nwg * steps * 2 * (width / 256);
    timeit("cal_step_pattern", [&] { hipLaunchKernelGGL(cal_step_pattern, dim3(nwg), dim3(64), 0, 0, buf, slab, wcap, steps, width, sink); }, windows * 12 * 512);
    printf("{\"kernel\": \"cal_step_pattern\", \"read_bytes\": %.0f, \"write_bytes\": %.0f}\n", windows * 7 * 512, windows * 5 * 512);
  }
  return 0;
}
