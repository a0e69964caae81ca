// VALU issue cost by instruction kind, second table (round 3): valurate.hip found v_add_u32 / v_fma_f32 at ~2.7 cycles per
// wave-instruction (4 waves per SIMD, wall clock) where v_max_i32 / packed 16-bit / v_alignbit / v_med3 / v_ffbl cost ~4.3.  Which
// kinds sit on which side decides how the step code should be written.  Same harness: 8 independent chains, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define KINDS(X) \
  X(0, "v_add_u32", "v_add_u32 %0, %0, %1") \
  X(1, "v_sub_u32", "v_sub_u32 %0, %0, %1") \
  X(2, "v_and_b32", "v_and_b32 %0, %0, %1") \
  X(3, "v_or_b32", "v_or_b32 %0, %0, %1") \
  X(4, "v_xor_b32", "v_xor_b32 %0, %0, %1") \
  X(5, "v_lshlrev_b32", "v_lshlrev_b32 %0, 3, %1") \
  X(6, "v_lshrrev_b32", "v_lshrrev_b32 %0, 3, %1") \
  X(7, "v_ashrrev_i32", "v_ashrrev_i32 %0, 3, %1") \
  X(8, "v_min_i32", "v_min_i32 %0, %0, %1") \
  X(9, "v_max_i32", "v_max_i32 %0, %0, %1") \
  X(10, "v_min_u32", "v_min_u32 %0, %0, %1") \
  X(11, "v_max_u32", "v_max_u32 %0, %0, %1") \
  X(12, "v_bfe_i32", "v_bfe_i32 %0, %1, 0, 16") \
  X(13, "v_bfe_u32", "v_bfe_u32 %0, %1, 4, 12") \
  X(14, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 2, %1") \
  X(15, "v_add3_u32", "v_add3_u32 %0, %0, %1, 7") \
  X(16, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 16, %1") \
  X(17, "v_and_or_b32", "v_and_or_b32 %0, %0, 63, %1") \
  X(18, "v_mov_b32", "v_mov_b32 %0, %1") \
  X(19, "v_perm_b32", "v_perm_b32 %0, %0, %1, %1") \
  X(20, "v_min3_i32", "v_min3_i32 %0, %0, %1, 9") \
  X(21, "v_max3_i32", "v_max3_i32 %0, %0, %1, 9") \
  X(22, "v_med3_i32", "v_med3_i32 %0, %0, -1, %1") \
  X(23, "v_pk_max_i16", "v_pk_max_i16 %0, %0, %1") \
  X(24, "v_pk_min_i16", "v_pk_min_i16 %0, %0, %1") \
  X(25, "v_pk_add_u16", "v_pk_add_u16 %0, %0, %1") \
  X(26, "v_pk_sub_i16", "v_pk_sub_i16 %0, %0, %1") \
  X(27, "v_pk_min_u16", "v_pk_min_u16 %0, %0, %1") \
  X(28, "v_ffbl_b32", "v_ffbl_b32 %0, %1") \
  X(29, "v_ffbh_u32", "v_ffbh_u32 %0, %1") \
  X(30, "v_alignbit_b32", "v_alignbit_b32 %0, %0, %1, %1") \
  X(31, "v_alignbyte_b32", "v_alignbyte_b32 %0, %0, %1, 2") \
  X(32, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %1") \
  X(33, "v_mul_u32_u24", "v_mul_u32_u24 %0, %0, %1") \
  X(34, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1") \
  X(35, "v_bfi_b32", "v_bfi_b32 %0, %0, %1, %1") \
  X(36, "v_cmp_gt_i32 (to vcc)", "v_cmp_gt_i32 vcc, %0, %1") \
  X(37, "v_cndmask_b32 (vcc)", "v_cndmask_b32 %0, %0, %1, vcc") \
  X(38, "v_mov_b32_dpp wave_shr", "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1") \
  X(39, "v_readlane_b32 (to sgpr)", "v_readlane_b32 s20, %1, 3") \
  X(40, "v_max_i16", "v_max_i16 %0, %0, %1") \
  X(41, "v_add_u16", "v_add_u16 %0, %0, %1") \
  X(42, "v_sad_u32", "v_sad_u32 %0, %0, %1, %1") \
  X(43, "v_fma_f32", "v_fma_f32 %0, %0, %1, %1") \
  X(44, "v_add_f32", "v_add_f32 %0, %0, %1") \
  X(45, "v_max_f32", "v_max_f32 %0, %0, %1") \
  X(46, "v_pk_add_f32 (2 regs)", "v_pk_add_f32 %0, %0, %0") \
  X(47, "v_xad_u32", "v_xad_u32 %0, %0, %1, %1") \
  X(48, "v_sub_u32 from sgpr", "v_sub_u32 %0, s20, %1") \
  X(49, "v_cmp_gt_i32 (to sgpr pair)", "v_cmp_gt_i32 s[22:23], %0, %1")

template <int KIND>
__global__ __launch_bounds__(64, 4) void rate(int iters, unsigned* out, unsigned long long* cyc, unsigned long long* ticks) {
  unsigned r[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
  unsigned long long q = ((unsigned long long)r[0] << 32) | r[1];
  const unsigned long long t0 = __builtin_readcyclecounter(), w0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#define X(id, name, text) if (KIND == id) { if (id == 46) asm volatile(text : "+v"(q)); else asm volatile(text : "+v"(r[i]) : "v"(r[(i + 1) & 7]) : "vcc", "s20", "s22", "s23"); }
        KINDS(X)
#undef X
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), w1 = __builtin_amdgcn_s_memrealtime();
  unsigned acc = (unsigned)q;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc ^= r[i];
  out[blockIdx.x * 64 + threadIdx.x] = acc;
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; ticks[blockIdx.x] = w1 - w0; }
}

template <int KIND>
void run(const char* name, int waves_per_simd, int iters, unsigned* out, unsigned long long* cyc, unsigned long long* ticks, hipEvent_t e0, hipEvent_t e1, int ncu) {
  const int blocks = ncu * 4 * waves_per_simd;
  hipLaunchKernelGGL(rate<KIND>, dim3(blocks), dim3(64), 0, 0, 10, out, cyc, ticks);
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(rate<KIND>, dim3(blocks), dim3(64), 0, 0, iters, out, cyc, ticks);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long hc[64], ht[64];
  CHECK(hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(ht, ticks, sizeof(ht), hipMemcpyDeviceToHost));
  double mean = 0, tick = 0;
  for (int i = 0; i < 64; ++i) { mean += (double)hc[i]; tick += (double)ht[i]; }
  const double ghz = tick > 0 ? mean / tick * 0.1 : 0.0;
  const double n_inst = (double)iters * 32;
  printf("{\"inst\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"clock_ghz_measured\": %.3f, \"simd_cycles_per_wave_inst_by_wall_clock\": %.2f}\n",
         name, waves_per_simd, ms, ghz, ms * 1e-3 * ghz * 1e9 / (n_inst * waves_per_simd));
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  unsigned* out; unsigned long long *cyc, *ticks;
  CHECK(hipMalloc(&out, (size_t)ncu * 16 * 64 * 4)); CHECK(hipMalloc(&cyc, (size_t)ncu * 16 * 8)); CHECK(hipMalloc(&ticks, (size_t)ncu * 16 * 8));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 20000;
#define X(id, name, text) run<id>(name, 4, iters, out, cyc, ticks, e0, e1, ncu);
  KINDS(X)
#undef X
  return 0;
}
