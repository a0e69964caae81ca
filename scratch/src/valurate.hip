// VALU issue-rate probe for the instruction kinds the step kernel is made of: how many shader cycles
// one wave-instruction costs a SIMD at 1, 2 and 4 resident waves per SIMD.  This is what turns
// SQ_INSTS_VALU into a VALU-roofline fraction (rocprof's SQ_ACTIVE_INST_VALU is a count in quad-cycles,
// i.e. it assumes 4 cycles per instruction whatever the real issue cost).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

enum { K_ADD = 0, K_MAX, K_PKMAX, K_PKADD, K_ALIGNBIT, K_DPP, K_MED3, K_CNDMASK, K_FFBL, K_MIX, K_FMA, K_N };
static const char* names[K_N] = {"v_add_u32", "v_max_i32", "v_pk_max_i16", "v_pk_add_u16", "v_alignbit_b32", "v_mov_b32_dpp(wave_shr)",
                                 "v_med3_i32", "v_cmp+v_cndmask", "v_ffbl_b32", "step-kernel mix", "v_fma_f32 (the guide's reference row)"};

template <int KIND>
__global__ __launch_bounds__(64, 4) void rate(int iters, unsigned* out, unsigned long long* cyc, unsigned long long* ticks) {
  unsigned r[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
  const unsigned long long t0 = __builtin_readcyclecounter(), w0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == K_ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
        if (KIND == K_MAX) asm volatile("v_max_i32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
        if (KIND == K_PKMAX) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
        if (KIND == K_PKADD) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
        if (KIND == K_ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
        if (KIND == K_DPP) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
        if (KIND == K_MED3) asm volatile("v_med3_i32 %0, %0, -1, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
        if (KIND == K_CNDMASK) asm volatile("v_cmp_gt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(r[(i + 1) & 7]) : "vcc");
        if (KIND == K_FFBL) asm volatile("v_ffbl_b32 %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
        if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
        if (KIND == K_MIX) {  // roughly the step kernel's blend: packed max/add, alignbit, dpp, 32-bit compare/select, shifts
          switch (i) {
            case 0: asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(r[i]) : "v"(r[1])); break;
            case 1: asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(r[i]) : "v"(r[2])); break;
            case 2: asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i]) : "v"(r[3])); break;
            case 3: asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r[i]) : "v"(r[4])); break;
            case 4: asm volatile("v_cmp_gt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(r[5]) : "vcc"); break;
            case 5: asm volatile("v_lshrrev_b32 %0, 4, %1" : "+v"(r[i]) : "v"(r[6])); break;
            case 6: asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(r[7])); break;
            default: asm volatile("v_min_i32 %0, %0, %1" : "+v"(r[i]) : "v"(r[0])); break;
          }
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), w1 = __builtin_amdgcn_s_memrealtime();
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc ^= r[i];
  out[blockIdx.x * 64 + threadIdx.x] = acc;
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; ticks[blockIdx.x] = w1 - w0; }
}

template <int KIND>
void run(int waves_per_simd, int iters, unsigned* out, unsigned long long* cyc, unsigned long long* ticks, hipEvent_t e0, hipEvent_t e1, int ncu) {
  const int blocks = ncu * 4 * waves_per_simd;
  hipLaunchKernelGGL(rate<KIND>, dim3(blocks), dim3(64), 0, 0, 10, out, cyc, ticks);  // warm
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(rate<KIND>, dim3(blocks), dim3(64), 0, 0, iters, out, cyc, ticks);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long hc[64];
  CHECK(hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost));
  unsigned long long ht[64];
  CHECK(hipMemcpy(ht, ticks, sizeof(ht), hipMemcpyDeviceToHost));
  double mean = 0, tick = 0;
  for (int i = 0; i < 64; ++i) { mean += (double)hc[i]; tick += (double)ht[i]; }
  mean /= 64;
  tick /= 64;
  const double ghz = tick > 0 ? mean / tick * 0.1 : 0.0;  // shader cycles per 100 MHz tick
  const double n_inst = (double)iters * 32 * (KIND == K_CNDMASK ? 2 : (KIND == K_MIX ? 36.0 / 32 : 1));
  // per-wave cycles / instruction and the SIMD's cost per wave-instruction (= per-wave cycles / waves on the SIMD)
  // the SIMD's cost per wave-instruction by WALL CLOCK: kernel time x measured clock / (instructions per wave x waves on the SIMD)
  printf("{\"inst\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"wave_cycles_per_inst\": %.2f, \"clock_ghz_measured\": %.3f, "
         "\"simd_cycles_per_wave_inst_by_wall_clock\": %.2f}\n",
         names[KIND], waves_per_simd, ms, mean / n_inst, ghz, ms * 1e-3 * ghz * 1e9 / (n_inst * waves_per_simd));
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  unsigned* out; unsigned long long *cyc, *ticks;
  CHECK(hipMalloc(&out, (size_t)ncu * 16 * 64 * 4)); CHECK(hipMalloc(&cyc, (size_t)ncu * 16 * 8)); CHECK(hipMalloc(&ticks, (size_t)ncu * 16 * 8));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("{\"cus\": %d, \"clock_khz\": %d}\n", ncu, prop.clockRate);
  const int iters = 20000;
  for (int w : {1, 2, 4}) {
    run<K_ADD>(w, iters, out, cyc, ticks, e0, e1, ncu);
    run<K_MAX>(w, iters, out, cyc, ticks, e0, e1, ncu);
    run<K_PKMAX>(w, iters, out, cyc, ticks, e0, e1, ncu);
    run<K_PKADD>(w, iters, out, cyc, ticks, e0, e1, ncu);
    run<K_ALIGNBIT>(w, iters, out, cyc, ticks, e0, e1, ncu);
    run<K_DPP>(w, iters, out, cyc, ticks, e0, e1, ncu);
    run<K_MED3>(w, iters, out, cyc, ticks, e0, e1, ncu);
    run<K_CNDMASK>(w, iters, out, cyc, ticks, e0, e1, ncu);
    run<K_FFBL>(w, iters, out, cyc, ticks, e0, e1, ncu);
    run<K_MIX>(w, iters, out, cyc, ticks, e0, e1, ncu);
    run<K_FMA>(w, iters, out, cyc, ticks, e0, e1, ncu);
  }
  return 0;
}
