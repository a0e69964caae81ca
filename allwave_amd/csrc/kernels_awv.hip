// kernels_awv.hip -- namespace awv:: of biwfa_device.hpp (one wave per pair: the throughput kernels) and their launcher.
// A translation unit of its own: these kernels are bound by VALU issue and gain 2-3 % from the compiler's
// "max-ilp" scheduling strategy, which in the multi-wave kernels of engine.hip spills a lane vector inside the pass
// loop (not allowed) -- so the option is given to this file only (allwave_amd/build.py).
#include "kernels_awv.hpp"
#define AWV_NS awv
#define AWV_WG AWV_THRU_WG
#if AWV_THRU_WG == 128
#define AWV_DIRSPLIT 1
#endif
#include "biwfa_device.hpp"

namespace {
template <typename K>
int launch(K kern, unsigned grid, size_t dyn_lds, hipStream_t stream, const awv::KParams& kp) {
  const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(AWV_THRU_WG), dyn_lds, stream, kp);
  return (int)hipSuccess;
}
}  // namespace

int awv_launch_one_wave(int two_piece, int narrow, unsigned grid, size_t dyn_lds, hipStream_t stream, const void* kparams) {
  const awv::KParams& kp = *static_cast<const awv::KParams*>(kparams);
  if (two_piece) return narrow ? launch(awv::biwfa_align_kernel<true, int16_t>, grid, dyn_lds, stream, kp)
                               : launch(awv::biwfa_align_kernel<true, int32_t>, grid, dyn_lds, stream, kp);
  return narrow ? launch(awv::biwfa_align_kernel<false, int16_t>, grid, dyn_lds, stream, kp)
                : launch(awv::biwfa_align_kernel<false, int32_t>, grid, dyn_lds, stream, kp);
}
