// kernels_awv.hpp -- the one-wave-per-pair (throughput) kernels live in a translation unit of their own
// (kernels_awv.hip) so that they can be compiled with scheduler options of their own: allwave_amd/build.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#ifndef AWV_THRU_WG
#define AWV_THRU_WG 64  // 128: two waves per pair, one per search direction (AWV_DIRSPLIT)
#endif

// Launches awv::biwfa_align_kernel<two_piece, narrow ? int16_t : int32_t> with `grid` workgroups of AWV_THRU_WG threads and
// `dyn_lds` bytes of dynamic LDS on `stream`; `kparams` points to an awv::KParams.  Returns the hipError_t of the set-up
// (launch errors are picked up by the caller's hipGetLastError, as for the kernels it launches itself).
int awv_launch_one_wave(int two_piece, int narrow, unsigned grid, size_t dyn_lds, hipStream_t stream, const void* kparams);
