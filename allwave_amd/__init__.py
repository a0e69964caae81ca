"""allwave_amd -- MI355X-native (gfx950) replacement for allwave's per-pair BiWFA hot path.

Only what the path needs lives here: csrc/ (HIP kernels + the C ABI of include/allwave_hip.h and the
C++ host mirror of allwave's API), ffi.py (ctypes binding), synth.py (bench inputs), build.py.
"""
__all__ = ["ffi", "synth", "build"]
