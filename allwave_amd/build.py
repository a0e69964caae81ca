"""In-tree builds of the native libraries (hipcc cross-compiles gfx950 without a GPU).

  liballwave_hip.so   HIP kernels + C ABI (include/allwave_hip.h)      <- the product
  liballwave_host.so  C++ mirror of allwave's host API over the C ABI  <- the product's host side
The CPU oracle under oracle/ has its own Makefile and is test infrastructure; nothing here touches it.
"""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "allwave_amd")
CSRC = os.path.join(PKG, "csrc")
HIP_LIB = os.path.join(PKG, "liballwave_hip.so")
HOST_LIB = os.path.join(PKG, "liballwave_host.so")
CLI_BIN = os.path.join(PKG, "allwave_hip")  # command-line driver (csrc/host/main.cpp)


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


HIP_FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950"]
# kernels_awv.hip (the one-wave-per-pair kernels) only: the instruction scheduler orders for instruction-level parallelism
# rather than for register pressure -- those kernels are bound by VALU issue (config 2, same-box A/B of the whole library:
# 1723-1730 ms against 1762-1780 ms; "max-memory-clause": 1790-1796 ms; for this unit alone "iterative-ilp": 1845 ms,
# "iterative-maxocc": 1779-1785 ms against 1719-1737 ms).  Not for engine.hip: there it spills a lane
# vector inside a pass loop of the 16-bit min(h, v) kernels (scratch/spill_audit.py).
HIP_FLAGS_AWV = HIP_FLAGS + ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
HIP_UNITS = (("engine.hip", HIP_FLAGS), ("kernels_awv.hip", HIP_FLAGS_AWV))


def build_hip(force=False, verbose=False):
    srcs = [os.path.join(CSRC, f) for f in ("engine.hip", "kernels_awv.hip", "kernels_awv.hpp", "biwfa_device.hpp")] + \
           [os.path.join(ROOT, "include", "allwave_hip.h")]
    if force or _newer(HIP_LIB, srcs):
        inc = ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
        objs, procs = [], []
        for unit, flags in HIP_UNITS:  # the two translation units compile side by side
            obj = os.path.join(CSRC, unit.replace(".hip", ".o"))
            cmd = [hipcc()] + flags + ["-fPIC", "-c"] + inc + ["-o", obj, os.path.join(CSRC, unit)]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
            objs.append(obj)
        for cmd, p in procs:
            if p.wait() != 0:
                raise subprocess.CalledProcessError(p.returncode, cmd)
        cmd = [hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", HIP_LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        for obj in objs:
            os.remove(obj)
    return HIP_LIB


def build_host(force=False, verbose=False):
    host_dir = os.path.join(CSRC, "host")
    if not os.path.isdir(host_dir):
        return None
    srcs = [os.path.join(host_dir, f) for f in sorted(os.listdir(host_dir)) if f.endswith((".cpp", ".hpp", ".h"))]
    cpps = [s for s in srcs if s.endswith(".cpp") and not s.endswith("main.cpp")]
    if not cpps:
        return None
    if force or _newer(HOST_LIB, srcs + [os.path.join(ROOT, "include", "allwave_hip.h")]):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I" + os.path.join(ROOT, "include"),
               "-I" + host_dir, "-o", HOST_LIB] + cpps + ["-L" + PKG, "-lallwave_hip", "-Wl,-rpath,$ORIGIN",
                                                           "-Wl,-rpath-link,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    cli_src = os.path.join(host_dir, "main.cpp")
    if os.path.exists(cli_src) and (force or _newer(CLI_BIN, srcs)):
        cmd = ["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "include"), "-I" + host_dir, "-o", CLI_BIN,
               cli_src, "-L" + PKG, "-lallwave_host", "-lallwave_hip", "-lz", "-Wl,-rpath,$ORIGIN",
               "-Wl,-rpath-link,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HOST_LIB


def build_all(force=False, verbose=False):
    return build_hip(force, verbose), build_host(force, verbose)


if __name__ == "__main__":
    print(build_all(force=True, verbose=True))
