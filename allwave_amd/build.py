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


def build_hip(force=False, verbose=False):
    srcs = [os.path.join(CSRC, f) for f in ("engine.hip", "biwfa_device.hpp")] + \
           [os.path.join(ROOT, "include", "allwave_hip.h")]
    if force or _newer(HIP_LIB, srcs):
        cmd = [hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
               "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", HIP_LIB, os.path.join(CSRC, "engine.hip")]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
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
