"""In-tree builds of the native libraries (hipcc cross-compiles gfx950 without a GPU).

  liballwave_hip.so   HIP kernels + C ABI (include/allwave_hip.h)      <- the product
  liballwave_host.so  C++ mirror of allwave's host API over the C ABI  <- the product's host side
  allwave_hip         command-line driver (csrc/host/main.cpp)
The CPU oracle under oracle/ has its own Makefile and is test infrastructure; nothing here touches it.

Staleness is decided by CONTENT, not by file times: every target records the SHA-256 of its sources, its
command lines and the compiler's version in BUILD_INFO.json (next to the binaries, git-ignored like them, and
travelling with them to the GPU box); a target is rebuilt when that hash differs or the file is missing -- a
checkout that resets mtimes can neither force a rebuild nor hide a stale binary.  Objects are compiled in a
per-process temporary directory and the finished file is moved into place (os.replace), so two builds at once
cannot mix their objects or expose a half-written library.
"""
import hashlib
import json
import os
import shutil
import subprocess
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "allwave_amd")
CSRC = os.path.join(PKG, "csrc")
HOST_DIR = os.path.join(CSRC, "host")
HIP_LIB = os.path.join(PKG, "liballwave_hip.so")
HOST_LIB = os.path.join(PKG, "liballwave_host.so")
CLI_BIN = os.path.join(PKG, "allwave_hip")  # command-line driver (csrc/host/main.cpp)
INFO = os.path.join(PKG, "BUILD_INFO.json")
ABI_HEADER = os.path.join(ROOT, "include", "allwave_hip.h")


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
HIP_SOURCES = [os.path.join(CSRC, f) for f in ("engine.hip", "kernels_awv.hip", "kernels_awv.hpp", "biwfa_device.hpp")] + [ABI_HEADER]


def _host_sources():
    return [os.path.join(HOST_DIR, f) for f in sorted(os.listdir(HOST_DIR)) if f.endswith((".cpp", ".hpp", ".h"))]


_VERSIONS = {}


def _tool_version(tool):
    if tool not in _VERSIONS:
        try:
            _VERSIONS[tool] = subprocess.run([tool, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode(errors="replace")
        except OSError:
            _VERSIONS[tool] = "?"
    return _VERSIONS[tool]


def content_hash(sources, commands, tool):
    """SHA-256 over the sources' bytes (by repo-relative name), the command lines and the compiler version."""
    h = hashlib.sha256()
    for s in sorted(sources):
        h.update(os.path.relpath(s, ROOT).encode() + b"\0")
        with open(s, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    for c in commands:
        h.update(" ".join(c).encode() + b"\n")
    h.update(_tool_version(tool).encode())
    return h.hexdigest()


def _load_info():
    try:
        with open(INFO) as f:
            d = json.load(f)
        return d if isinstance(d.get("targets"), dict) else {"targets": {}}
    except (OSError, ValueError):
        return {"targets": {}}


def _record(target, digest, action):
    info = _load_info()
    info["targets"][os.path.basename(target)] = {"sha256": digest, "action": action, "when": time.strftime("%Y-%m-%dT%H:%M:%S")}
    tmp = INFO + ".%d.tmp" % os.getpid()
    with open(tmp, "w") as f:
        json.dump(info, f, indent=1)
    os.replace(tmp, INFO)


def _up_to_date(target, digest):
    return os.path.exists(target) and _load_info()["targets"].get(os.path.basename(target), {}).get("sha256") == digest


def _rel(cmd):
    """Command line with repo-relative paths: what goes into the hash (the GPU box holds the tree elsewhere)."""
    return [a.replace(ROOT + os.sep, "") for a in cmd]


def build_hip(force=False, verbose=False):
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    compile_cmds = [[hipcc()] + flags + ["-fPIC", "-c"] + inc + [os.path.join(CSRC, unit)] for unit, flags in HIP_UNITS]
    link_cmd = [hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared"]
    digest = content_hash(HIP_SOURCES, [_rel(c) for c in compile_cmds] + [_rel(link_cmd)], hipcc())
    if not force and _up_to_date(HIP_LIB, digest):
        _record(HIP_LIB, digest, "reused (content hash of sources + flags + compiler matches)")
        return HIP_LIB
    tmp = tempfile.mkdtemp(prefix="awv_build_%d_" % os.getpid())
    try:
        objs, procs = [], []
        for (unit, _), cmd in zip(HIP_UNITS, compile_cmds):  # the two translation units compile side by side
            obj = os.path.join(tmp, unit.replace(".hip", ".o"))
            full = cmd + ["-o", obj]
            if verbose:
                print(" ".join(full))
            procs.append((full, subprocess.Popen(full)))
            objs.append(obj)
        failed = None
        for full, p in procs:  # every child is waited for before anything is raised: none keeps running behind an error
            if p.wait() != 0 and failed is None:
                failed = (p.returncode, full)
        if failed:
            raise subprocess.CalledProcessError(failed[0], failed[1])
        out = os.path.join(tmp, "liballwave_hip.so")
        full = link_cmd + ["-o", out] + objs
        if verbose:
            print(" ".join(full))
        subprocess.check_call(full)
        staged = HIP_LIB + ".%d.tmp" % os.getpid()  # (the temporary directory may be on another filesystem: copy next to the target, then rename)
        shutil.copy2(out, staged)
        os.replace(staged, HIP_LIB)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    _record(HIP_LIB, digest, "compiled")
    return HIP_LIB


def _build_one(target, sources, cmd, force, verbose, tool="g++"):
    digest = content_hash(sources, [_rel(cmd)], tool)
    if not force and _up_to_date(target, digest):
        _record(target, digest, "reused (content hash of sources + flags + compiler matches)")
        return
    tmp = target + ".%d.tmp" % os.getpid()
    full = cmd[:cmd.index("-o") + 1] + [tmp] + cmd[cmd.index("-o") + 2:]
    if verbose:
        print(" ".join(full))
    try:
        subprocess.check_call(full)
        os.replace(tmp, target)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    _record(target, digest, "compiled")


def build_host(force=False, verbose=False):
    if not os.path.isdir(HOST_DIR):
        return None
    srcs = _host_sources()
    cpps = [s for s in srcs if s.endswith(".cpp") and not s.endswith("main.cpp")]
    if not cpps:
        return None
    # both depend on the ABI header (AllPairIterator embeds an awv_stats: a header-only change must rebuild the library
    # AND the driver, or the two disagree on the object's layout); liballwave_hip itself is linked dynamically
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I" + os.path.join(ROOT, "include"), "-I" + HOST_DIR,
           "-o", HOST_LIB] + cpps + ["-L" + PKG, "-lallwave_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link,/opt/rocm/lib"]
    _build_one(HOST_LIB, srcs + [ABI_HEADER], cmd, force, verbose)
    cli_src = os.path.join(HOST_DIR, "main.cpp")
    if os.path.exists(cli_src):
        cmd = ["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "include"), "-I" + HOST_DIR, "-o", CLI_BIN, cli_src,
               "-L" + PKG, "-lallwave_host", "-lallwave_hip", "-lz", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link,/opt/rocm/lib"]
        _build_one(CLI_BIN, srcs + [ABI_HEADER], cmd, force, verbose)
    return HOST_LIB


def build_all(force=False, verbose=False):
    return build_hip(force, verbose), build_host(force, verbose)


def build_report():
    """{target: action} of the last build_all, for __graft_entry__.build()'s record."""
    t = _load_info()["targets"]
    return {k: "%s [sha256 %s]" % (v.get("action", "?"), v.get("sha256", "")[:12]) for k, v in sorted(t.items())}


if __name__ == "__main__":
    print(build_all(force=True, verbose=True))
    print(json.dumps(build_report(), indent=1))
