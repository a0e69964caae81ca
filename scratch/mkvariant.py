"""Builds an alternative liballwave_hip (extra -D / -mllvm options) next to the product library, for same-box A/Bs:
    python scratch/mkvariant.py <name> [--awv-only] [-DAWV_FOO=1 ...]      -> scratch/bin/liballwave_hip_<name>.so
--awv-only: the options go to kernels_awv.hip alone (the one-wave throughput kernels: what config 2 runs); engine.hip's object is
built once without them and reused (scratch/bin/engine_base_<hash>.o) -- a variant then takes one compile instead of two.
Used with `scratch/exp.py --lib scratch/bin/liballwave_hip_<name>.so` (scratch/r03_try.sh runs several in one gpurun call).
Objects go to a per-variant temporary directory, so several variants can be built side by side."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from allwave_amd import build as B

args = sys.argv[1:]
name = args.pop(0)
awv_only = "--awv-only" in args
extra = [a for a in args if a != "--awv-only"]
bindir = os.path.join(ROOT, "scratch", "bin")
out = os.path.join(bindir, "liballwave_hip_%s.so" % name)
os.makedirs(bindir, exist_ok=True)
tmp = tempfile.mkdtemp(prefix="awv_" + name + "_")
inc = ["-I" + os.path.join(ROOT, "include"), "-I" + B.CSRC]
procs, objs = [], []
for unit, flags in B.HIP_UNITS:
    src = os.path.join(B.CSRC, unit)
    if awv_only and unit == "engine.hip":
        digest = B.content_hash(B.HIP_SOURCES, [list(flags)], B.hipcc())[:16]
        obj = os.path.join(bindir, "engine_base_%s.o" % digest)
        if not os.path.exists(obj):
            procs.append(subprocess.Popen([B.hipcc()] + list(flags) + ["-fPIC", "-c"] + inc + ["-o", obj, src]))
        objs.append(obj)
        continue
    obj = os.path.join(tmp, unit.replace(".hip", ".o"))
    procs.append(subprocess.Popen([B.hipcc()] + list(flags) + extra + ["-fPIC", "-c"] + inc + ["-o", obj, src]))
    objs.append(obj)
if any(p.wait() != 0 for p in procs):
    sys.exit("compile failed")
subprocess.check_call([B.hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out] + objs)
for o in objs:
    if o.startswith(tmp):
        os.remove(o)
os.rmdir(tmp)
print(out)
