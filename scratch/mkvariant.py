"""Builds an alternative liballwave_hip (extra -D / -mllvm options) next to the product library, for same-box A/Bs:
    python scratch/mkvariant.py <name> [-DAWV_FOO=1 ...]      -> scratch/bin/liballwave_hip_<name>.so
Used with `scratch/exp.py --lib scratch/bin/liballwave_hip_<name>.so` (scratch/r02_ab.sh runs several in one gpurun call).
Objects go to a per-variant temporary directory, so several variants can be built side by side."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from allwave_amd import build as B

name, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "scratch", "bin", "liballwave_hip_%s.so" % name)
os.makedirs(os.path.dirname(out), exist_ok=True)
tmp = tempfile.mkdtemp(prefix="awv_" + name + "_")
inc = ["-I" + os.path.join(ROOT, "include"), "-I" + B.CSRC]
procs, objs = [], []
for unit, flags in B.HIP_UNITS:
    obj = os.path.join(tmp, unit.replace(".hip", ".o"))
    procs.append(subprocess.Popen([B.hipcc()] + list(flags) + extra + ["-fPIC", "-c"] + inc + ["-o", obj, os.path.join(B.CSRC, unit)]))
    objs.append(obj)
if any(p.wait() != 0 for p in procs):
    sys.exit("compile failed")
subprocess.check_call([B.hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out] + objs)
for o in objs:
    os.remove(o)
os.rmdir(tmp)
print(out)
