"""Register spills inside the pass functions of every shipped kernel (a lane vector spilled inside
compute_rows_multi's window loop, written under a partial EXEC mask, once returned wrong results).  Compiles engine.hip
to device assembly and lists, per multi_phase / base_phase instantiation, the `Folded Spill` / `Folded Reload` lines that
lie away from the function's prologue and epilogue.   usage: python scratch/spill_audit.py"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from allwave_amd import build as B
lines = []
for unit, flags in B.HIP_UNITS:  # every translation unit with the options the library is built with
    out = os.path.join(tempfile.gettempdir(), "awv_" + unit + ".s")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "allwave_amd", "csrc"),
                           "-S", "--cuda-device-only", os.path.join(ROOT, "allwave_amd", "csrc", unit), "-o", out], stderr=subprocess.DEVNULL)
    lines += open(out).read().split("\n")
funcs = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"^(_Z\w+):\s", l)] if m] + [(len(lines), "END")]
worst = 0
for (a, name), (b, _) in zip(funcs, funcs[1:]):
    if "multi_phase" not in name and "base_phase" not in name and "deep_phase" not in name:
        continue
    body = lines[a:b]
    n = len(body)
    mid = [j for j, l in enumerate(body) if ("Folded Spill" in l or "Folded Reload" in l) and 0.08 * n < j < 0.90 * n]
    late = [j for j in mid if j > 0.22 * n]  # (the window loop starts about a fifth into these functions)
    worst = max(worst, len(late))
    print("%3d in the body, %3d inside the loop region  %s" % (len(mid), len(late), name))
sys.exit(1 if worst else 0)
