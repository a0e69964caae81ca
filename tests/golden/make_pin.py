"""Writes tests/golden/pin/: two small FASTA read sets and the PAF lines this build produces for them, so that whoever has a
built allwave (cargo + lib_wfa2: neither exists in this build environment) can pin this build against the reference in one
command -- tests/golden/pin/pin.sh runs `allwave` on the same FASTA inputs and diffs its PAF against the committed lines.

PROVENANCE: the expected lines come from the build's own CPU oracle (oracle/biwfa_oracle.c) formatted as the reference's
alignment_to_paf does (/root/reference/src/lib.rs:71-112); they are NOT outputs of the reference (parity vs WFA2-lib is
unpinned until pin.sh or integration/hip_parity.rs has been run somewhere with cargo).  The GPU suite checks that the HIP
command-line driver reproduces these files byte for byte (tests/test_planner.py::test_cli_reproduces_the_pin_files).

  c1.fa          BASELINE configs[0]: 8 x 1 kbp, 5 %, scores 0,1,1,1, -p none  -> 56 lines
  c2_8x10k.fa    the first 8 reads of configs[1] (256 x 10 kbp, 5 %), default scores 0,5,8,2,24,1, -p none -> 56 lines
Usage: python tests/golden/make_pin.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import oracle as O  # noqa: E402
from allwave_amd import synth  # noqa: E402
from util import rle  # noqa: E402

SETS = (("c1", "c1", 8, (0, 1, 1, 1)), ("c2_8x10k", "c2", 8, (0, 5, 8, 2, 24, 1)))


def paf_line(ids, seqs, i, j, al):
    """alignment_to_paf (lib.rs:71-112) of the oracle's alignment of (query i, target j), forward strand"""
    pen, ops = al.align(seqs[i], seqs[j])
    m, x = ops.count(b"M"), ops.count(b"X")
    qe, te = m + x + ops.count(b"D"), m + x + ops.count(b"I")
    ident = (m / (m + x)) if (m + x) else 0.0
    return "%s\t%d\t0\t%d\t+\t%s\t%d\t0\t%d\t%d\t%d\t60\tgi:f:%.6f\tcg:Z:%s" % (
        ids[i], len(seqs[i]), qe, ids[j], len(seqs[j]), te, m, max(qe, te), ident, rle(ops))


def build(name, cfg_name, nseq, scores):
    cfg = synth.CONFIGS[cfg_name]
    data, offs, ids = synth.generate(cfg["nseq"], cfg["length"], cfg["d"], cfg["seed"])
    seqs = [bytes(data[offs[i]:offs[i + 1]]) for i in range(nseq)]
    ids = ids[:nseq]
    fasta = "".join(">%s\n%s\n" % (i, s.decode()) for i, s in zip(ids, seqs))
    al = O.Aligner(scores)
    lines = [paf_line(ids, seqs, i, j, al) for i in range(nseq) for j in range(nseq) if i != j]
    return fasta, "".join(l + "\n" for l in lines)


def main():
    out = os.path.join(HERE, "pin")
    os.makedirs(out, exist_ok=True)
    for name, cfg_name, nseq, scores in SETS:
        fasta, paf = build(name, cfg_name, nseq, scores)
        open(os.path.join(out, name + ".fa"), "w").write(fasta)
        open(os.path.join(out, name + ".expected.paf"), "w").write(paf)
        print(name, len(paf.splitlines()), "lines")


if __name__ == "__main__":
    main()
