"""Host-side mirror of allwave's API (csrc/host): CPU tests for the pure host logic, GPU tests for
the AllPairIterator / PAF / wfa::align_sequences paths.  Expected values restate the reference's own
unit tests (src/lib.rs:155-193, src/validation_correct.rs:139-175) and integration properties
(tests/integration_tests.rs)."""
import random

import pytest

from util import DEFAULT_2P, EDIT, mutate, rand_seq, rle


@pytest.fixture(scope="module")
def host(hip_lib):
    from allwave_amd import build, host as H
    build.build_host()
    H.load()
    return H


def test_parse_scores(host):
    """src/lib.rs:159-177"""
    assert host.parse_scores("0,1,1,1") == (0, 1, 1, 1)
    assert host.parse_scores("0,5,8,2,24,1") == (0, 5, 8, 2, 24, 1)
    assert host.parse_scores(" 0, 5 ,8,2 ") == (0, 5, 8, 2)
    with pytest.raises(ValueError, match="Invalid number of scores: 3. Expected 4 or 6 values."):
        host.parse_scores("0,1,1")
    with pytest.raises(ValueError, match="Failed to parse scores"):
        host.parse_scores("0,x,1,1")


def test_alignment_mode_detection(host):
    """src/lib.rs:179-192 + alignment.rs:263-289: "edit" builds gap-affine (x,x,x)."""
    assert host.mode_and_penalties("0,1,1,1") == ("EditDistance", (0, 1, 1, 1, 0, 0, 0))
    assert host.mode_and_penalties("0,3,3,3") == ("EditDistance", (0, 3, 3, 3, 0, 0, 0))
    assert host.mode_and_penalties("0,5,8,2") == ("SinglePieceAffine", (0, 5, 8, 2, 0, 0, 0))
    assert host.mode_and_penalties("0,5,8,2,24,1") == ("TwoPieceAffine", (0, 5, 8, 2, 24, 1, 1))


def test_cigar_string_and_rc(host):
    assert host.cigar_bytes_to_string(b"MMMXMIIDM") == "3=1X1=2D1I1="   # I<->D swap, M -> '='
    assert host.cigar_bytes_to_string(b"") == ""
    assert host.cigar_bytes_to_string(b"MMQ") == "2=1?"
    assert host.reverse_complement(b"ACGTNacgtx") == b"NACGTNACGT"


def test_cigar_run_lengths_every_alignment(host):
    """The run-length encoder scans eight op bytes at a time: runs that start, end and straddle every offset inside a
    word, runs of 1 .. 40 and a few long ones, against a byte-by-byte restatement of alignment.rs:347-376."""
    import random
    sym = {ord("M"): "=", ord("X"): "X", ord("I"): "D", ord("D"): "I"}

    def rle(ops):
        out, i = [], 0
        while i < len(ops):
            j = i
            while j < len(ops) and ops[j] == ops[i]:
                j += 1
            out.append("%d%s" % (j - i, sym.get(ops[i], "?")))
            i = j
        return "".join(out)

    rng = random.Random(8)
    for lead in range(0, 9):
        for run in list(range(1, 41)) + [63, 64, 65, 1000, 12345]:
            ops = b"X" * lead + b"M" * run + b"I" + b"D" * (run % 7) + b"M"
            assert host.cigar_bytes_to_string(ops) == rle(ops), (lead, run)
    for _ in range(200):
        ops = b"".join(bytes([rng.choice(b"MMMMXID")]) * rng.choice([1, 1, 2, 3, 7, 8, 9, 15, 16, 17, 50]) for _ in range(rng.randint(1, 60)))
        assert host.cigar_bytes_to_string(ops) == rle(ops)


def test_validate_cigar(host):
    """validation_correct.rs:139-175 semantics via wfa.rs:105-176: I consumes the reference, D the query."""
    assert host.validate_cigar(b"MMMM", 4, 4) is None
    assert host.validate_cigar(b"MMXM", 4, 4) is None
    assert host.validate_cigar(b"MMIIMM", 4, 6) is None
    assert host.validate_cigar(b"MMDDMM", 6, 4) is None
    assert "doesn't cover full query" in host.validate_cigar(b"MMM", 4, 3)
    assert "Invalid CIGAR operation" in host.validate_cigar(b"MMZ", 3, 3)


def test_alignment_to_paf_format(host):
    """lib.rs:95-111 field layout, identity = matches / (M + X), block_len = max(q_aligned, t_aligned)."""
    line = host.format_paf("q1", 10, "t1", 12, 0, 10, 0, 12, False, 9, 10, b"MMMMMXMMMMII")
    assert line == "q1\t10\t0\t10\t+\tt1\t12\t0\t12\t9\t12\t60\tgi:f:0.900000\tcg:Z:5=1X4=2D"
    empty = host.format_paf("q", 5, "t", 7, 0, 0, 0, 0, True, 0, 0, b"")
    assert empty == "q\t5\t0\t0\t-\tt\t7\t0\t0\t0\t0\t60\tgi:f:0.000000\tcg:Z:"


def expected_paf(oracle, ids, seqs, i, j, scores, rev=False):
    q = seqs[i]
    if rev:
        comp = {65: 84, 84: 65, 67: 71, 71: 67, 97: 84, 116: 65, 99: 71, 103: 67}
        q = bytes(comp.get(b, 78) for b in reversed(q))
    pen, ops = oracle.Aligner(scores).align(q, seqs[j])
    m, x = ops.count(b"M"), ops.count(b"X")
    qe, te = m + x + ops.count(b"D"), m + x + ops.count(b"I")
    ident = (m / (m + x)) if (m + x) else 0.0
    return "%s\t%d\t0\t%d\t%s\t%s\t%d\t0\t%d\t%d\t%d\t60\tgi:f:%.6f\tcg:Z:%s" % (
        ids[i], len(seqs[i]), qe, "-" if rev else "+", ids[j], len(seqs[j]), te, m, max(qe, te), ident, rle(ops))


@pytest.mark.gpu
def test_all_pairs_paf_matches_oracle(host, oracle):
    """tests/integration_tests.rs:755-836: n(n-1) lines, each ordered pair once -- and every line
    equals alignment_to_paf of the oracle's alignment."""
    rng = random.Random(31)
    base = rand_seq(rng, 700)
    seqs = [base] + [mutate(base, d, rng) for d in (0.01, 0.05, 0.1)] + [rand_seq(rng, 90)]
    ids = ["seq%d" % i for i in range(len(seqs))]
    lines = host.all_pairs_paf(ids, seqs, "0,5,8,2,24,1", orientation="forward")
    n = len(seqs)
    assert len(lines) == n * (n - 1)
    it = iter(lines)
    for i in range(n):
        for j in range(n):
            if i != j:
                assert next(it) == expected_paf(oracle, ids, seqs, i, j, DEFAULT_2P)


@pytest.mark.gpu
def test_wfa_orientation(host, oracle):
    """tests/integration_tests.rs:443-555, 866-923: a reverse-complemented query is reported on '-'
    with coordinates in the reverse-complemented frame; forward wins ties."""
    rng = random.Random(8)
    ref = rand_seq(rng, 1200)
    fwd = mutate(ref, 0.03, rng)
    rcq = host.reverse_complement(mutate(ref, 0.03, rng))
    ids, seqs = ["ref", "fwd", "rev"], [ref, fwd, rcq]
    lines = host.all_pairs_paf(ids, seqs, "0,5,8,2,24,1", orientation="wfa")
    by = {(l.split("\t")[0], l.split("\t")[5]): l for l in lines}
    assert by[("fwd", "ref")].split("\t")[4] == "+"
    assert by[("rev", "ref")].split("\t")[4] == "-"
    assert by[("rev", "ref")] == expected_paf(oracle, ids, seqs, 2, 0, DEFAULT_2P, rev=True)
    assert by[("fwd", "ref")] == expected_paf(oracle, ids, seqs, 1, 0, DEFAULT_2P)
    ident = lambda l: float([f for f in l.split("\t") if f.startswith("gi:f:")][0][5:])
    assert abs(ident(by[("fwd", "ref")]) - ident(by[("rev", "ref")])) < 0.02


@pytest.mark.gpu
def test_wfa_align_sequences(host, oracle):
    """tests/integration_tests.rs:1143-1176: direct API in edit mode; counts use standard letters."""
    r = host.align_sequences(b"ACGTTACGT", b"ACGTACGT", (1, 1, 1, 0, 0), "edit")
    assert r["score"] == -2 and r["insertions"] == 1 and r["deletions"] == 0 and r["cigar"].count("I") == 1
    rng = random.Random(2)
    s = rand_seq(rng, 900)
    t = mutate(s, 0.08, rng)
    r = host.align_sequences(s, t, (5, 8, 2, 24, 1), "affine2p")
    pen, ops = oracle.Aligner(DEFAULT_2P).align(s, t)
    assert r["score"] == -pen and r["cigar"] == rle(ops)
    assert r["alignment_length"] == ops.count(b"M") + ops.count(b"X")


@pytest.mark.gpu
def test_mash_orientation_agrees_with_wfa(host, oracle):
    """tests/integration_tests.rs:866-1033: mash and WFA orientation agree with the ground truth on
    forward / reverse-complemented queries at several mutation rates (the CLI default is mash)."""
    rng = random.Random(12345)
    seqs, ids, truth = [], [], []
    for n, d, rev in ((1000, 0.0, False), (1000, 0.01, False), (1000, 0.01, True), (1000, 0.05, False),
                      (1000, 0.05, True), (100, 0.0, True), (5000, 0.001, False)):
        ref = rand_seq(rng, n)
        q = mutate(ref, d, rng)
        if rev:
            q = host.reverse_complement(q)
        ids += ["ref%d" % len(truth), "qry%d" % len(truth)]
        seqs += [ref, q]
        truth.append(rev)
    for orient in ("mash", "wfa"):
        lines = host.all_pairs_paf(ids, seqs, "0,5,8,2,24,1", orientation=orient)
        by = {(l.split("\t")[0], l.split("\t")[5]): l.split("\t")[4] for l in lines}
        for i, rev in enumerate(truth):
            assert by[("qry%d" % i, "ref%d" % i)] == ("-" if rev else "+"), (orient, i)
