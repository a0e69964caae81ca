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


def test_with_sparsification_replans_the_pair_list(host):
    """iterator.rs:101-110: with_sparsification plans the pair list again through with_options -- the result is the list a
    direct with_options(strategy) gives, for every strategy (CPU only: planning does not touch the GPU)."""
    rng = random.Random(5)
    base = rand_seq(rng, 600)
    seqs = [mutate(base, 0.03 * (i % 5), rng) for i in range(14)]
    ids = ["s%02d" % i for i in range(len(seqs))]
    for strat in ("none", "random:0.4", "giant:0.9", "auto", "tree:2:1:0.1"):
        direct = host.plan_pairs(ids, seqs, strat)
        assert host.plan_pairs(ids, seqs, strat, resparsify=True) == direct, strat
        if strat == "none":
            assert len(direct) == len(seqs) * (len(seqs) - 1)


@pytest.mark.gpu
def test_every_consumer_of_the_pair_list(host, oracle):
    """The reference offers five ways to consume the alignments of a pair list: for_each_with_callback (iterator.rs:127-137),
    the sequential Iterator (:151-171), into_par_iter() with a callback on several threads (:113-125, :206-253) or collected
    (:182-203), and process_alignments_with_callback (lib.rs:57-68).  All of them must yield the same records -- the
    oracle's -- in pair-list order where the reference's order is defined (everything but the threaded callback)."""
    rng = random.Random(77)
    base = rand_seq(rng, 500)
    seqs = [base] + [mutate(base, d, rng) for d in (0.02, 0.06, 0.12)] + [rand_seq(rng, 120), b"ACGT" * 30]
    ids = ["q%d" % i for i in range(len(seqs))]
    sc = "0,5,8,2,24,1"
    n = len(seqs)
    want = [expected_paf(oracle, ids, seqs, i, j, DEFAULT_2P) for i in range(n) for j in range(n) if i != j]
    assert host.iterate(ids, seqs, sc, mode="for_each") == want
    assert host.iterate(ids, seqs, sc, mode="next", chunk=7) == want          # several engine calls (30 pairs in chunks of 7)
    assert host.iterate(ids, seqs, sc, mode="next") == want                    # one engine call
    assert host.iterate(ids, seqs, sc, mode="par_collect") == want
    assert sorted(host.iterate(ids, seqs, sc, mode="par_for_each", threads=4)) == sorted(want)
    # process_alignments_with_callback = with_options(.., exclude_self, mash orientation, strategy): all reads are '+' here
    assert host.iterate(ids, seqs, sc, mode="process_alignments", orientation="mash") == \
        host.all_pairs_paf(ids, seqs, sc, orientation="mash")
    # a sparsified list through with_sparsification: the planned pairs, in plan order, each record the oracle's
    for strat in ("random:0.5", "tree:1:1:0.2"):
        plan = host.plan_pairs(ids, seqs, strat)
        got = host.iterate(ids, seqs, sc, mode="next", chunk=4, sparsification=strat, resparsify=True)
        assert got == [expected_paf(oracle, ids, seqs, i, j, DEFAULT_2P) for i, j in plan], strat


@pytest.mark.gpu
def test_callback_error_stops_every_consumer(host):
    """iterator.rs:220-251: the first error a callback returns wins, ends the run and is what the caller sees."""
    rng = random.Random(3)
    seqs = [rand_seq(rng, 200) for _ in range(5)]
    ids = ["e%d" % i for i in range(5)]
    for mode, kw in (("for_each", {}), ("par_for_each", {"threads": 3}), ("process_alignments", {"orientation": "mash"})):
        with pytest.raises(host.HostError, match="callback failed at record 6"):
            host.iterate(ids, seqs, "0,1,1,1", mode=mode, fail_at=6, **kw)


@pytest.mark.gpu
def test_failed_pair_is_an_empty_record(host, oracle):
    """alignment.rs:49-64 + lib.rs:95-111: a pair whose alignment fails is still a record -- score i32::MAX, no CIGAR, all
    coordinates zero -- and still a PAF line.  Forced here the only way this engine fails a pair: rows capped at 2048 columns
    (first_row_cols) and no wider re-run (AWV_F_NO_RERUN), so the divergent 6 kbp pairs stay AWV_ST_CAPACITY while the
    similar ones complete."""
    from allwave_amd import ffi
    rng = random.Random(4242)
    a = rand_seq(rng, 6000)
    seqs = [a, mutate(a, 0.08, rng), mutate(a, 0.01, rng)]
    ids = ["a", "far", "near"]
    host.set_engine_config(flags=ffi.AWV_F_NO_RERUN, first_row_cols=2048)
    try:
        lines = host.all_pairs_paf(ids, seqs, "0,5,8,2,24,1", orientation="forward")
    finally:
        host.set_engine_config()  # back to the defaults: the next test gets a fresh engine
    by = {(l.split("\t")[0], l.split("\t")[5]): l for l in lines}
    assert len(lines) == 6
    for q, t in (("a", "far"), ("far", "a"), ("far", "near"), ("near", "far")):
        qi, ti = ids.index(q), ids.index(t)
        assert by[(q, t)] == "%s\t%d\t0\t0\t+\t%s\t%d\t0\t0\t0\t0\t60\tgi:f:0.000000\tcg:Z:" % (q, len(seqs[qi]), t, len(seqs[ti])), (q, t)
    assert by[("a", "near")] == expected_paf(oracle, ids, seqs, 0, 2, DEFAULT_2P)
    assert by[("near", "a")] == expected_paf(oracle, ids, seqs, 2, 0, DEFAULT_2P)
    # the same pairs complete when the re-run is allowed
    host.set_engine_config(first_row_cols=2048)
    try:
        lines = host.all_pairs_paf(ids, seqs, "0,5,8,2,24,1", orientation="forward")
    finally:
        host.set_engine_config()
    assert lines[0] == expected_paf(oracle, ids, seqs, 0, 1, DEFAULT_2P)
