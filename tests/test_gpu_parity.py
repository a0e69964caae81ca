"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, must equal the CPU oracle
bit for bit -- penalties and CIGAR op bytes -- on the same seeded inputs, on the committed golden
fixtures, and satisfy size-independent properties at BASELINE.json's full shapes."""
import json
import os
import random

import numpy as np
import pytest

from util import DEFAULT_2P, EDIT, PENALTY_SETS, mutate, rand_seq, random_pair, rle

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def check_against_oracle(engine, oracle, seqs, pairs, scores):
    engine.set_sequences(seqs)
    res, cigs = engine.align_pairs(scores, pairs)
    al = oracle.Aligner(scores)
    for i, p in enumerate(pairs):
        a, b = p[0], p[1]
        pen, ops = al.align(seqs[a], seqs[b])
        assert res["status"][i] == 0, (scores, i)
        assert res["penalty"][i] == pen and res["score"][i] == -pen, (scores, i, len(seqs[a]), len(seqs[b]))
        assert cigs[i] == ops, (scores, i, rle(cigs[i])[:60], rle(ops)[:60])
        c = {k: ops.count(k.encode()) for k in "MXID"}
        assert (res["num_matches"][i], res["num_mismatches"][i], res["num_ins"][i], res["num_del"][i]) == \
               (c["M"], c["X"], c["I"], c["D"])
        # parse_cigar_lengths of /root/reference/src/alignment.rs:320-344
        assert res["q_end"][i] == c["M"] + c["X"] + c["D"] == len(seqs[a])
        assert res["t_end"][i] == c["M"] + c["X"] + c["I"] == len(seqs[b])


def test_golden_fixtures(engine):
    with open(os.path.join(GOLDEN, "oracle_kats.json")) as f:
        doc = json.load(f)
    for case in doc["cases"]:
        r, ops = engine.align_one(tuple(case["scores"]), case["pattern"].encode("latin1"),
                                  case["text"].encode("latin1"))
        assert r["status"] == 0 and r["penalty"] == case["penalty"], case["name"]
        assert rle(ops) == case["cigar"], case["name"]


def test_reference_kat_counts(engine):
    """tests/integration_tests.rs:599-672 through the HIP path."""
    ref = b"ATCG" * 25
    q = bytearray(ref)
    q[10] = ord("G"); q[20] = ord("C"); del q[30]; q.insert(40, ord("A"))
    r, ops = engine.align_one(DEFAULT_2P, bytes(q), ref)
    assert (r["num_mismatches"], r["num_ins"], r["num_del"], r["penalty"]) == (2, 1, 1, 30)


@pytest.mark.parametrize("scores", PENALTY_SETS)
def test_random_pairs_bit_exact(engine, oracle, scores):
    rng = random.Random(1000 + (hash(scores) & 0xFFF))
    seqs, pairs = [], []
    for _ in range(150):
        s, t = random_pair(rng, 1500)
        seqs += [s, t]
        pairs.append((len(seqs) - 2, len(seqs) - 1))
    check_against_oracle(engine, oracle, seqs, pairs, scores)


def test_edge_cases(engine, oracle):
    """Empty and ragged inputs, single bases, the 100/101 length threshold (A.6), identical
    sequences (end reached at score 0), all-gap alignments, non-ACGT and lower-case bytes."""
    rng = random.Random(77)
    s1k = rand_seq(rng, 1000)
    seqs = [b"", b"A", b"C", b"ACGT", s1k, s1k, rand_seq(rng, 100), rand_seq(rng, 101), s1k[:100], s1k[:101],
            s1k.lower(), b"N" * 300, b"ACGTN" * 60, bytes(range(256)) * 2, rand_seq(rng, 5000), b"A" * 700, b"AC" * 350]
    n = len(seqs)
    pairs = [(i, j) for i in range(n) for j in range(n)]
    for scores in (DEFAULT_2P, EDIT):
        check_against_oracle(engine, oracle, seqs, pairs, scores)


def test_config1_all_pairs(engine, oracle):
    """BASELINE.json configs[0]: 8 x 1 kbp, 5% divergence, scores 0,1,1,1, -p none => 56 pairs."""
    from allwave_amd import synth
    data, offs, _ = synth.generate(8, 1000, 0.05, 1)
    seqs = [bytes(data[offs[i]:offs[i + 1]]) for i in range(8)]
    pairs = synth.all_pairs(8)
    assert len(pairs) == 56
    check_against_oracle(engine, oracle, seqs, [tuple(p) for p in pairs], EDIT)


def test_config2_sample_and_invariants(engine, oracle):
    """BASELINE.json configs[1] shape (256 x 10 kbp, 5%, default 2-piece scores): a seeded sample
    is compared bit-exact with the oracle; a larger slice is checked through properties that do
    not need the oracle: full consumption, M/X columns verified against the sequences, CIGAR
    re-scores to the reported penalty, and penalty(i,j) == penalty(j,i)."""
    from allwave_amd import synth
    data, offs, _ = synth.generate(256, 10000, 0.05, 2)
    allp = synth.all_pairs(256)
    engine.set_sequences((data, offs))
    sample = allp[::1571][:40]
    res, cigs = engine.align_pairs(DEFAULT_2P, sample)
    al = oracle.Aligner(DEFAULT_2P)
    for i, (a, b) in enumerate(sample):
        pen, ops = al.align(bytes(data[offs[a]:offs[a + 1]]), bytes(data[offs[b]:offs[b + 1]]))
        assert res["status"][i] == 0 and res["penalty"][i] == pen and cigs[i] == ops, (a, b)
    sl = np.concatenate([allp[:300], allp[:300][:, ::-1]])
    res, cigs = engine.align_pairs(DEFAULT_2P, sl)
    assert (res["status"] == 0).all()
    assert (res["penalty"][:300] == res["penalty"][300:]).all()  # symmetric penalties
    for i, (a, b) in enumerate(sl[:300]):
        p, t = bytes(data[offs[a]:offs[a + 1]]), bytes(data[offs[b]:offs[b + 1]])
        rc, rescored = oracle.cigar_check(cigs[i], p, t, DEFAULT_2P)  # validator only (wfa.rs:105-176)
        assert rc == 0 and rescored == res["penalty"][i]


def test_penalty_is_the_gotoh_optimum(engine, oracle):
    """The HIP path's penalty against the independent full-DP scorer (oracle/gotoh.c: Gotoh's recurrences, 2-piece), directly
    -- not through the BiWFA restatement: an exact WFA with no heuristic (alignment.rs:228) must return the optimum.  Random
    pairs up to 2 kbp of every shape random_pair() draws, all six penalty sets; the CIGAR must re-score to it (wfa.rs:105-176)."""
    for scores in PENALTY_SETS:
        rng = random.Random(2200 + len(scores) + scores[1])
        seqs, pairs = [], []
        for _ in range(40):
            s, t = random_pair(rng, maxlen=2000)
            seqs += [s, t]
            pairs.append((len(seqs) - 2, len(seqs) - 1))
        engine.set_sequences(seqs)
        res, cigs = engine.align_pairs(scores, pairs)
        for i, (a, b) in enumerate(pairs):
            want = oracle.gotoh_penalty(seqs[a], seqs[b], scores)
            assert res["status"][i] == 0 and res["penalty"][i] == want, (scores, i, len(seqs[a]), len(seqs[b]))
            rc, rescored = oracle.cigar_check(cigs[i], seqs[a], seqs[b], scores)
            assert rc == 0 and rescored == want, (scores, i)


def test_cell_steps_agree_with_the_oracles_count(oracle):
    """the roofline's unit, C(pair) = sum of (hi - lo + 1) over every compute-next call of the BiWFA tree, is
    "counted by the CPU oracle"; the bench line uses the kernel's own count (awv_stats.cell_steps).  On the same 96 config-2
    pairs the two must agree.  Against the oracle in the kernel's mode (known-optimum stop on): the step-by-step kernel
    computes one row ahead of the official score (+0.2 % measured); with passes, up to T - 1 rows past the end of phase 1
    (+0.05 %) and a search that met inside a far-apart pass is run again (each restart at most half a pair: +0.5 % of this
    sample).  Against the oracle's plain WFA2-order search the kernel counts 3-5 % FEWER: the known-optimum stop."""
    from allwave_amd import ffi, synth
    data, offs, _ = synth.generate(256, 10000, 0.05, 2)
    sample = synth.all_pairs(256)[::677][:96]
    _, ores, ost, _ = oracle.all_pairs(data, offs, sample, DEFAULT_2P, nthreads=8, fast_overlap=True)
    _, _, ost_plain, _ = oracle.all_pairs(data, offs, sample, DEFAULT_2P, nthreads=8, fast_overlap=False)
    for flags in (ffi.AWV_F_ONE_WAVE | ffi.AWV_F_SINGLE_STEP, ffi.AWV_F_ONE_WAVE):
        e = ffi.Engine(flags=flags)
        try:
            e.set_sequences((data, offs))
            res, _ = e.align_pairs(DEFAULT_2P, sample, want_cigars=False)
            st = e.stats()
        finally:
            e.close()
        assert (res["penalty"] == ores["penalty"]).all()
        rel = int(st.cell_steps) / int(ost.cell_steps) - 1.0
        assert 0.0 <= rel < 0.005 + 0.006 * int(st.restarts), (flags, st.cell_steps, ost.cell_steps, st.restarts)
        rel_plain = int(st.cell_steps) / int(ost_plain.cell_steps) - 1.0
        assert -0.07 < rel_plain < 0.0, (flags, st.cell_steps, ost_plain.cell_steps)


def test_reverse_complement_pairs(engine, oracle):
    """q_revcomp aligns reverse_complement(query) (alignment.rs:178-190): upper-cases, unknown -> N."""
    rng = random.Random(5)
    comp = {65: 84, 84: 65, 67: 71, 71: 67, 97: 84, 116: 65, 99: 71, 103: 67}

    def rc(s):
        return bytes(comp.get(b, 78) for b in reversed(s))

    t = rand_seq(rng, 1800)
    qs = [rc(mutate(t, 0.05, rng)), rc(mutate(t, 0.1, rng)).lower(), rc(mutate(t, 0.02, rng, b"ACGTN"))]
    seqs = qs + [t]
    engine.set_sequences(seqs)
    res, cigs = engine.align_pairs(DEFAULT_2P, [(i, 3, 1) for i in range(3)])
    al = oracle.Aligner(DEFAULT_2P)
    for i in range(3):
        pen, ops = al.align(rc(qs[i]), t)
        assert res["status"][i] == 0 and res["penalty"][i] == pen and cigs[i] == ops


def test_shard_invariance(engine):
    """K logical shards of the pair list give the same results as one call
    (the multi-GPU decomposition, exercised on one device)."""
    from allwave_amd import synth
    data, offs, _ = synth.generate(12, 2000, 0.05, 11)
    pairs = synth.all_pairs(12)
    engine.set_sequences((data, offs))
    res1, c1 = engine.align_pairs(DEFAULT_2P, pairs)
    for k in (2, 4, 8):
        got = {}
        for r in range(k):
            sub = pairs[r::k]
            rs, cs = engine.align_pairs(DEFAULT_2P, sub)
            for j, p in enumerate(sub):
                got[tuple(p)] = (int(rs["penalty"][j]), cs[j])
        for i, p in enumerate(pairs):
            assert got[tuple(p)] == (int(res1["penalty"][i]), c1[i])


def test_engine_stats_and_batching(engine):
    from allwave_amd import ffi, synth
    data, offs, _ = synth.generate(10, 1500, 0.05, 3)
    pairs = synth.all_pairs(10)
    e2 = ffi.Engine(max_batch_pairs=7)  # forces 13 launches
    try:
        e2.set_sequences((data, offs))
        r2, c2 = e2.align_pairs(DEFAULT_2P, pairs)
        st = e2.stats()
        assert st.launches == 13 and st.pairs_completed == 90 and st.cell_steps > 0
        assert st.aligned_bp == sum(int(offs[a + 1] - offs[a]) for a, _ in pairs)
    finally:
        e2.close()
    engine.set_sequences((data, offs))
    r1, c1 = engine.align_pairs(DEFAULT_2P, pairs)
    assert (r1["penalty"] == r2["penalty"]).all() and c1 == c2


def test_sink_error_stops_the_call():
    """A sink that returns non-zero ends the call with AWV_ERR_SINK and no later sink call (first error
    wins, iterator.rs:236-251) -- also when the failing sink ran on the engine's helper thread, which
    takes every batch but the last of a call of several batches."""
    import ctypes as C
    from allwave_amd import ffi, synth
    data, offs, _ = synth.generate(10, 800, 0.05, 9)
    pairs = synth.all_pairs(10)  # 90 pairs: 13 batches of at most 7
    p = np.zeros(len(pairs), dtype=ffi.PAIR_DTYPE)
    p["q_idx"], p["t_idx"] = pairs[:, 0], pairs[:, 1]
    pen = ffi.Penalties.from_scores(DEFAULT_2P)
    e2 = ffi.Engine(max_batch_pairs=7)
    try:
        e2.set_sequences((data, offs))
        for fail_at in (0, 1, 12):
            firsts = []

            def sink(user, first, n, rptr, arena):
                firsts.append(int(first))
                return 7 if len(firsts) - 1 == fail_at else 0

            cb = ffi.SINK_FN(sink)
            rc = ffi.load().awv_align_pairs(e2._h, C.byref(pen), p.ctypes.data, len(p), None, cb, None)
            assert rc == ffi.AWV_ERR_SINK, (fail_at, rc)
            assert firsts == [7 * i for i in range(fail_at + 1)], (fail_at, firsts)
        r, c = e2.align_pairs(DEFAULT_2P, pairs)  # the engine is usable afterwards
        assert (r["status"] == 0).all() and all(x is not None for x in c)
    finally:
        e2.close()


def test_bad_arguments(engine):
    from allwave_amd import ffi
    engine.set_sequences([b"ACGT", b"ACGA"])
    with pytest.raises(ffi.EngineError):
        engine.align_pairs((1, 5, 8, 2), [(0, 1)])  # match != 0
    with pytest.raises(ffi.EngineError):
        engine.align_pairs(DEFAULT_2P, [(0, 2)])     # index out of range
    res, _ = engine.align_pairs(DEFAULT_2P, np.zeros((0, 2), dtype=np.int32))
    assert len(res) == 0


def test_row_width_and_sequence_paths_agree(oracle):
    """The kernel variants must be interchangeable: one wave vs four waves per pair, multi-step passes
    (chained sweeps / single sweeps) vs the step-by-step path, 16-bit vs forced 32-bit wavefront rows,
    2-bit packed LDS staging vs raw-byte probes from HBM -- all bit-exact against the oracle."""
    from allwave_amd import ffi
    rng = random.Random(4242)
    seqs, pairs = [], []
    for _ in range(60):
        s, t = random_pair(rng, 2500)
        seqs += [s, t]
        pairs.append((len(seqs) - 2, len(seqs) - 1))
    for flags in (ffi.AWV_F_ONE_WAVE, ffi.AWV_F_FOUR_WAVES, ffi.AWV_F_ONE_WAVE | ffi.AWV_F_SINGLE_STEP, ffi.AWV_F_ONE_WAVE | ffi.AWV_F_NO_CHAIN,
                  ffi.AWV_F_FOUR_WAVES | ffi.AWV_F_SINGLE_STEP, ffi.AWV_F_ONE_WAVE | ffi.AWV_F_FORCE_INT32,
                  ffi.AWV_F_FOUR_WAVES | ffi.AWV_F_FORCE_INT32, ffi.AWV_F_ONE_WAVE | ffi.AWV_F_NO_PACKED_SEQ,
                  ffi.AWV_F_FOUR_WAVES | ffi.AWV_F_NO_PACKED_SEQ, ffi.AWV_F_ONE_WAVE | ffi.AWV_F_FORCE_INT32 | ffi.AWV_F_NO_PACKED_SEQ):
        e = ffi.Engine(flags=flags)
        try:
            for scores in (DEFAULT_2P, (0, 4, 6, 2)):
                check_against_oracle(e, oracle, seqs, pairs, scores)
        finally:
            e.close()


@pytest.mark.parametrize("scores", [DEFAULT_2P, (0, 7, 12, 2, 36, 1), (0, 4, 6, 2, 18, 1), (0, 3, 4, 1), (0, 4, 6, 2), (0, 5, 8, 2, 12, 1)])
def test_multi_step_passes_all_presets(oracle, scores):
    """The far-apart phase in multi-step passes under every penalty shape it is instantiated for -- the
    default (chained sweeps: x = 5, o1+e1 = 10), the CLI's other ANI presets (main.rs:83-124: single sweeps of
    5 / 4 / 3 scores), gap-affine with e = 2, and a 2-piece set whose second gap opens early (o2+e2 = 13:
    chains of two) -- on pairs long enough (3-12 kbp, 3-12 %) for the phase to run, including unequal
    lengths; stats confirm the passes ran, results equal the oracle's and the step-by-step kernel's."""
    from allwave_amd import ffi
    rng = random.Random(hash(scores) & 0xFFFF)
    seqs, pairs = [], []
    for n, d in ((3000, 0.05), (6000, 0.03), (12000, 0.08), (5000, 0.12), (9000, 0.04)):
        a = rand_seq(rng, n)
        b = mutate(a, d, rng)
        seqs += [a, b, b[: n - n // 7]]
        k = len(seqs) - 3
        pairs += [(k, k + 1), (k + 1, k), (k + 2, k), (k, k + 2)]
    e = ffi.Engine(flags=ffi.AWV_F_ONE_WAVE)
    try:
        check_against_oracle(e, oracle, seqs, pairs, scores)
        st = e.stats()
        assert st.multi_cell_steps > 0.3 * st.cell_steps, (st.multi_cell_steps, st.cell_steps)
        assert st.deep_cell_steps > 0, "the margin zone ran in passes that store every I/D row (deep_phase)"
        res_multi, cig_multi = e.align_pairs(scores, pairs)
    finally:
        e.close()
    # the round-2 path: far-apart passes only, the margin zone step by step
    e = ffi.Engine(flags=ffi.AWV_F_ONE_WAVE | ffi.AWV_F_NO_DEEP)
    try:
        e.set_sequences(seqs)
        res_nodeep, cig_nodeep = e.align_pairs(scores, pairs)
        st = e.stats()
        assert st.deep_cell_steps == 0 and st.multi_cell_steps > 0.2 * st.cell_steps
    finally:
        e.close()
    assert (res_multi["penalty"] == res_nodeep["penalty"]).all() and cig_multi == cig_nodeep
    e = ffi.Engine(flags=ffi.AWV_F_ONE_WAVE | ffi.AWV_F_SINGLE_STEP)
    try:
        e.set_sequences(seqs)
        res_single, cig_single = e.align_pairs(scores, pairs)
        assert e.stats().multi_cell_steps == 0
    finally:
        e.close()
    assert (res_multi["penalty"] == res_single["penalty"]).all() and cig_multi == cig_single
    # the same passes on 32-bit rows (five-score sweeps whose sources are loaded one step ahead; chains of two under the
    # default scores), one wave and four waves per pair
    for flags in (ffi.AWV_F_ONE_WAVE | ffi.AWV_F_FORCE_INT32, ffi.AWV_F_FOUR_WAVES | ffi.AWV_F_FORCE_INT32):
        e = ffi.Engine(flags=flags)
        try:
            e.set_sequences(seqs)
            res32, cig32 = e.align_pairs(scores, pairs)
            st = e.stats()
            assert st.multi_cell_steps > 0.3 * st.cell_steps
        finally:
            e.close()
        assert (res32["status"] == 0).all() and (res32["penalty"] == res_multi["penalty"]).all() and cig32 == cig_multi, flags


def test_long_sequences_use_32bit_rows(engine, oracle):
    """Lengths >= 32760 select the int32 kernel and sequences too long for LDS staging at the top
    levels (BASELINE config 4's regime, scaled down): 40 kbp at 2%, plus a length-mismatched pair."""
    rng = random.Random(99)
    a = rand_seq(rng, 40000)
    b = mutate(a, 0.02, rng)
    c = a[:9000]
    seqs = [a, b, c]
    check_against_oracle(engine, oracle, seqs, [(0, 1), (1, 0), (2, 1), (0, 2)], DEFAULT_2P)
    st = engine.stats()
    assert st.n_breakpoints > 10 and st.pairs_completed == 4


def test_unstaged_long_pairs_chain_through_lds(oracle):
    """70 kbp at 2 %: 32-bit rows, four waves per pair, and a top BiWFA level too long for the LDS staging of the packed sequences
    (140 k bases against the 32 KB region) -- the regime in which the far-apart passes chain three sweeps with the middle
    sweep's rows kept in that region (AWV_LDS_CHAIN, biwfa_device.hpp).  Against the oracle, and against the same engine with
    chaining off (AWV_F_NO_CHAIN)."""
    from allwave_amd import ffi
    rng = random.Random(1234)
    a = rand_seq(rng, 70000)
    b = mutate(a, 0.02, rng)
    c = mutate(a, 0.03, rng)
    seqs = [a, b, c]
    pairs = [(0, 1), (1, 2), (2, 0)]
    out = []
    for flags in (0, ffi.AWV_F_NO_CHAIN):
        e = ffi.Engine(flags=flags)
        try:
            check_against_oracle(e, oracle, seqs, pairs, DEFAULT_2P)
            st = e.stats()
            assert st.multi_cell_steps > 0.8 * st.cell_steps
            res, cigs = e.align_pairs(DEFAULT_2P, pairs)
            out.append((res["penalty"].tolist(), cigs))
        finally:
            e.close()
    assert out[0] == out[1]


def test_packed_probes_of_unstaged_sub_problems(oracle):
    """Sub-problems too long for the LDS staging probe the 2-bit words where they lie in HBM (seq_mode 2, biwfa_device.hpp):
    forward and reverse probes, sub-problems that begin inside a word and inside the sequence (the second BiWFA level of a
    150 kbp pair is still unstaged), a sequence's very first and last bases (the first sequence of the set: the pad words in
    front of the packed array), reverse-complemented queries, one wave per pair (a 6 KB staging region: 20 kbp pairs are
    unstaged there) and four, 16- and 32-bit rows -- against the oracle and against the raw-byte probes (AWV_F_NO_PACKED_SEQ).
    A pair with a non-ACGT base keeps the raw bytes.  The 150 kbp pair is also the case of a launch with 32-bit rows whose
    sub-problems below 32,760 bases are searched with 16-bit rows (AWV_SUB16); AWV_F_FORCE_INT32 pins 32-bit rows throughout."""
    from allwave_amd import ffi
    rng = random.Random(777)
    comp = {65: 84, 84: 65, 67: 71, 71: 67}

    def rc(s):
        return bytes(comp.get(b, 78) for b in reversed(s))

    a = rand_seq(rng, 150000)
    b = mutate(a, 0.015, rng)
    c = rand_seq(rng, 20000)
    d = mutate(c, 0.06, rng)
    f = mutate(c, 0.04, rng)
    n = bytearray(mutate(c, 0.03, rng))
    n[7000] = ord("N")
    seqs = [a, b, c, d, rc(f), bytes(n)]
    long_pairs = [(0, 1, 0), (1, 0, 0)]
    short_pairs = [(2, 3, 0), (3, 2, 0), (4, 2, 1), (4, 3, 1), (5, 2, 0), (2, 5, 0)]
    al = oracle.Aligner(DEFAULT_2P)
    plain = {4: f}
    want = {}
    for p in long_pairs + short_pairs:
        want[p] = al.align(plain.get(p[0], seqs[p[0]]), seqs[p[1]])
    for flags, pairs in ((0, long_pairs + short_pairs), (ffi.AWV_F_ONE_WAVE, short_pairs), (ffi.AWV_F_ONE_WAVE | ffi.AWV_F_FORCE_INT32, short_pairs),
                         (ffi.AWV_F_FOUR_WAVES | ffi.AWV_F_NO_PACKED_SEQ, long_pairs[:1] + short_pairs),
                         (ffi.AWV_F_FORCE_INT32, long_pairs[:1])):  # (32-bit rows throughout: no 16-bit searches of the short sub-problems)
        e = ffi.Engine(flags=flags)
        try:
            e.set_sequences(seqs)
            res, cigs = e.align_pairs(DEFAULT_2P, pairs)
            for i, p in enumerate(pairs):
                assert res["status"][i] == 0, (flags, p)
                assert (res["penalty"][i], cigs[i]) == want[p], (flags, p, res["penalty"][i], want[p][0])
        finally:
            e.close()


def test_narrow_first_attempt_is_rerun_with_wider_rows(oracle):
    """Long sequences start with rows narrower than plen + tlen; pairs whose wavefronts outgrow them
    come back CAPACITY from the first launch and are re-run wider.  Forced here on 6 kbp pairs by
    capping the first attempt at 2048 columns: several launches, identical results."""
    from allwave_amd import ffi
    rng = random.Random(4242)
    a = rand_seq(rng, 6000)
    seqs = [a, mutate(a, 0.08, rng), mutate(a, 0.01, rng), rand_seq(rng, 1500)]
    pairs = [(i, j) for i in range(4) for j in range(4) if i != j]
    e = ffi.Engine(first_row_cols=2048)
    try:
        check_against_oracle(e, oracle, seqs, pairs, DEFAULT_2P)
        st = e.stats()
        assert st.launches >= 2 and st.pairs_completed == len(pairs)
    finally:
        e.close()


@pytest.mark.parametrize("scores", [(0, 6, 10, 3, 70, 2), (0, 80, 5, 2), (0, 3, 90, 1), (0, 9, 3, 7, 30, 5)])
def test_wide_scope_penalties(engine, oracle, scores):
    """Penalty sets whose lookback (scope = max(x, o1+e1, o2+e2) + 1) exceeds one wave's 64 lanes or
    the LDS staging budget: deeper rings, the row filter of the overlap search in several blocks."""
    rng = random.Random(hash(scores) & 0xFFFF)
    seqs, pairs = [], []
    for _ in range(14):
        s, t = random_pair(rng, 3500)
        seqs += [s, t]
        pairs.append((len(seqs) - 2, len(seqs) - 1))
    check_against_oracle(engine, oracle, seqs, pairs, scores)


def test_config3_shaped_batches():
    """BASELINE configs[2] (config 3: 4096 x 10 kbp, 5 %), a 40 k-pair random slice of its all-pairs list,
    CIGAR arena capped so that the call takes several launches: every pair completes, the CIGAR
    consumes both sequences, op counts add up, and penalty(a, b) == penalty(b, a); the first 256
    pairs of the slice are compared with the oracle bit for bit."""
    from allwave_amd import ffi, synth
    cfg = synth.CONFIGS["c3"]
    data, offs, _ = synth.generate(cfg["nseq"], cfg["length"], cfg["d"], cfg["seed"])
    n = 40000
    rng = np.random.default_rng(3)
    qi = rng.integers(0, cfg["nseq"], n)
    ti = (qi + 1 + rng.integers(0, cfg["nseq"] - 1, n)) % cfg["nseq"]
    pairs = np.stack([qi, ti], axis=1).astype(np.int32)
    pairs[1::2] = pairs[0::2][:, ::-1]  # odd entries: the swapped even pair
    e = ffi.Engine(flags=ffi.AWV_F_KEEP_ON_DEVICE, max_arena_bytes=256 << 20)
    try:
        e.set_sequences((data, offs))
        res, _ = e.align_pairs(cfg["scores"], pairs, want_cigars=False)
        st = e.stats()
        assert st.launches >= 3 and st.pairs_completed == n
        assert (res["status"] == 0).all()
        ql = (offs[pairs[:, 0] + 1] - offs[pairs[:, 0]]).astype(np.int64)
        tl = (offs[pairs[:, 1] + 1] - offs[pairs[:, 1]]).astype(np.int64)
        assert (res["q_end"] == ql).all() and (res["t_end"] == tl).all()
        assert (res["num_matches"] + res["num_mismatches"] + res["num_ins"] + res["num_del"] == res["cigar_len"]).all()
        assert (res["penalty"][0::2] == res["penalty"][1::2]).all()
        # ... and 256 pairs of config 3's own read set against the oracle: penalty, op counts and the
        # FNV-1a of the op bytes, pair by pair (the oracle's thread-pool driver, allpairs_cpu.c)
        from oracle import oracle as O
        sub = np.ascontiguousarray(pairs[:256])
        e2 = ffi.Engine()
        try:
            e2.set_sequences((data, offs))
            gres, gcigs = e2.align_pairs(cfg["scores"], sub)
        finally:
            e2.close()
        _, ores, _, _ = O.all_pairs(data, offs, sub, cfg["scores"], nthreads=min(16, os.cpu_count() or 1))
        assert (gres["status"] == 0).all() and (ores["status"] == 0).all()
        assert (gres["penalty"] == ores["penalty"]).all()
        assert (gres["penalty"] == res["penalty"][:256]).all()  # the capped multi-launch run gave the same answers
        assert (gres["cigar_len"] == ores["cigar_len"].astype(np.uint32)).all()
        for k_g, k_o in (("num_matches", "num_matches"), ("num_mismatches", "num_mismatches"),
                         ("num_ins", "num_ins_text"), ("num_del", "num_del_pattern")):
            assert (gres[k_g] == ores[k_o]).all(), k_g
        for i in range(len(sub)):
            assert O.fnv1a(gcigs[i]) == int(ores["cigar_hash"][i]), ("config 3 pair", tuple(sub[i]))
    finally:
        e.close()


def test_config4_shaped_budgeted_rows(oracle):
    """BASELINE config 4's regime: 100 kbp at 2 % (32-bit rows, sequences too long to stage in LDS at
    the top levels) with the scratch budget set so low that full-width rows (plen + tlen columns) do
    not fit: the first launch runs with budgeted rows, anything that outgrows them is re-run wider."""
    from allwave_amd import ffi, synth
    data, offs, _ = synth.generate(3, 100000, 0.02, 4)
    seqs = [bytes(data[offs[i]:offs[i + 1]]) for i in range(3)]
    e = ffi.Engine(max_scratch_bytes=1 << 30)
    try:
        check_against_oracle(e, oracle, seqs, [(0, 1), (1, 2), (2, 0)], DEFAULT_2P)
        assert e.stats().pairs_completed == 3
    finally:
        e.close()


def test_config5_shaped_sparsified_mixed_lengths(engine, oracle):
    """BASELINE config 5's shape, scaled down: prefixes of 1-9 kbp of a common root at 10 %, the pair
    list from the host planner's tree sparsifier (-p tree:3:1:0.1); bit-exact against the oracle."""
    from allwave_amd import host as H, synth
    data, offs, ids = synth.generate(20, 9000, 0.10, 5, mixed_lengths=(1000, 9000))
    seqs = [bytes(data[offs[i]:offs[i + 1]]) for i in range(20)]
    pairs = [tuple(int(v) for v in p) for p in H.plan_pairs(ids, seqs, "tree:3:1:0.1")]
    assert 20 <= len(pairs) < 20 * 19
    check_against_oracle(engine, oracle, seqs, pairs, DEFAULT_2P)


def test_config5_real_lengths(oracle):
    """BASELINE configs[4] (config 5) at its REAL lengths: 512 prefixes of 1-50 kbp of a common root at
    10 %, pair list from the planner's `-p tree:3:1:0.1` (iterator.rs:30-50 -> knn_graph.rs); eight
    pairs of that list -- the shortest query against the longest target it meets, the longest against
    the shortest, and a spread in between: forced gaps of tens of kbp, 16- and 32-bit rows and the
    wide flavours in one call -- against the oracle bit for bit."""
    from allwave_amd import ffi, host as H, synth
    cfg = synth.CONFIGS["c5"]
    data, offs, ids = synth.generate(cfg["nseq"], cfg["length"], cfg["d"], cfg["seed"], mixed_lengths=cfg["mixed_lengths"])
    lens = (offs[1:] - offs[:-1]).astype(np.int64)
    assert lens.min() < 1500 and lens.max() > 48000
    seqs = [bytes(data[offs[i]:offs[i + 1]]) for i in range(cfg["nseq"])]
    plist = np.asarray(H.plan_pairs(ids, seqs, cfg["sparsify"]), dtype=np.int32)[:, :2]
    assert 10000 < len(plist) < 60000
    dl = lens[plist[:, 0]] - lens[plist[:, 1]]
    order = np.argsort(dl, kind="stable")
    pick = [order[0], order[-1]] + [order[int(f * (len(order) - 1))] for f in (0.02, 0.25, 0.5, 0.75, 0.98)]
    pick.append(int(np.argmin(lens[plist[:, 0]] + lens[plist[:, 1]])))
    sub = np.ascontiguousarray(plist[pick])
    assert abs(int(dl[order[0]])) > 30000 and abs(int(dl[order[-1]])) > 30000
    e = ffi.Engine()
    try:
        e.set_sequences((data, offs))
        gres, gcigs = e.align_pairs(cfg["scores"], sub)
        st = e.stats()
        assert st.launches >= 2  # more than one kernel flavour / row width in the call
    finally:
        e.close()
    _, ores, _, _ = oracle.all_pairs(data, offs, sub, cfg["scores"], nthreads=min(8, os.cpu_count() or 1), fast_overlap=True)
    assert (gres["status"] == 0).all() and (ores["status"] == 0).all()
    assert (gres["penalty"] == ores["penalty"]).all(), (gres["penalty"], ores["penalty"])
    assert (gres["q_end"] == lens[sub[:, 0]]).all() and (gres["t_end"] == lens[sub[:, 1]]).all()
    for i in range(len(sub)):
        assert oracle.fnv1a(gcigs[i]) == int(ores["cigar_hash"][i]), ("config 5 pair", tuple(sub[i]), int(lens[sub[i, 0]]), int(lens[sub[i, 1]]))


@pytest.mark.parametrize("scores", [DEFAULT_2P, (0, 4, 6, 2), (0, 3, 5, 1, 20, 1), EDIT])
def test_wide16_rows(oracle, scores):
    """Pairs whose LONGER sequence has 32760 bases or more while the shorter one fits 16 bits run on 16-bit rows that
    hold min(h, v) per cell (AWV_WIDE16: 32-bit row metadata, insertions / deletions add 1 on one side of the main
    diagonal only).  Forced gaps of tens of kbp in both roles (short text, short pattern), a related and an unrelated
    short sequence, the boundary length 32759 and reverse-complemented queries: against the oracle bit for bit, and
    field by field against the 32-bit-row kernels on the same pairs."""
    from allwave_amd import ffi
    rng = random.Random(32760 + len(scores))
    comp = {65: 84, 84: 65, 67: 71, 71: 67}
    root = rand_seq(rng, 41000)
    long_a = root[:40000]
    long_b = mutate(root, 0.03, rng)[:36000]
    short_rel = mutate(root[18000:21000], 0.05, rng)      # 3 kbp from the middle: forced gaps on both sides
    short_unrel = rand_seq(rng, 2000)
    edge = mutate(root, 0.02, rng)[:32759]                # the longest sequence 16-bit values can hold
    seqs = [long_a, long_b, short_rel, short_unrel, edge, bytes(comp[b] for b in reversed(short_rel))]
    pairs = [(0, 2, 0), (2, 0, 0), (1, 3, 0), (3, 1, 0), (0, 4, 0), (4, 1, 0), (5, 0, 1), (0, 5, 0)]
    out = {}
    for name, flags in (("wide16", 0), ("rows32", ffi.AWV_F_NO_WIDE16)):
        e = ffi.Engine(flags=flags)
        try:
            e.set_sequences(seqs)
            out[name] = e.align_pairs(scores, pairs)
        finally:
            e.close()
    (res, cigs), (res32, cigs32) = out["wide16"], out["rows32"]
    assert (res["status"] == 0).all() and (res32["status"] == 0).all()
    for f in res.dtype.names:
        assert (res[f] == res32[f]).all(), f
    assert cigs == cigs32
    # the oracle's thread-pool driver on the same pairs (the reverse-complemented query is sequence 2 itself)
    data = np.frombuffer(b"".join(seqs), dtype=np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(x) for x in seqs])]).astype(np.uint64)
    opairs = np.asarray([(2 if rc else a, b) for a, b, rc in pairs], dtype=np.int32)
    _, ores, _, _ = oracle.all_pairs(data, offs, opairs, scores, nthreads=min(8, os.cpu_count() or 1))
    assert (ores["status"] == 0).all()
    for i in range(len(pairs)):
        assert res["penalty"][i] == ores["penalty"][i] and oracle.fnv1a(cigs[i]) == int(ores["cigar_hash"][i]), (scores, pairs[i])


@pytest.mark.parametrize("length", [32759, 32760])
def test_row_width_boundary(engine, oracle, length):
    """The longest sequences that still use 16-bit rows (32759) and the shortest that take 32-bit
    rows (32760): offsets, NULL encoding and row metadata at the edge of the 16-bit range."""
    rng = random.Random(length)
    a = rand_seq(rng, length)
    b = mutate(a, 0.03, rng)[:length]
    b = b + rand_seq(rng, length - len(b))
    check_against_oracle(engine, oracle, [a, b], [(0, 1), (1, 0)], DEFAULT_2P)


def test_very_unequal_lengths(oracle):
    """A length difference forces a gap that long: |dlen| >= 4096 routes a pair to the four-wave
    flavour, >= 16384 to the sixteen-wave one (one pair per CU); both bit-exact, and identical to
    the one-wave kernel's answer."""
    from allwave_amd import ffi
    rng = random.Random(31337)
    a = rand_seq(rng, 19000)
    b = mutate(a, 0.04, rng)[7000:8500]      # 1.5 kbp infix: dlen = 17.5 k
    c = mutate(a, 0.04, rng)[:13000]         # 13 kbp prefix: dlen = 6 k
    seqs = [a, b, c]
    pairs = [(1, 0), (0, 1), (2, 0), (0, 2)]
    for flags in (0, ffi.AWV_F_ONE_WAVE):
        e = ffi.Engine(flags=flags)
        try:
            check_against_oracle(e, oracle, seqs, pairs, DEFAULT_2P)
        finally:
            e.close()
