"""Host-side pair planning (csrc/host/planner.cpp): SipHash-1-3 as Rust's DefaultHasher, mash
sketches / orientation, the hash sparsifiers and kNN/tree pairs, and the CLI driver.  Expected values
restate the reference's unit tests (src/knn_graph.rs:195-387, src/mash.rs:186-260) and the formulae
of src/iterator.rs:256-334."""
import math
import os
import random
import subprocess

import numpy as np
import pytest

from util import mutate, rand_seq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host(hip_lib):
    from allwave_amd import build, host as H
    build.build_host()
    H.load()
    return H


def test_siphash_core_matches_reference_vector(host):
    """SipHash-2-4 test vector of the SipHash paper (key 00..0f, message 00..0e): validates the
    round function and finalisation the 1-3 variant (Rust's DefaultHasher) shares."""
    k0 = int.from_bytes(bytes(range(8)), "little")
    k1 = int.from_bytes(bytes(range(8, 16)), "little")
    assert host.siphash(bytes(range(15)), k0, k1, 2, 4) == 0xA129CA6149BE45E5
    assert host.siphash(b"", k0, k1, 2, 4) == 0x726FDB47DD0E0E31
    # DefaultHasher framing: [u8] = u64 length prefix + bytes, str = bytes + 0xFF
    assert host.hash_bytes(b"ACGT") == host.siphash((4).to_bytes(8, "little") + b"ACGT")
    assert host.hash_str("a:b") == host.siphash(b"a:b\xff")
    assert host.hash_str("a:b") != host.hash_str("b:a")


def _siphash_py(key, msg, c, d):
    """SipHash-c-d written out from the paper (Aumasson & Bernstein 2012, section 2) -- an implementation
    independent of csrc/host/planner.cpp, pinned below by published vectors."""
    import struct
    M = (1 << 64) - 1
    rotl = lambda x, b: ((x << b) | (x >> (64 - b))) & M
    k0, k1 = struct.unpack("<QQ", key)
    v = [k0 ^ 0x736f6d6570736575, k1 ^ 0x646f72616e646f6d, k0 ^ 0x6c7967656e657261, k1 ^ 0x7465646279746573]

    def rnd():
        v[0] = (v[0] + v[1]) & M; v[1] = rotl(v[1], 13); v[1] ^= v[0]; v[0] = rotl(v[0], 32)
        v[2] = (v[2] + v[3]) & M; v[3] = rotl(v[3], 16); v[3] ^= v[2]
        v[0] = (v[0] + v[3]) & M; v[3] = rotl(v[3], 21); v[3] ^= v[0]
        v[2] = (v[2] + v[1]) & M; v[1] = rotl(v[1], 17); v[1] ^= v[2]; v[2] = rotl(v[2], 32)

    n = len(msg)
    b = (n & 0xFF) << 56
    for i in range(0, n - n % 8, 8):
        m = struct.unpack_from("<Q", msg, i)[0]
        v[3] ^= m
        for _ in range(c):
            rnd()
        v[0] ^= m
    for i, x in enumerate(msg[n - n % 8:]):
        b |= x << (8 * i)
    v[3] ^= b
    for _ in range(c):
        rnd()
    v[0] ^= b
    v[2] ^= 0xFF
    for _ in range(d):
        rnd()
    return v[0] ^ v[1] ^ v[2] ^ v[3]


def test_siphash_known_answers(host):
    """Known-answer vectors in the reference implementation's format (key 00..0f, message 00..n-1, output
    as little-endian bytes): the first five of SipHash-2-4's published table (vectors.h `vectors_sip64`),
    the paper's Appendix A value, and the first vector of the Rust standard library's own SipHash-1-3 test
    (library/core/tests/hash/sip.rs `test_siphash_1_3`).  Then planner::siphash against the independent
    implementation above on all 64 message lengths of that format for both variants, and the two
    DefaultHasher framings the reference relies on (iterator.rs:261-281 `str`, alignment.rs:142-149 /
    mash.rs:109-113 `[u8]`) on top of the zero-key 1-3 variant."""
    key = bytes(range(16))
    k0, k1 = int.from_bytes(key[:8], "little"), int.from_bytes(key[8:], "little")
    published_24 = ["310e0edd47db6f72", "fd67dc93c539f874", "5a4fa9d909806c0d", "2d7efbd796666785", "b7877127e09427cf"]
    for n, hexv in enumerate(published_24):
        assert host.siphash(bytes(range(n)), k0, k1, 2, 4).to_bytes(8, "little").hex() == hexv
        assert _siphash_py(key, bytes(range(n)), 2, 4).to_bytes(8, "little").hex() == hexv
    assert host.siphash(bytes(range(15)), k0, k1, 2, 4) == 0xA129CA6149BE45E5
    assert host.siphash(b"", k0, k1, 1, 3).to_bytes(8, "little").hex() == "dcc40f055801acab"  # Rust's test_siphash_1_3, vector 0
    assert _siphash_py(key, b"", 1, 3).to_bytes(8, "little").hex() == "dcc40f055801acab"
    for n in range(64):
        msg = bytes(range(n))
        assert host.siphash(msg, k0, k1, 1, 3) == _siphash_py(key, msg, 1, 3), n
        assert host.siphash(msg, k0, k1, 2, 4) == _siphash_py(key, msg, 2, 4), n
        assert host.siphash(msg, 0, 0, 1, 3) == _siphash_py(bytes(16), msg, 1, 3), n  # DefaultHasher::new(): zero keys
    zero = bytes(16)
    for kmer in (b"ACGTACGTACGTACG", b"A" * 15, b"acgtnACGTN", b""):
        assert host.hash_bytes(kmer) == _siphash_py(zero, len(kmer).to_bytes(8, "little") + kmer, 1, 3)
    for text in ("s00001:s00002", "seq1:seq2", ""):
        assert host.hash_str(text) == _siphash_py(zero, text.encode() + b"\xff", 1, 3)


def test_reference_unit_tests_restated(host):
    """The assertions of the reference's own unit tests, on the reference's own inputs:
    src/mash.rs:186-260 (identical sequences: Jaccard 1 / distance ~0; distance matrix of three 16-mers)
    and src/knn_graph.rs:195-387 (kNN / stranger / tree pair counts, empty and single-sequence sets)."""
    s_a, s_g = b"ATCGATCGATCGATCG", b"GGGGGGGGGGGGGGGG"
    # mash.rs test_distance_matrix (compute_distance_matrix: k = 15, sketch 1000)
    m = host.mash_matrix(["seq1", "seq2", "seq3"], [s_a, s_a, s_g], k=15)
    assert m.shape == (3, 3)
    assert m[0, 0] < 1e-6 and m[1, 1] < 1e-6 and m[2, 2] < 1e-6
    assert m[0, 1] < 1e-6 and m[1, 0] < 1e-6
    assert m[0, 2] > 0.0 and m[2, 0] > 0.0
    # mash.rs test_jaccard_identical / test_mash_distance_identical: `ATCGATCGATCG`, k = 4
    m4 = host.mash_matrix(["a", "b"], [b"ATCGATCGATCG", b"ATCGATCGATCG"], k=4)
    assert m4[0, 1] < 1e-10
    # mash.rs test_reverse_complement: reverse_complement_kmer(b"ATCG") == b"CGAT"
    assert host.reverse_complement(b"ATCG") == b"CGAT"
    # knn_graph.rs test_knn_graph_basic: extract_knn_pairs(k = 1, no random) on seq1 == seq2 != seq3
    ids3, seqs3 = ["seq1", "seq2", "seq3"], [s_a, s_a, s_g]
    p = host.plan_pairs(ids3, seqs3, "tree:1:0:0")
    assert 2 <= len(p) <= 3
    # test_tree_sampling_with_strangers: 1 nearest + 1 farthest
    p = host.plan_pairs(ids3, seqs3, "tree:1:1:0")
    assert 4 <= len(p) <= 6
    # test_only_nearest_neighbors / test_only_strangers
    assert len(host.plan_pairs(["seq1", "seq2"], [s_a, s_a], "tree:1:0:0")) == 2
    assert len(host.plan_pairs(["seq1", "seq2"], [s_a, s_g], "tree:0:1:0")) == 2
    # test_empty_sequences / test_single_sequence
    assert host.plan_pairs([], [], "tree:1:0:0") == []
    assert host.plan_pairs(["seq1"], [b"ATCG"], "tree:1:0:0") == []
    # test_build_knn_graph / test_build_knn_graph_farthest / test_knn_with_k_equals_2
    d3 = [[0.0, 0.1, 0.9], [0.1, 0.0, 0.8], [0.9, 0.8, 0.0]]
    near = host.knn_graph(d3, 1)
    assert len(near) == 3 and (0, 1) in near and (1, 0) in near and ((2, 0) in near or (2, 1) in near)
    far = host.knn_graph(d3, 1, farthest=True)
    assert len(far) == 3 and (0, 2) in far and (1, 2) in far and ((2, 0) in far or (2, 1) in far)
    d4 = [[0.0, 0.1, 0.5, 0.9], [0.1, 0.0, 0.6, 0.8], [0.5, 0.6, 0.0, 0.2], [0.9, 0.8, 0.2, 0.0]]
    assert len(host.knn_graph(d4, 2)) == 8


def test_connectivity_probability(host):
    """iterator.rs:300-334"""
    assert host.connectivity_probability(1, 0.99) == 1.0
    assert [host.connectivity_probability(n, 0.99) for n in (2, 3, 4, 5, 6, 10)] == [1.0, 0.8, 0.7, 0.6, 0.5, 0.5]
    n, x = 1024, 0.99
    assert abs(host.connectivity_probability(n, x) - (math.log(n) - math.log(-math.log(x))) / n) < 1e-15
    assert abs(host.connectivity_probability(1024, 0.99) - 0.011261) < 1e-6
    assert host.connectivity_probability(11, 1e-9) == host.connectivity_probability(11, 0.001)  # clamp
    assert host.connectivity_probability(10**7, 0.5) == 0.001                                    # floor


def test_knn_graph_kats(host):
    """knn_graph.rs:225-262, 318-340"""
    d3 = [[0.0, 0.1, 0.9], [0.1, 0.0, 0.8], [0.9, 0.8, 0.0]]
    near = host.knn_graph(d3, 1)
    assert len(near) == 3 and (0, 1) in near and (1, 0) in near and (2, 1) in near
    far = host.knn_graph(d3, 1, farthest=True)
    assert len(far) == 3 and (0, 2) in far and (1, 2) in far and (2, 0) in far
    d4 = [[0.0, 0.1, 0.5, 0.9], [0.1, 0.0, 0.6, 0.8], [0.5, 0.6, 0.0, 0.2], [0.9, 0.8, 0.2, 0.0]]
    assert len(host.knn_graph(d4, 2)) == 8
    assert host.knn_graph([[0.0, 0.5], [0.5, 0.0]], 5) == [(0, 1), (1, 0)]  # k larger than n-1
    # ties keep input order (Rust's sort_by is stable)
    assert host.knn_graph([[0, 1, 1, 1]] + [[1, 0, 1, 1], [1, 1, 0, 1], [1, 1, 1, 0]], 2)[:2] == [(0, 1), (0, 2)]


def test_plan_pairs_strategies(host):
    rng = random.Random(4)
    ids = ["seq%03d" % i for i in range(40)]
    seqs = [rand_seq(rng, 30) for _ in ids]
    allp = host.plan_pairs(ids, seqs, "none")
    assert len(allp) == 40 * 39 and allp[0] == (0, 1) and allp[39] == (1, 0)       # row-major, i != j
    assert len(host.plan_pairs(ids, seqs, "none", exclude_self=False)) == 1600
    r = host.plan_pairs(ids, seqs, "random:0.25")
    assert r == host.plan_pairs(ids, seqs, "random:0.25")                              # deterministic in the ids
    assert 0.15 < len(r) / len(allp) < 0.35 and set(r) <= set(allp)
    keep = [(i, j) for (i, j) in allp if host.hash_str("%s:%s" % (ids[i], ids[j])) / float(2**64 - 1) < 0.25]
    assert r == keep                                                                   # iterator.rs:261-281
    g = host.plan_pairs(ids, seqs, "giant:0.99")
    pg = host.connectivity_probability(40, 0.99)
    assert g == [(i, j) for (i, j) in allp if host.hash_str("%s:%s" % (ids[i], ids[j])) / float(2**64 - 1) < pg]
    assert host.plan_pairs(ids, seqs, "auto") == [(i, j) for (i, j) in allp if host.hash_str("%s:%s" % (ids[i], ids[j])) / float(2**64 - 1) < host.connectivity_probability(40, 0.95)]
    assert len(host.plan_pairs(ids[:2], seqs[:2], "giant:0.99")) == 2                 # n = 2 keeps the single edge both ways
    for bad, msg in (("giant:1.5", "Giant component probability must be between 0 and 1"), ("random:0", "Random fraction"),
                     ("tree:0:0:0.1", "At least one of k_nearest"), ("tree:1:1", "Invalid tree format"),
                     ("tree:1:1:0.1:40", "K-mer size must be between 3 and 31"), ("bogus", "Invalid sparsification strategy")):
        with pytest.raises(ValueError, match=msg):
            host.plan_pairs(ids, seqs, bad)


def test_mash_matrix_and_tree_pairs(host):
    """mash.rs:186-260 properties; knn_graph.rs:12-52 composition (nearest + farthest + hashed random,
    sorted and deduplicated)."""
    rng = random.Random(6)
    a = rand_seq(rng, 3000)
    seqs = [a, a, mutate(a, 0.02, rng), mutate(a, 0.2, rng), rand_seq(rng, 3000)]
    ids = ["s%d" % i for i in range(5)]
    m = host.mash_matrix(ids, seqs, k=15)
    assert (np.diag(m) == 0).all() and np.allclose(m, m.T)
    assert m[0, 1] < 1e-10 and 0 < m[0, 2] < m[0, 3] <= m[0, 4] == 1.0
    t = host.plan_pairs(ids, seqs, "tree:1:1:0")
    near, far = host.knn_graph(m, 1), host.knn_graph(m, 1, farthest=True)
    assert t == sorted(set(near + far))
    assert (0, 1) in t and (1, 0) in t and (0, 4) in t
    t2 = host.plan_pairs(ids, seqs, "tree:2:0:0.5:11")
    assert t2 == sorted(set(t2)) and len(t2) >= 10


def test_mash_orientation_host_only(host):
    """alignment.rs:69-94: higher strand-specific Jaccard wins, forward wins ties (e.g. unrelated or
    too-short sequences give 0 vs 0)."""
    rng = random.Random(9)
    ref = rand_seq(rng, 2000)
    fwd = mutate(ref, 0.05, rng)
    rc = host.reverse_complement(mutate(ref, 0.05, rng))
    seqs = [ref, fwd, rc, rand_seq(rng, 2000), b"ACGT", ref.lower()]
    ids = ["s%d" % i for i in range(len(seqs))]
    got = host.orient_mash(ids, seqs, [(1, 0), (2, 0), (3, 0), (4, 0), (0, 2), (5, 0)])
    assert got == [False, True, False, False, True, False]


def test_cli_mash_matrix_and_errors(host, tmp_path):
    """main.rs:281-293 (matrix to stdout, "{:.6}") and argument validation; needs no GPU."""
    from allwave_amd import build
    rng = random.Random(1)
    a = rand_seq(rng, 1500)
    fa = tmp_path / "x.fa"
    fa.write_text(">alpha desc\n%s\n>beta\n%s\n%s\n" % (a.decode(), a[:700].decode(), a[700:].decode()))
    out = subprocess.run([build.CLI_BIN, "-i", str(fa), "--mash-matrix"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0
    # identical sequences: (-1/k) * ln(1) is IEEE negative zero, which Rust's {:.6} (and printf) print with its sign
    assert out.stdout.splitlines() == ["sequence\talpha\tbeta", "alpha\t0.000000\t-0.000000", "beta\t-0.000000\t0.000000"]
    bad = subprocess.run([build.CLI_BIN, "-i", str(fa), "-p", "giant:2"], capture_output=True, text=True, timeout=60)
    assert bad.returncode != 0 and "Giant component probability must be between 0 and 1" in bad.stderr
    bad = subprocess.run([build.CLI_BIN, "-i", str(fa), "-s", "0,1", "--mash-matrix", "-x", "90"], capture_output=True, text=True)
    assert bad.returncode != 0 and "cannot be used with" in bad.stderr
    bad = subprocess.run([build.CLI_BIN, "-i", str(fa), "-k", "zzz"], capture_output=True, text=True)
    assert bad.returncode != 0 and "No sequences match the specified keep prefixes" in bad.stderr


@pytest.mark.gpu
def test_cli_end_to_end(host, oracle, tmp_path):
    """The reference's integration tests drive the binary (tests/integration_tests.rs:279-282): default
    options on the correctness KAT (:599-672 => exactly 2 X, 1 I, 1 D), -p none line count (:755-836),
    strands for a reverse-complemented query (:443-555), prefix filters (:1239-1804)."""
    from allwave_amd import build
    ref = b"ATCG" * 25
    q = bytearray(ref)
    q[10] = ord("G"); q[20] = ord("C"); del q[30]; q.insert(40, ord("A"))
    fa = tmp_path / "kat.fa"
    fa.write_text(">reference\n%s\n>query\n%s\n" % (ref.decode(), bytes(q).decode()))
    out = subprocess.run([build.CLI_BIN, "--input", str(fa)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    assert len(lines) == 2
    cg = [f for f in lines[0].split("\t") if f.startswith("cg:Z:")][0][5:]
    import re
    ops = re.findall(r"(\d+)([=XID])", cg)
    assert sum(int(n) for n, o in ops if o == "X") == 2
    assert sum(int(n) for n, o in ops if o == "I") == 1 and sum(int(n) for n, o in ops if o == "D") == 1
    # -p none on 4 sequences, one of them reverse-complemented
    rng = random.Random(3)
    base = rand_seq(rng, 900)
    seqs = [base, mutate(base, 0.03, rng), host.reverse_complement(mutate(base, 0.03, rng)), mutate(base, 0.08, rng)]
    ids = ["grpA_1", "grpA_2", "grpB_1", "grpB_2"]
    fa2 = tmp_path / "four.fa"
    fa2.write_text("".join(">%s\n%s\n" % (i, s.decode()) for i, s in zip(ids, seqs)))
    out = subprocess.run([build.CLI_BIN, "-i", str(fa2), "-p", "none", "-t", "4", "--no-progress"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    assert len(lines) == 12 and len({(l.split("\t")[0], l.split("\t")[5]) for l in lines}) == 12
    strand = {(l.split("\t")[0], l.split("\t")[5]): l.split("\t")[4] for l in lines}
    assert strand[("grpB_1", "grpA_1")] == "-" and strand[("grpA_2", "grpA_1")] == "+" and strand[("grpA_1", "grpB_1")] == "-"
    assert lines == host.all_pairs_paf(ids, seqs, "0,5,8,2,24,1", orientation="mash")
    # one process per GPU: --shard R/N keeps pairs R, R+N, ... -- the shards together are the whole run
    shards = []
    for r in range(3):
        o = subprocess.run([build.CLI_BIN, "-i", str(fa2), "-p", "none", "--no-progress", "--shard", "%d/3" % r], capture_output=True, text=True, timeout=300)
        assert o.returncode == 0, o.stderr
        assert len(o.stdout.splitlines()) == 4
        shards += o.stdout.splitlines()
    assert sorted(shards) == sorted(lines)
    assert subprocess.run([build.CLI_BIN, "-i", str(fa2), "--shard", "3/3"], capture_output=True, text=True, timeout=60).returncode != 0
    out = subprocess.run([build.CLI_BIN, "-i", str(fa2), "-p", "none", "-k", "grpA", "--wfa-orientation", "-o", str(tmp_path / "o.paf")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "Kept sequences with prefixes: 4 -> 2" in out.stderr
    assert len((tmp_path / "o.paf").read_text().splitlines()) == 2


@pytest.mark.gpu
def test_cli_reports_write_errors(host, tmp_path):
    """A PAF that cannot be written must not pass for a complete one: the reference propagates the writer's error
    (main.rs:355 `writeln!(..)?`, joined at :451-453) and exits non-zero; so does the driver (/dev/full: every
    write fails with ENOSPC)."""
    rng = random.Random(12)
    a = rand_seq(rng, 1500)
    fa = tmp_path / "in.fa"
    fa.write_text("".join(">s%d\n%s\n" % (i, mutate(a, 0.03, rng).decode()) for i in range(6)))
    cli = os.path.join(ROOT, "allwave_amd", "allwave_hip")
    ok = subprocess.run([cli, "-i", str(fa), "-o", str(tmp_path / "ok.paf"), "-p", "none", "--no-progress"], capture_output=True, timeout=300)
    assert ok.returncode == 0 and len((tmp_path / "ok.paf").read_text().splitlines()) == 30
    bad = subprocess.run([cli, "-i", str(fa), "-o", "/dev/full", "-p", "none", "--no-progress"], capture_output=True, timeout=300)
    assert bad.returncode != 0 and b"write error" in bad.stderr


@pytest.mark.gpu
def test_cli_reproduces_the_pin_files(host, tmp_path):
    """The read sets under tests/golden/pin/ through the HIP command-line driver with the reference CLI's flags (pin.sh runs a
    built allwave the same way): default mash orientation, -p none, the config's scores -- byte-identical PAF lines."""
    from allwave_amd import build
    pin = os.path.join(ROOT, "tests", "golden", "pin")
    for name, scores in (("c1", "0,1,1,1"), ("c2_8x10k", "0,5,8,2,24,1")):
        out = subprocess.run([build.CLI_BIN, "-i", os.path.join(pin, name + ".fa"), "-p", "none", "-s", scores, "-t", "4", "--no-progress"],
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        assert sorted(out.stdout.splitlines()) == sorted(open(os.path.join(pin, name + ".expected.paf")).read().splitlines()), name
