"""Shared helpers for the parity tests (seeded inputs, CIGAR utilities)."""
import random

DEFAULT_2P = (0, 5, 8, 2, 24, 1)
EDIT = (0, 1, 1, 1)
PENALTY_SETS = [DEFAULT_2P, EDIT, (0, 4, 6, 2), (0, 3, 5, 1, 20, 1), (0, 7, 0, 3), (0, 2, 12, 1, 40, 1)]


def rand_seq(rng, n, alphabet=b"ACGT"):
    return bytes(rng.choice(alphabet) for _ in range(n))


def mutate(s, d, rng, alphabet=b"ACGT"):
    out = bytearray()
    for c in s:
        u = rng.random()
        if u < 0.8 * d:
            out.append(rng.choice([x for x in alphabet if x != c] or list(alphabet)))
        elif u < 0.9 * d:
            pass
        elif u < d:
            out.append(rng.choice(alphabet))
            out.append(c)
        else:
            out.append(c)
    return bytes(out)


def rle(ops):
    """cigar_bytes_to_string of /root/reference/src/alignment.rs:347-376 (M->'=', I<->D swap)."""
    out = []
    i = 0
    tr = {ord("M"): "=", ord("X"): "X", ord("I"): "D", ord("D"): "I"}
    while i < len(ops):
        j = i
        while j < len(ops) and ops[j] == ops[i]:
            j += 1
        out.append("%d%s" % (j - i, tr.get(ops[i], "?")))
        i = j
    return "".join(out)


def random_pair(rng, maxlen=1500):
    """A pair drawn from a mix of shapes the reference's tests exercise: clean mutations, indel-heavy,
    length-mismatched, unrelated."""
    kind = rng.random()
    n = rng.choice([1, 2, 7, 33, 64, 99, 100, 101, 130, 257, 600, maxlen])
    s = rand_seq(rng, n)
    if kind < 0.55:
        t = mutate(s, rng.choice([0.0, 0.01, 0.05, 0.15, 0.3]), rng)
    elif kind < 0.7:
        cut = rng.randrange(0, n + 1)
        t = s[:cut] + rand_seq(rng, rng.choice([1, 5, 40, 200])) + s[cut:]
        t = mutate(t, 0.03, rng)
    elif kind < 0.8:
        a, b = sorted((rng.randrange(0, n + 1), rng.randrange(0, n + 1)))
        t = mutate(s[:a] + s[b:], 0.03, rng)
    elif kind < 0.9:
        t = rand_seq(rng, rng.choice([1, 3, 50, n]))
    else:
        t = mutate(s, 0.05, rng)[: max(1, n // 3)]
    if rng.random() < 0.5:
        s, t = t, s
    return s, t
