"""Deterministic synthetic read sets for the bench configs.

The reference's own generator (/root/reference/src/test_framework.rs:78-317) draws from rand's
StdRng and is not reproduced; this is the build's generator with the same intent: a random root,
every sequence an independently mutated copy, so that the PAIRWISE divergence is ~d (the sense in
which the reference's tests use "divergence", tests/integration_tests.rs:187).

PRNG = splitmix64 used counter-style (output k of a stream with state s0 is mix(s0 + k*G)), so the
whole set is generated vectorised and any (sequence, position) can be regenerated alone:
  root[j]     = "ACGT"[mix(seed + (j+1)*G) & 3]
  sequence i  : s0 = seed ^ (G*(i+1)); for root base j: u = u01(mix(s0 + (2j+1)*G)),
                aux = mix(s0 + (2j+2)*G); with p = d/2:
                u < 0.8p substitute by one of the 3 other bases ((code + 1 + aux%3) & 3);
                0.8p <= u < 0.9p delete; 0.9p <= u < p insert "ACGT"[aux&3] before it; else copy.
IDs are s{i:05}; upper-case ACGT only; all '+' strand.
"""
import numpy as np

G = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def _mix(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _u01(z):
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def root_sequence(length, seed):
    with np.errstate(over="ignore"):
        j = np.arange(1, length + 1, dtype=np.uint64)
        codes = (_mix(np.uint64(seed) + j * G) & np.uint64(3)).astype(np.uint8)
    return codes


def mutate(root_codes, i, d, seed, length=None):
    """Sequence i as uint8 ASCII. `length` (optional) mutates only a prefix of the root."""
    rc = root_codes if length is None else root_codes[:length]
    n = len(rc)
    with np.errstate(over="ignore"):
        s0 = np.uint64(seed) ^ (G * np.uint64(i + 1))
        j = np.arange(n, dtype=np.uint64)
        u = _u01(_mix(s0 + (np.uint64(2) * j + np.uint64(1)) * G))
        aux = _mix(s0 + (np.uint64(2) * j + np.uint64(2)) * G)
    p = d / 2.0
    sub = u < 0.8 * p
    dele = (~sub) & (u < 0.9 * p)
    ins = (~sub) & (~dele) & (u < p)
    code = rc.copy()
    code[sub] = ((rc[sub].astype(np.uint64) + np.uint64(1) + aux[sub] % np.uint64(3)) & np.uint64(3)).astype(np.uint8)
    out = np.empty((n, 2), dtype=np.uint8)
    out[:, 0] = _BASES[(aux & np.uint64(3)).astype(np.uint8)]
    out[:, 1] = _BASES[code]
    keep = np.empty((n, 2), dtype=bool)
    keep[:, 0] = ins
    keep[:, 1] = ~dele
    return out[keep]


def generate(nseq, length, d, seed, mixed_lengths=None):
    """Returns (uint8 concat, uint64 offsets[n+1], ids).  mixed_lengths=(lo, hi) draws per-sequence
    prefix lengths lo + next()%(hi-lo+1) (config 5)."""
    root = root_sequence(length, seed)
    parts = []
    for i in range(nseq):
        ln = None
        if mixed_lengths is not None:
            lo, hi = mixed_lengths
            with np.errstate(over="ignore"):
                r = _mix(np.uint64(seed) ^ (G * np.uint64(i + 1)) ^ np.uint64(0xA5A5A5A5A5A5A5A5))
            ln = int(lo + int(r) % (hi - lo + 1))
        parts.append(mutate(root, i, d, seed, ln))
    offs = np.zeros(nseq + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(p) for p in parts], dtype=np.uint64)
    data = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8)
    ids = ["s%05d" % i for i in range(nseq)]
    return data, offs, ids


def all_pairs(n, exclude_self=True):
    """Row-major (i, j), i != j -- the `-p none` enumeration of
    /root/reference/src/iterator.rs:38-46."""
    i, j = np.meshgrid(np.arange(n, dtype=np.int32), np.arange(n, dtype=np.int32), indexing="ij")
    pairs = np.stack([i.ravel(), j.ravel()], axis=1)
    if exclude_self:
        pairs = pairs[pairs[:, 0] != pairs[:, 1]]
    return np.ascontiguousarray(pairs)


#: BASELINE.json configs (name -> generator arguments, scores, sparsification)
CONFIGS = {
    "c1": dict(nseq=8, length=1000, d=0.05, seed=1, scores=(0, 1, 1, 1), sparsify="none"),
    "c2": dict(nseq=256, length=10000, d=0.05, seed=2, scores=(0, 5, 8, 2, 24, 1), sparsify="none"),
    "c3": dict(nseq=4096, length=10000, d=0.05, seed=3, scores=(0, 5, 8, 2, 24, 1), sparsify="none"),
    "c4": dict(nseq=1024, length=100000, d=0.02, seed=4, scores=(0, 5, 8, 2, 24, 1), sparsify="giant:0.99"),
    "c5": dict(nseq=512, length=50000, d=0.10, seed=5, scores=(0, 5, 8, 2, 24, 1), sparsify="tree:3:1:0.1",
               mixed_lengths=(1000, 50000)),
}
