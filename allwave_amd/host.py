"""ctypes binding of liballwave_host.so -- the C++ mirror of allwave's host API (csrc/host/allwave.hpp).
Product path: no oracle, no CPU fallback (alignment entry points need the GPU)."""
import ctypes as C
import os

import numpy as np

from . import ffi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liballwave_host.so")
_LIB = None
ERRCAP = 512
_CAP = C.c_size_t(ERRCAP)  # size_t arguments are always passed as c_size_t (stack slots are 8 bytes wide)


class HostError(RuntimeError):
    pass


def load():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: run __graft_entry__.build()" % LIB_PATH)
        ffi.load()  # liballwave_hip.so first (same directory; also resolved through $ORIGIN)
        L = C.CDLL(LIB_PATH)
        L.awh_free.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def _err():
    return C.create_string_buffer(ERRCAP)


def parse_scores(s):
    """lib.rs:116-153: returns the 4 or 6 parsed scores; raises ValueError with the reference's message."""
    out = (C.c_int32 * 6)()
    n = C.c_int(0)
    e = _err()
    if load().awh_parse_scores(s.encode(), out, C.byref(n), e, _CAP) != 0:
        raise ValueError(e.value.decode())
    return tuple(out[:n.value])


def mode_and_penalties(s):
    """AlignmentMode::from_params (types.rs:107-116) + the penalties create_wfa_aligner passes down."""
    mode = C.c_int(0)
    pen = (C.c_int32 * 7)()
    e = _err()
    if load().awh_mode_from_scores(s.encode(), C.byref(mode), pen, e, _CAP) != 0:
        raise ValueError(e.value.decode())
    return ("EditDistance", "SinglePieceAffine", "TwoPieceAffine")[mode.value], tuple(pen)


def cigar_bytes_to_string(ops):
    ops = bytes(ops)
    buf = C.create_string_buffer(4 * len(ops) + 16)
    n = load().awh_cigar_to_string(ops, C.c_size_t(len(ops)), buf, C.c_size_t(len(buf)))
    if n < 0:
        raise HostError("buffer")
    return buf.value.decode()


def reverse_complement(seq):
    seq = bytes(seq)
    out = C.create_string_buffer(len(seq) + 1)
    load().awh_reverse_complement(seq, C.c_size_t(len(seq)), out)
    return out.raw[:len(seq)]


def format_paf(qid, qlen, tid, tlen, qs, qe, ts, te, is_reverse, num_matches, alignment_length, ops):
    ops = bytes(ops)
    buf = C.create_string_buffer(4 * len(ops) + 512)
    n = load().awh_format_paf(qid.encode(), C.c_size_t(qlen), tid.encode(), C.c_size_t(tlen), C.c_size_t(qs),
                              C.c_size_t(qe), C.c_size_t(ts), C.c_size_t(te), int(is_reverse),
                              C.c_size_t(num_matches), C.c_size_t(alignment_length), ops, C.c_size_t(len(ops)), buf,
                              C.c_size_t(len(buf)))
    if n < 0:
        raise HostError("buffer")
    return buf.value.decode()


def validate_cigar(ops, qlen, rlen):
    """wfa.rs:105-176: returns None when valid, else the reference's error text."""
    ops = bytes(ops)
    e = _err()
    rc = load().awh_validate_cigar(ops, C.c_size_t(len(ops)), C.c_size_t(qlen), C.c_size_t(rlen), e, _CAP)
    return None if rc == 0 else e.value.decode()


def set_engine_config(flags=0, first_row_cols=0, release=True):
    """awv_engine_config.flags / first_row_cols of the per-device engines the host library creates from now on; `release`
    destroys the cached engines first, so the next run gets a fresh one under this configuration."""
    load().awh_set_engine_config(int(flags), int(first_row_cols), int(bool(release)))


ORIENT = {"forward": 0, "wfa": 1, "mash": 2}


def _seq_args(ids, seqs):
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    data = np.frombuffer(b"".join(bytes(s) for s in seqs) + b"\0", dtype=np.uint8)
    cids = (C.c_char_p * len(ids))(*[i.encode() for i in ids])
    return cids, data, offs


def all_pairs_paf(ids, seqs, scores, orientation="wfa", exclude_self=True, device=0, sparsification="none"):
    """AllPairIterator + alignment_to_paf per record; returns the list of PAF lines."""
    cids, data, offs = _seq_args(ids, seqs)
    out = C.c_void_p()
    n = C.c_size_t(0)
    e = _err()
    rc = load().awh_all_pairs_paf(len(ids), cids, data.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p),
                                  scores.encode(), sparsification.encode(), ORIENT[orientation], int(exclude_self),
                                  device, C.byref(out), C.byref(n), e, _CAP)
    if rc != 0:
        raise HostError(e.value.decode())
    txt = C.string_at(out, n.value).decode()
    load().awh_free(out)
    return txt.splitlines()


ITER_MODES = {"for_each": 0, "next": 1, "par_for_each": 2, "par_collect": 3, "process_alignments": 4}


def iterate(ids, seqs, scores, mode="for_each", sparsification="none", orientation="forward", threads=4, chunk=0,
            resparsify=False, fail_at=-1, device=0):
    """Every consumer of the pair list (iterator.rs:101-253, lib.rs:57-68) through one hook; returns the PAF lines in arrival
    order.  `fail_at` >= 0 makes the callback throw at that record: HostError carries its message."""
    cids, data, offs = _seq_args(ids, seqs)
    out = C.c_void_p()
    n, nrec = C.c_size_t(0), C.c_size_t(0)
    e = _err()
    rc = load().awh_iterate(len(ids), cids, data.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p), scores.encode(),
                            sparsification.encode(), ORIENT[orientation], ITER_MODES[mode], int(threads), int(chunk), int(bool(resparsify)),
                            C.c_long(int(fail_at)), device, C.byref(out), C.byref(n), C.byref(nrec), e, _CAP)
    if rc != 0:
        raise HostError(e.value.decode())
    txt = C.string_at(out, n.value).decode()
    load().awh_free(out)
    return txt.splitlines()


def all_pairs_paf_count(ids, seqs, scores, orientation="forward", device=0, format_threads=8):
    """End to end: upload -> align -> D2H -> format into a counting sink. Returns (bytes, lines, secs, ffi.Stats)."""
    cids, data, offs = _seq_args(ids, seqs)
    nb, nl, secs = C.c_uint64(0), C.c_uint64(0), C.c_double(0)
    st = ffi.Stats()
    e = _err()
    rc = load().awh_all_pairs_paf_count(len(ids), cids, data.ctypes.data_as(C.c_void_p),
                                        offs.ctypes.data_as(C.c_void_p), scores.encode(), ORIENT[orientation], device,
                                        format_threads, C.byref(nb), C.byref(nl), C.byref(secs), C.byref(st), e, _CAP)
    if rc != 0:
        raise HostError(e.value.decode())
    return nb.value, nl.value, secs.value, st


def align_sequences(pattern, text, penalties, mode, device=0):
    """wfa.rs:178-258. penalties = (mismatch, o1, e1, o2, e2); mode in {"edit","affine","affine2p"}.
    Returns dict(score, cigar, matches, mismatches, insertions, deletions, alignment_length)."""
    pattern, text = bytes(pattern), bytes(text)
    pen = (C.c_int32 * 5)(*penalties)
    score = C.c_int32(0)
    cig = C.create_string_buffer(4 * (len(pattern) + len(text)) + 16)
    counts = (C.c_uint64 * 5)()
    e = _err()
    rc = load().awh_align_sequences(pattern, C.c_size_t(len(pattern)), text, C.c_size_t(len(text)), pen,
                                    {"edit": 0, "affine": 1, "affine2p": 2}[mode], device, C.byref(score), cig,
                                    C.c_size_t(len(cig)), counts, e, _CAP)
    if rc != 0:
        raise HostError(e.value.decode())
    return dict(score=score.value, cigar=cig.value.decode(), matches=counts[0], mismatches=counts[1],
                insertions=counts[2], deletions=counts[3], alignment_length=counts[4])


# ---- planner (host-only, no GPU) ---------------------------------------------------------------
def siphash(data, k0=0, k1=0, c=1, d=3):
    L = load()
    L.awh_siphash.restype = C.c_uint64
    data = bytes(data)
    return L.awh_siphash(data, C.c_size_t(len(data)), C.c_uint64(k0), C.c_uint64(k1), c, d)


def hash_bytes(data):
    """Rust: DefaultHasher over a `[u8]` (length prefix + bytes), as hash_kmer does."""
    L = load()
    L.awh_hash_bytes.restype = C.c_uint64
    data = bytes(data)
    return L.awh_hash_bytes(data, C.c_size_t(len(data)))


def hash_str(s):
    """Rust: DefaultHasher over a `str` (bytes + 0xFF), as the pair sparsifier does."""
    L = load()
    L.awh_hash_str.restype = C.c_uint64
    return L.awh_hash_str(s.encode())


def connectivity_probability(n, x):
    L = load()
    L.awh_connectivity_probability.restype = C.c_double
    return L.awh_connectivity_probability(C.c_size_t(n), C.c_double(x))


def _pairs_out(out, n):
    a = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_int64)), shape=(max(n, 1) * 2,))[:2 * n].reshape(n, 2).copy()
    load().awh_free(out)
    return [tuple(int(v) for v in r) for r in a]


def plan_pairs(ids, seqs, sparsification, exclude_self=True, resparsify=False):
    """Pair list AllPairIterator::with_options would align (iterator.rs:30-92); resparsify: planned with `-p none` first and
    then through with_sparsification (iterator.rs:101-110)."""
    cids, data, offs = _seq_args(ids, seqs)
    out = C.c_void_p()
    n = C.c_size_t(0)
    e = _err()
    rc = load().awh_plan_pairs(len(ids), cids, data.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p),
                               sparsification.encode(), int(bool(exclude_self)) | (2 if resparsify else 0), C.byref(out), C.byref(n), e, _CAP)
    if rc != 0:
        raise ValueError(e.value.decode())
    return _pairs_out(out, n.value)


def shard_assignment(pairs, lens, scores, world):
    """Shard of every pair under the cost-balanced (LPT) partition AllPairIterator::with_shard uses
    (planner::assign_shards_lpt) and the predicted per-pair costs.  pairs: int array [n, 2]; lens:
    sequence lengths; scores: "m,x,o,e[,o2,e2]"."""
    p = np.ascontiguousarray(np.asarray(pairs)[:, :2], dtype=np.int64)
    ln = np.ascontiguousarray(lens, dtype=np.int64)
    shard = np.zeros(len(p), dtype=np.uint32)
    cost = np.zeros(len(p), dtype=np.float64)
    e = _err()
    rc = load().awh_shard_pairs(p.ctypes.data_as(C.c_void_p), C.c_size_t(len(p)), ln.ctypes.data_as(C.c_void_p), C.c_size_t(len(ln)), scores.encode(),
                                C.c_size_t(int(world)), shard.ctypes.data_as(C.c_void_p), cost.ctypes.data_as(C.c_void_p), e, _CAP)
    if rc != 0:
        raise ValueError(e.value.decode())
    return shard, cost


def knn_graph(dist, k, farthest=False):
    d = np.ascontiguousarray(dist, dtype=np.float64)
    out = C.c_void_p()
    n = C.c_size_t(0)
    load().awh_knn_graph(d.ctypes.data_as(C.c_void_p), len(d), int(k), int(farthest), C.byref(out), C.byref(n))
    return _pairs_out(out, n.value)


def mash_matrix(ids, seqs, k=15):
    cids, data, offs = _seq_args(ids, seqs)
    out = np.zeros((len(ids), len(ids)), dtype=np.float64)
    load().awh_mash_matrix(len(ids), cids, data.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p), int(k),
                           out.ctypes.data_as(C.c_void_p))
    return out


def orient_mash(ids, seqs, pairs):
    cids, data, offs = _seq_args(ids, seqs)
    p = np.ascontiguousarray(pairs, dtype=np.int64).reshape(-1, 2)
    out = np.zeros(len(p), dtype=np.uint8)
    load().awh_orient_mash(len(ids), cids, data.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p),
                           p.ctypes.data_as(C.c_void_p), C.c_size_t(len(p)), out.ctypes.data_as(C.c_void_p))
    return [bool(x) for x in out]
