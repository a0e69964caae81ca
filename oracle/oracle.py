"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by the product package (allwave_amd/), which must fail loudly without the HIP library.
See oracle/biwfa_oracle.h for what is restated and the "parity unpinned" note.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Penalties(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("match", "mismatch", "gap_open1", "gap_ext1", "gap_open2", "gap_ext2", "two_piece")]

    @classmethod
    def from_scores(cls, scores):
        """scores = (match, x, o, e) or (match, x, o1, e1, o2, e2): mode selection follows
        AlignmentMode::from_params (/root/reference/src/types.rs:107-116) + create_wfa_aligner
        (/root/reference/src/alignment.rs:263-289): 6 scores -> 2-piece; o==e==x -> "edit"
        which is gap-affine (x, x, x); else gap-affine (x, o, e)."""
        s = list(scores)
        if len(s) == 6:
            return cls(s[0], s[1], s[2], s[3], s[4], s[5], 1)
        if len(s) == 4:
            return cls(s[0], s[1], s[2], s[3], 0, 0, 0)
        raise ValueError("Invalid number of scores: %d. Expected 4 or 6 values." % len(s))


class Stats(C.Structure):
    _fields_ = [("cell_steps", C.c_uint64), ("extend_bytes", C.c_uint64), ("n_breakpoints", C.c_uint32),
                ("n_base", C.c_uint32), ("n_trivial", C.c_uint32), ("max_level", C.c_uint32),
                ("max_width", C.c_uint32), ("overlap_rows", C.c_uint32)]


class PairResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("penalty", C.c_int32), ("cigar_len", C.c_int32),
                ("num_matches", C.c_int32), ("num_mismatches", C.c_int32), ("num_ins_text", C.c_int32),
                ("num_del_pattern", C.c_int32), ("cigar_hash", C.c_uint64)]


PAIR_RESULT_DTYPE = np.dtype([("status", "<i4"), ("penalty", "<i4"), ("cigar_len", "<i4"),
                              ("num_matches", "<i4"), ("num_mismatches", "<i4"), ("num_ins_text", "<i4"),
                              ("num_del_pattern", "<i4"), ("_pad", "<i4"), ("cigar_hash", "<u8")])


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in
            ("biwfa_oracle.c", "gotoh.c", "cigar_check.c", "allpairs_cpu.c", "biwfa_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.awo_aligner_new.restype = C.c_void_p
        L.awo_aligner_new.argtypes = [C.POINTER(Penalties)]
        L.awo_aligner_delete.argtypes = [C.c_void_p]
        sig = [C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_void_p, C.c_int,
               C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(Stats)]
        L.awo_align.argtypes = sig
        L.awo_align_unidirectional.argtypes = sig
        L.awo_gotoh_penalty.restype = C.c_int64
        L.awo_gotoh_penalty.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(Penalties)]
        L.awo_cigar_check.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int,
                                      C.POINTER(Penalties), C.POINTER(C.c_int64)]
        for fn in (L.awo_all_pairs, L.awo_all_pairs_fast):
            fn.restype = C.c_double
            fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.POINTER(Penalties), C.c_int,
                           C.c_void_p, C.POINTER(Stats), C.POINTER(C.c_uint64)]
        L.awo_aligner_set_fast_overlap.argtypes = [C.c_void_p, C.c_int]
        L.awo_fnv1a.restype = C.c_uint64
        L.awo_fnv1a.argtypes = [C.c_char_p, C.c_int64]
        _LIB = L
    return _LIB


class Aligner:
    """One WFA2-style aligner (mirrors a cached lib_wfa2 AffineWavefronts, alignment.rs:11-22)."""

    def __init__(self, scores):
        self.pen = scores if isinstance(scores, Penalties) else Penalties.from_scores(scores)
        self._h = lib().awo_aligner_new(C.byref(self.pen))
        if not self._h:
            raise ValueError("penalties rejected (match must be 0, x>0, e>0)")

    def set_fast_overlap(self, on):
        """Exact pre-filter for the overlap search (CPU-baseline mode); results are identical."""
        lib().awo_aligner_set_fast_overlap(self._h, int(bool(on)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().awo_aligner_delete(self._h)
            self._h = None

    def _run(self, fn, pattern, text, stats):
        pattern, text = bytes(pattern), bytes(text)
        cap = len(pattern) + len(text) + 1
        buf = C.create_string_buffer(cap)
        n, pen = C.c_int(0), C.c_int(0)
        rc = fn(self._h, pattern, len(pattern), text, len(text), buf, cap, C.byref(n), C.byref(pen),
                C.byref(stats) if stats is not None else None)
        if rc != 0:
            raise RuntimeError("oracle alignment failed: status %d" % rc)
        return pen.value, buf.raw[:n.value]

    def align(self, pattern, text, stats=None):
        """BiWFA path (MemoryMode::Ultralow). Returns (penalty, op_bytes)."""
        return self._run(lib().awo_align, pattern, text, stats)

    def align_unidirectional(self, pattern, text, stats=None):
        """Plain WFA + backtrace on the whole problem (cross-check only)."""
        return self._run(lib().awo_align_unidirectional, pattern, text, stats)


def gotoh_penalty(pattern, text, scores):
    pen = scores if isinstance(scores, Penalties) else Penalties.from_scores(scores)
    pattern, text = bytes(pattern), bytes(text)
    return lib().awo_gotoh_penalty(pattern, len(pattern), text, len(text), C.byref(pen))


def cigar_check(cigar, pattern, text, scores):
    """Returns (rc, rescored_penalty); rc == 0 means valid."""
    pen = scores if isinstance(scores, Penalties) else Penalties.from_scores(scores)
    cigar, pattern, text = bytes(cigar), bytes(pattern), bytes(text)
    out = C.c_int64(0)
    rc = lib().awo_cigar_check(cigar, len(cigar), pattern, len(pattern), text, len(text), C.byref(pen),
                               C.byref(out))
    return rc, out.value


def fnv1a(data):
    """FNV-1a of a byte string (the per-pair CIGAR hash all_pairs() reports)."""
    data = bytes(data)
    return lib().awo_fnv1a(data, len(data))


def all_pairs(seqs, offsets, pairs, scores, nthreads=1, want_paf=False, fast_overlap=False):
    """Thread-pool all-pairs run. seqs: uint8 array (concatenated), offsets: uint64[n+1],
    pairs: int32[npairs,2]. Returns (seconds, results structured array, Stats, paf_bytes)."""
    pen = scores if isinstance(scores, Penalties) else Penalties.from_scores(scores)
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    pairs = np.ascontiguousarray(pairs, dtype=np.int32)
    res = np.zeros(len(pairs), dtype=PAIR_RESULT_DTYPE)
    assert res.dtype.itemsize == C.sizeof(PairResult)
    st = Stats()
    paf = C.c_uint64(0)
    fn = lib().awo_all_pairs_fast if fast_overlap else lib().awo_all_pairs
    secs = fn(seqs.ctypes.data, offsets.ctypes.data, len(offsets) - 1, pairs.ctypes.data,
                               len(pairs), C.byref(pen), int(nthreads), res.ctypes.data, C.byref(st),
                               C.byref(paf) if want_paf else None)
    if secs < 0:
        raise RuntimeError("oracle all-pairs run failed")
    return secs, res, st, paf.value
