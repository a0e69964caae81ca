"""ctypes binding of liballwave_hip.so (include/allwave_hip.h).

This is the product path: it never imports oracle/ and has no CPU fallback -- a missing library
or a missing GPU raises immediately.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AWV_HIP_LIB") or os.path.join(_HERE, "liballwave_hip.so")

AWV_OK = 0
AWV_ERR_NO_DEVICE = -1
AWV_ERR_SINK = -7
AWV_F_KEEP_ON_DEVICE = 1
AWV_F_FORCE_INT32 = 2
AWV_F_NO_PACKED_SEQ = 4
AWV_F_ONE_WAVE = 8
AWV_F_FOUR_WAVES = 16
AWV_F_NO_ARENA_PROBE = 32
AWV_F_SINGLE_STEP = 64
AWV_F_NO_CHAIN = 128
AWV_F_NO_WIDE16 = 256
AWV_F_NO_DEEP = 512
AWV_F_NO_RERUN = 1024

#: every symbol include/allwave_hip.h declares
EXPORTS = ("awv_abi_version", "awv_last_error", "awv_engine_create", "awv_engine_destroy",
           "awv_engine_set_sequences", "awv_align_pairs", "awv_align_one", "awv_engine_stats")


class EngineConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("workgroups", C.c_int32), ("max_batch_pairs", C.c_int64),
                ("max_arena_bytes", C.c_int64), ("flags", C.c_int32), ("first_row_cols", C.c_int32),
                ("max_scratch_bytes", C.c_int64)]


class Penalties(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("match", "mismatch", "gap_open1", "gap_ext1", "gap_open2", "gap_ext2", "two_piece")]

    @classmethod
    def from_scores(cls, scores):
        """(m,x,o,e) or (m,x,o1,e1,o2,e2) -> aligner construction of
        /root/reference/src/alignment.rs:263-289 (4 scores: gap-affine, incl. the "edit" mode
        x,x,x; 6 scores: 2-piece)."""
        s = [int(v) for v in scores]
        if len(s) == 6:
            return cls(s[0], s[1], s[2], s[3], s[4], s[5], 1)
        if len(s) == 4:
            return cls(s[0], s[1], s[2], s[3], 0, 0, 0)
        raise ValueError("Invalid number of scores: %d. Expected 4 or 6 values." % len(s))


class Stats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("h2d_ms", C.c_double), ("d2h_ms", C.c_double),
                ("launches", C.c_uint64), ("cell_steps", C.c_uint64), ("extend_steps", C.c_uint64),
                ("n_breakpoints", C.c_uint64), ("n_base", C.c_uint64), ("overlap_scans", C.c_uint64),
                ("aligned_bp", C.c_uint64), ("pairs_completed", C.c_uint64), ("scratch_bytes", C.c_uint64),
                ("prof", C.c_uint64 * 14), ("restarts", C.c_uint64), ("multi_cell_steps", C.c_uint64), ("windows", C.c_uint64 * 4),
                ("clock_cycles", C.c_uint64), ("clock_ticks", C.c_uint64), ("clock_tick_khz", C.c_uint64),
                ("deep_cell_steps", C.c_uint64)]


PAIR_DTYPE = np.dtype([("q_idx", "<i4"), ("t_idx", "<i4"), ("q_revcomp", "<i4")])
RESULT_DTYPE = np.dtype([("status", "<i4"), ("penalty", "<i4"), ("score", "<i4"), ("cigar_len", "<u4"),
                         ("cigar_off", "<u8"), ("num_matches", "<i4"), ("num_mismatches", "<i4"),
                         ("num_ins", "<i4"), ("num_del", "<i4"), ("q_end", "<i4"), ("t_end", "<i4")])

SINK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p)

_LIB = None


def load():
    """Loads the in-tree library; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.awv_abi_version.restype = C.c_int
        L.awv_last_error.restype = C.c_char_p
        L.awv_engine_create.argtypes = [C.POINTER(EngineConfig), C.POINTER(C.c_void_p)]
        L.awv_engine_destroy.argtypes = [C.c_void_p]
        L.awv_engine_destroy.restype = None
        L.awv_engine_set_sequences.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.awv_align_pairs.argtypes = [C.c_void_p, C.POINTER(Penalties), C.c_void_p, C.c_int64, C.c_void_p,
                                      SINK_FN, C.c_void_p]
        L.awv_align_one.argtypes = [C.c_void_p, C.POINTER(Penalties), C.c_char_p, C.c_int32, C.c_char_p, C.c_int32,
                                    C.c_void_p, C.c_void_p, C.c_size_t]
        L.awv_engine_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        _LIB = L
    return _LIB


class EngineError(RuntimeError):
    def __init__(self, code, where):
        msg = load().awv_last_error()
        super().__init__("%s failed with %d: %s" % (where, code, (msg or b"").decode(errors="replace")))
        self.code = code


class Engine:
    """One engine per GPU (owns device copies of the sequences, scratch arenas, one stream)."""

    def __init__(self, device=0, workgroups=0, max_batch_pairs=0, max_arena_bytes=0, flags=0, max_scratch_bytes=0, first_row_cols=0):
        L = load()
        self._h = C.c_void_p()
        cfg = EngineConfig(device, workgroups, max_batch_pairs, max_arena_bytes, flags, first_row_cols, max_scratch_bytes)
        rc = L.awv_engine_create(C.byref(cfg), C.byref(self._h))
        if rc != AWV_OK:
            self._h = C.c_void_p()
            raise EngineError(rc, "awv_engine_create")
        self.nseq = 0

    def close(self):
        if getattr(self, "_h", None):
            load().awv_engine_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def set_sequences(self, seqs):
        """seqs: list of bytes, or (uint8 array, uint64 offsets[n+1])."""
        if isinstance(seqs, tuple):
            data = np.ascontiguousarray(seqs[0], dtype=np.uint8)
            offs = np.ascontiguousarray(seqs[1], dtype=np.uint64)
        else:
            offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
            offs[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
            data = np.frombuffer(b"".join(bytes(s) for s in seqs), dtype=np.uint8)
            if data.size == 0:
                data = np.zeros(1, dtype=np.uint8)
        n = len(offs) - 1
        rc = load().awv_engine_set_sequences(self._h, n, data.ctypes.data, offs.ctypes.data)
        if rc != AWV_OK:
            raise EngineError(rc, "awv_engine_set_sequences")
        self.nseq = n

    def align_pairs(self, scores, pairs, want_cigars=True):
        """pairs: int array [n,2] (q,t) or [n,3] (q,t,revcomp), or a PAIR_DTYPE array.
        Returns (results structured array, list of op-byte strings or None)."""
        pen = scores if isinstance(scores, Penalties) else Penalties.from_scores(scores)
        if not (isinstance(pairs, np.ndarray) and pairs.dtype == PAIR_DTYPE):
            a = np.asarray(pairs, dtype=np.int32)
            a = a.reshape(len(a), -1) if len(a) else np.zeros((0, 2), dtype=np.int32)
            p = np.zeros(len(a), dtype=PAIR_DTYPE)
            if len(a):
                p["q_idx"], p["t_idx"] = a[:, 0], a[:, 1]
                if a.shape[1] > 2:
                    p["q_revcomp"] = a[:, 2]
            pairs = p
        pairs = np.ascontiguousarray(pairs)
        res = np.zeros(len(pairs), dtype=RESULT_DTYPE)
        cigars = [None] * len(pairs) if want_cigars else None

        def _sink(user, first, n, rptr, arena):
            if cigars is not None and arena:
                r = np.ctypeslib.as_array(C.cast(rptr, C.POINTER(C.c_uint8)), shape=(n * RESULT_DTYPE.itemsize,))
                r = r.view(RESULT_DTYPE)
                for i in range(n):
                    if r["status"][i] == 0:
                        cigars[first + i] = C.string_at(arena + int(r["cigar_off"][i]), int(r["cigar_len"][i]))
            return 0

        cb = SINK_FN(_sink) if want_cigars else SINK_FN()
        rc = load().awv_align_pairs(self._h, C.byref(pen), pairs.ctypes.data, len(pairs), res.ctypes.data, cb, None)
        if rc != AWV_OK:
            raise EngineError(rc, "awv_align_pairs")
        return res, cigars

    def align_one(self, scores, pattern, text):
        """Mirror of wf.align + wf.score + wf.cigar (alignment.rs:231-236). Returns (result, op_bytes)."""
        pen = scores if isinstance(scores, Penalties) else Penalties.from_scores(scores)
        pattern, text = bytes(pattern), bytes(text)
        res = np.zeros(1, dtype=RESULT_DTYPE)
        cap = len(pattern) + len(text) + 1
        buf = C.create_string_buffer(cap)
        rc = load().awv_align_one(self._h, C.byref(pen), pattern, len(pattern), text, len(text), res.ctypes.data,
                                  buf, cap)
        if rc != AWV_OK:
            raise EngineError(rc, "awv_align_one")
        return res[0], buf.raw[:int(res[0]["cigar_len"])]

    def stats(self):
        st = Stats()
        rc = load().awv_engine_stats(self._h, C.byref(st))
        if rc != AWV_OK:
            raise EngineError(rc, "awv_engine_stats")
        return st
