/*
 * allwave_hip.h -- C ABI of the MI355X-native BiWFA engine (liballwave_hip.so).
 *
 * This is the drop-in boundary for allwave's per-pair hot path.  The reference reaches the
 * arithmetic through the lib_wfa2 crate (Rust FFI over WFA2-lib, not in /root/reference):
 *
 *   reference call (file:line, relative to /root/reference)            replaced by
 *   -----------------------------------------------------------------  --------------------------
 *   AffineWavefronts::with_penalties_and_memory_mode(m,x,o,e,Ultralow)  awv_penalties {two_piece=0}
 *     src/alignment.rs:265-278, src/wfa.rs:188-204
 *   AffineWavefronts::with_penalties_affine2p_and_memory_mode(...)      awv_penalties {two_piece=1}
 *     src/alignment.rs:279-287, src/wfa.rs:208-216
 *   set_alignment_scope(Alignment) / set_alignment_span(End2End) /      fixed behaviour of the engine
 *   set_heuristic(None)   src/alignment.rs:226-228, src/wfa.rs:221-223  (end-to-end, exact, with CIGAR)
 *   wf.align(query, target) -> AlignmentStatus   src/alignment.rs:231   awv_align_pairs / awv_align_one
 *   wf.score()                                   src/alignment.rs:235   awv_result.score (= -penalty)
 *   wf.cigar() -> &[u8]                          src/alignment.rs:236   CIGAR arena + awv_result.cigar_off/len
 *   per-thread aligner cache                     src/alignment.rs:11-22 awv_engine (owns all device state)
 *
 * A per-pair synchronous call cannot feed a GPU, so the primary entry point is batched:
 * the caller hands over the sequence set once and then lists of (query, target) index pairs
 * -- exactly the pair list AllPairIterator materialises (src/iterator.rs:38-50).
 *
 * Conventions kept from the reference boundary:
 *   - argument order: pattern = query, text = target (tests/debug/test_wfa_order.rs:1-31);
 *   - CIGAR op bytes, one per column, WFA2 alphabet: 'M' match, 'X' mismatch, 'I' consumes
 *     the text/target, 'D' consumes the pattern/query (src/alignment.rs:331-338,
 *     src/wfa.rs:128-149) -- so count_cigar_operations / parse_cigar_lengths /
 *     cigar_bytes_to_string (src/alignment.rs:292-376) apply unchanged;
 *   - bytes are compared verbatim (case-sensitive, 'N' == 'N');
 *   - only status 0 is success (AlignmentStatus::Completed, src/alignment.rs:233-258); the
 *     caller maps anything else to the "empty" result (src/alignment.rs:49-64).
 *
 * Plain C: pointers and sizes only, no C++ or torch types.  No global state; one engine per
 * GPU per process; calls on one engine must not overlap (single submitter).
 */
#ifndef ALLWAVE_HIP_H
#define ALLWAVE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AWV_ABI_VERSION 3

/* engine-level return codes (negative = failure; never aborts the process) */
#define AWV_OK 0
#define AWV_ERR_NO_DEVICE (-1)   /* no usable HIP device: the product path has no CPU fallback */
#define AWV_ERR_HIP (-2)         /* a HIP runtime call failed; see awv_last_error() */
#define AWV_ERR_ARG (-3)
#define AWV_ERR_PENALTIES (-4)   /* match != 0, x <= 0, e <= 0 ... (WFA2 would transform/reject) */
#define AWV_ERR_OOM (-5)
#define AWV_ERR_STATE (-6)       /* e.g. align before set_sequences */
#define AWV_ERR_SINK (-7)        /* the sink callback returned non-zero (first error wins) */

/* per-pair status (awv_result.status) */
#define AWV_ST_COMPLETED 0
#define AWV_ST_CAPACITY 1        /* an internal capacity bound was hit (wavefront width / history) */
#define AWV_ST_INTERNAL 2        /* invariant violated (would be a bug) */
#define AWV_ST_MAX_STEPS 3       /* step guard tripped */

typedef struct awv_engine awv_engine;

typedef struct {
  int32_t device;          /* HIP device ordinal */
  int32_t workgroups;      /* persistent workgroups = pairs in flight (0 = engine default: 16 per CU) */
  int64_t max_batch_pairs; /* pairs per launch (0 = default) */
  int64_t max_arena_bytes; /* CIGAR arena budget per launch (0 = default 8 GiB) */
  int32_t flags;           /* AWV_F_* */
  int32_t first_row_cols;  /* 0 = default; > 0 caps the row width (columns, >= 2048) of a batch's first attempt: pairs whose
                              wavefronts outgrow it come back CAPACITY and are re-run wider (diagnostic / test hook) */
  int64_t max_scratch_bytes; /* cap on the per-workgroup wavefront arenas (0 = default 160 GiB) */
} awv_engine_config;

/* ---- flags a caller has a use for */
#define AWV_F_KEEP_ON_DEVICE 1 /* do not copy CIGARs back (kernel-only measurements) */
#define AWV_F_NO_ARENA_PROBE 32 /* take the first ring-arena allocation as it comes (default: allocate up to four candidates and keep the
                                   one a 1 ms traffic probe finds fastest -- worth up to 6 % of kernel time, costs 1-3 s once per engine:
                                   for short-lived processes with little work) */
#define AWV_F_NO_RERUN 1024    /* a pair whose wavefronts outgrow the first attempt's rows keeps status AWV_ST_CAPACITY instead of being re-run with
                                   wider rows (fail fast; with first_row_cols: the way to see a failed pair's record end to end) */
/* ---- variant pins: tests and A/B measurements only.  Results are identical under every one of them (tests/test_gpu_parity.py
 * runs the pairs of variants against each other and against the oracle); a product caller leaves them alone. */
#define AWV_F_FORCE_INT32 2    /* always use 32-bit wavefront rows, for every sub-problem (default: 16-bit when lengths < 32760,
                                  and -- in the four- and sixteen-wave flavours -- when only the shorter length is: rows of
                                  min(h, v); a launch with 32-bit rows searches the sub-problems that fit with 16-bit rows) */
#define AWV_F_NO_WIDE16 256    /* 32-bit rows whenever the longer sequence has 32760 bases or more (no min(h, v) rows) */
#define AWV_F_NO_PACKED_SEQ 4  /* never use the 2-bit packed sequences, staged in LDS or in place (raw-byte probes from HBM only) */
#define AWV_F_ONE_WAVE 8       /* always one wave per pair (default: four waves per pair for small batches, long sequences and unequal lengths, sixteen for a few very unequal pairs) */
#define AWV_F_FOUR_WAVES 16    /* always four waves per pair */
#define AWV_F_NO_CHAIN 128     /* multi-step passes of one sweep only (no chaining of sweeps through registers / LDS) */
#define AWV_F_NO_DEEP 512      /* the margin zone of a breakpoint search runs step by step (the round-2 path) instead of in passes that store every I/D row */
#define AWV_F_SINGLE_STEP 64   /* never use multi-step passes (every step stores all five rows; the round-1 kernel path) */

/* penalties as allwave passes them to lib_wfa2 (src/alignment.rs:263-289) */
typedef struct {
  int32_t match;     /* must be 0 */
  int32_t mismatch;  /* x  */
  int32_t gap_open1; /* o1 */
  int32_t gap_ext1;  /* e1 */
  int32_t gap_open2; /* o2, used when two_piece */
  int32_t gap_ext2;  /* e2, used when two_piece */
  int32_t two_piece; /* 0 = gap-affine (also allwave's "edit" mode x,x,x), 1 = 2-piece */
} awv_penalties;

typedef struct {
  int32_t q_idx;     /* query  = pattern */
  int32_t t_idx;     /* target = text */
  int32_t q_revcomp; /* align reverse_complement(query) (src/alignment.rs:178-190) */
} awv_pair;

typedef struct {
  int32_t status;         /* AWV_ST_* */
  int32_t penalty;        /* >= 0 */
  int32_t score;          /* = -penalty: what WFA2's cigar->score / wf.score() reports */
  uint32_t cigar_len;     /* op bytes */
  uint64_t cigar_off;     /* offset of the op bytes in the arena handed to the sink */
  int32_t num_matches;    /* #M */
  int32_t num_mismatches; /* #X */
  int32_t num_ins;        /* #I (text/target consumed) */
  int32_t num_del;        /* #D (pattern/query consumed) */
  int32_t q_end;          /* #M + #X + #D  (src/alignment.rs:320-344) */
  int32_t t_end;          /* #M + #X + #I */
} awv_result;

/* Sink: called once per launch batch with results[first..first+n) and the batch's CIGAR arena
 * (valid only during the call).  Calls never overlap and come in batch order; with several batches
 * in one awv_align_pairs call all but the last come from an engine-owned helper thread while the
 * next batch is being aligned (the reference's callback is invoked from worker threads too,
 * src/iterator.rs:208-252).  Return non-zero to stop: the call then fails with AWV_ERR_SINK before
 * any further sink call. */
typedef int (*awv_sink)(void* user, int64_t first, int64_t n, const awv_result* results,
                        const uint8_t* cigar_arena);

typedef struct {
  double kernel_ms;         /* HIP-event time of the alignment kernel launches, last call */
  double h2d_ms, d2h_ms;    /* copies, last call */
  uint64_t launches;        /* kernel launches, last call */
  uint64_t cell_steps;      /* wavefront cells computed (all components count as one cell) */
  uint64_t extend_steps;    /* 8-byte compare iterations of the extend loop, summed over lanes */
  uint64_t n_breakpoints;   /* BiWFA breakpoint searches */
  uint64_t n_base;          /* base-case alignments */
  uint64_t overlap_scans;   /* wavefront pairs scanned by the overlap search */
  uint64_t aligned_bp;      /* sum of query lengths of completed pairs */
  uint64_t pairs_completed;
  uint64_t scratch_bytes;   /* device scratch currently allocated */
  /* shader-clock cycles summed over workgroups; only filled by the -DAWV_PROF diagnostic build:
   * [0] total, [1] step compute, [2] step barrier wait, [3] step finalize, [4] overlap search,
   * [5] base-case steps, [6] backtrace, [7] CIGAR emission, [8] number of fused step passes,
   * [9..13] inside the step: row loads, DP arithmetic, extend, stores, reductions */
  uint64_t prof[14];
  uint64_t restarts;          /* breakpoint searches run again step by step (multi-step passes met too early) */
  uint64_t multi_cell_steps;  /* cells computed by multi-step passes (I/D rows kept in registers) */
  uint64_t windows[4];        /* window iterations: [0] step-by-step (one step each), [1] multi-step passes (T steps each), [2] base case step-by-step, [3] base case multi-step passes */
  /* the shader clock the kernels actually ran at (ABI 3): every persistent workgroup stamps s_memtime (shader cycles) and
   * s_memrealtime (constant-rate ticks, `clock_tick_khz`) when it starts and when it has drained the work queue; summed over the
   * workgroups of the call's launches.  sustained clock = clock_cycles / clock_ticks * clock_tick_khz kHz */
  uint64_t clock_cycles, clock_ticks;
  uint64_t clock_tick_khz;    /* hipDeviceAttributeWallClockRate (100 000 on MI355X) */
  uint64_t deep_cell_steps;   /* of multi_cell_steps: cells of passes that also store every I/D row (the margin zone before the two searches meet) */
} awv_stats;

int awv_abi_version(void);
const char* awv_last_error(void); /* thread-local description of the last failure */

int awv_engine_create(const awv_engine_config* cfg, awv_engine** out);
void awv_engine_destroy(awv_engine* e);

/* Hands over the sequence set: n sequences, concatenated bytes, offsets[n+1].  The engine keeps
 * its own device copies (forward, reversed, and reverse-complement variants). */
int awv_engine_set_sequences(awv_engine* e, int32_t n, const uint8_t* concat_bytes,
                             const uint64_t* offsets);

/* Aligns pairs[0..npairs).  `out` (nullable) receives all results; `sink` (nullable) streams
 * them with their CIGARs.  out[i].cigar_off is relative to the batch arena passed to the sink. */
int awv_align_pairs(awv_engine* e, const awv_penalties* pen, const awv_pair* pairs, int64_t npairs,
                    awv_result* out, awv_sink sink, void* user);

/* Convenience mirror of AffineWavefronts::align + score + cigar for one pair
 * (src/alignment.rs:231-236).  cigar_buf needs plen + tlen bytes. */
int awv_align_one(awv_engine* e, const awv_penalties* pen, const uint8_t* pattern, int32_t plen,
                  const uint8_t* text, int32_t tlen, awv_result* result, uint8_t* cigar_buf,
                  size_t cigar_cap);

int awv_engine_stats(const awv_engine* e, awv_stats* out);

#ifdef __cplusplus
}
#endif
#endif
