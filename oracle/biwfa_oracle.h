/*
 * biwfa_oracle.h -- CPU restatement of the alignment arithmetic on allwave's hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the
 * checker / reported CPU baseline.  The shipped path is allwave_amd/csrc (HIP) behind
 * include/allwave_hip.h and fails loudly without a GPU.
 *
 * What this restates: the call `wf.align(query, target)` + `wf.score()` + `wf.cigar()` made by
 * the reference at /root/reference/src/alignment.rs:226-236 and src/wfa.rs:221-231 on an
 * aligner built by alignment.rs:263-289 / wfa.rs:185-218 (gap-affine or 2-piece gap-affine,
 * MemoryMode::Ultralow => BiWFA, End2End span, Alignment scope, no heuristic).
 *
 * The arithmetic itself lives in a third-party dependency that is NOT in /root/reference:
 *   lib_wfa2 @ 2f9d9a48addee5185d8ff6ed0594182558d60818 (Cargo.toml:27, Cargo.lock:598-601),
 *   a Rust FFI wrapper over WFA2-lib (C, smarco/WFA2-lib; submodule commit not recorded).
 * It is restated here from the published algorithms (Marco-Sola et al., Bioinformatics 2021
 * [WFA] and 2023 [BiWFA]) with WFA2-lib's conventions as recorded in SURVEY.md Appendix A.
 *
 * PARITY UNPINNED: the reference holds no golden CIGAR or score for this path (SURVEY.md
 * section 8c) and neither the reference nor WFA2-lib can be built here.  What IS pinned:
 *   - the penalty equals an independent full Gotoh DP (oracle/gotoh.c) -- optimality;
 *   - the CIGAR is valid and re-scores to that penalty (wfa.rs:105-176 restated);
 *   - the reference's known-answer properties (integration_tests.rs:599-672 etc.).
 * Tie-breaking (which of several optimal CIGARs) follows Appendix A and is expected, not
 * proven, to equal WFA2-lib's.
 */
#ifndef BIWFA_ORACLE_H
#define BIWFA_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Penalties as the reference passes them to lib_wfa2 (alignment.rs:265-287). */
typedef struct {
  int32_t match;       /* must be 0 (types.rs:50); non-zero is rejected */
  int32_t mismatch;    /* x  > 0 */
  int32_t gap_open1;   /* o1 >= 0 */
  int32_t gap_ext1;    /* e1 > 0 */
  int32_t gap_open2;   /* o2 (two_piece only) */
  int32_t gap_ext2;    /* e2 (two_piece only) */
  int32_t two_piece;   /* 0: gap-affine, 1: 2-piece gap-affine */
} awo_penalties_t;

typedef struct {
  uint64_t cell_steps;     /* sum of (hi-lo+1) over every compute-next call (SURVEY 8d unit) */
  uint64_t extend_bytes;   /* bytes compared by extend (matches + the terminating mismatch) */
  uint32_t n_breakpoints;  /* BiWFA find_breakpoint calls */
  uint32_t n_base;         /* base-case (plain WFA + backtrace) calls */
  uint32_t n_trivial;      /* trivial halves (plen==0 or tlen==0) */
  uint32_t max_level;      /* deepest recursion level */
  uint32_t max_width;      /* widest computed wavefront */
  uint32_t overlap_rows;   /* row pairs scanned by the overlap search */
} awo_stats_t;

/* Status codes (0 = completed; mirrors AlignmentStatus::Completed, alignment.rs:233-258). */
#define AWO_OK 0
#define AWO_ERR_PENALTIES (-1)
#define AWO_ERR_INTERNAL (-2)
#define AWO_ERR_CAPACITY (-3)

/* Opaque per-thread aligner (mirrors one lib_wfa2 AffineWavefronts; alignment.rs:11-22). */
typedef struct awo_aligner awo_aligner_t;
awo_aligner_t* awo_aligner_new(const awo_penalties_t* pen);
void awo_aligner_delete(awo_aligner_t* a);
/* CPU-baseline mode: skip score pairs in the overlap search whose M-row antidiagonal maxima cannot
 * reach plen + tlen (exact -- results are identical; tests/test_oracle.py checks it).  Off by default
 * so the checker keeps WFA2's plain search. */
void awo_aligner_set_fast_overlap(awo_aligner_t* a, int on);

/*
 * One end-to-end alignment (pattern = query, text = target; alignment.rs:231).
 * cigar_out receives one WFA2-alphabet op byte per column: 'M' match, 'X' mismatch,
 * 'I' consumes text/target, 'D' consumes pattern/query (alignment.rs:331-338).
 * cigar_cap must be >= plen + tlen.  *penalty >= 0; WFA2's score() would be -penalty.
 */
int awo_align(awo_aligner_t* a, const uint8_t* pattern, int plen, const uint8_t* text, int tlen,
              uint8_t* cigar_out, int cigar_cap, int* cigar_len, int* penalty,
              awo_stats_t* stats /* nullable, accumulated */);

/* Plain unidirectional WFA + backtrace on the whole problem (used to cross-check BiWFA). */
int awo_align_unidirectional(awo_aligner_t* a, const uint8_t* pattern, int plen,
                             const uint8_t* text, int tlen, uint8_t* cigar_out, int cigar_cap,
                             int* cigar_len, int* penalty, awo_stats_t* stats);

/* ---- oracle/gotoh.c : independent full-DP optimum (score only, O(plen*tlen)) ---- */
int64_t awo_gotoh_penalty(const uint8_t* pattern, int plen, const uint8_t* text, int tlen,
                          const awo_penalties_t* pen);

/* ---- oracle/cigar_check.c : restates wfa.rs:105-176 + validation_simple.rs:73-161 ---- */
/* Returns 0 when the op bytes consume exactly both sequences, every 'M' column really
 * matches and every 'X' column really differs; negative otherwise.  *rescored gets the
 * penalty of the CIGAR under pen. */
int awo_cigar_check(const uint8_t* cigar, int n, const uint8_t* pattern, int plen,
                    const uint8_t* text, int tlen, const awo_penalties_t* pen, int64_t* rescored);

/* FNV-1a over op bytes (the per-pair hash of awo_all_pairs) */
uint64_t awo_fnv1a(const uint8_t* p, int64_t n);

/* ---- oracle/allpairs_cpu.c : rayon-like all-pairs driver (iterator.rs:222-233) ---- */
typedef struct {
  int32_t status, penalty, cigar_len;
  int32_t num_matches, num_mismatches, num_ins_text, num_del_pattern; /* #M #X #I #D (WFA2 letters) */
  uint64_t cigar_hash; /* FNV-1a over the op bytes */
} awo_pair_result_t;

/* Aligns pairs[i] = (q_idx, t_idx) over sequences stored concatenated (offsets[n+1]).
 * Returns wall seconds; results[npairs].  If paf_sink_bytes != NULL each PAF line is formatted
 * (lib.rs:71-112 layout) into a per-thread memory sink and the total byte count is returned. */
double awo_all_pairs(const uint8_t* seqs, const uint64_t* offsets, int nseq,
                     const int32_t* pairs, int64_t npairs, const awo_penalties_t* pen,
                     int nthreads, awo_pair_result_t* results, awo_stats_t* stats_total,
                     uint64_t* paf_sink_bytes);
/* same, with the exact overlap pre-filter enabled in every worker's aligner */
double awo_all_pairs_fast(const uint8_t* seqs, const uint64_t* offsets, int nseq,
                          const int32_t* pairs, int64_t npairs, const awo_penalties_t* pen,
                          int nthreads, awo_pair_result_t* results, awo_stats_t* stats_total,
                          uint64_t* paf_sink_bytes);

#ifdef __cplusplus
}
#endif
#endif
