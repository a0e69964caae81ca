// allwave.hpp -- C++ host-side mirror of allwave's API surface for the per-pair hot path,
// implemented over the C ABI of include/allwave_hip.h (the reference's own host language, Rust,
// is not available in this environment; INTEGRATION.md shows the Rust-side binding instead).
//
// Same names, argument meaning and error behaviour as the reference so the tests read like the
// reference's own (file:line relative to /root/reference):
//   Sequence, AlignmentResult, AlignmentParams, AlignmentMode, AlignmentError   src/types.rs:7-131
//   parse_scores, alignment_to_paf                                               src/lib.rs:71-153
//   cigar_bytes_to_string, reverse_complement, align_pair's result mapping       src/alignment.rs:25-66,178-190,347-376
//   AllPairIterator (-p none enumeration, WFA orientation, callback streaming)   src/iterator.rs:12-253
//   wfa::align_sequences / validate_cigar_alignment                              src/wfa.rs:105-258
// Pair planning (mash orientation, sparsifiers, kNN/tree pairs) lives in planner.hpp; the CLI in main.cpp.
#pragma once

#include <cstdint>
#include <functional>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "allwave_hip.h"

namespace allwave {

struct Sequence {  // types.rs:7-10
  std::string id;
  std::vector<uint8_t> seq;
};

struct AlignmentResult {  // types.rs:14-33
  size_t query_idx = 0, target_idx = 0;
  size_t query_start = 0, query_end = 0, target_start = 0, target_end = 0;
  bool is_reverse = false;
  std::vector<uint8_t> cigar_bytes;  // raw WFA2-alphabet op bytes
  int32_t score = 0;                 // WFA2 score (= -penalty); i32::MAX on failure
  size_t num_matches = 0;
  size_t alignment_length = 0;       // #M + #X
};

struct AlignmentParams {  // types.rs:37-74
  int32_t match_score = 0, mismatch_penalty = 5, gap_open = 8, gap_extend = 2;
  std::optional<int32_t> gap2_open = 24, gap2_extend = 1;
  std::optional<double> max_divergence;
  static AlignmentParams edit_distance();
  bool operator==(const AlignmentParams& o) const;
};

enum class AlignmentMode { EditDistance, SinglePieceAffine, TwoPieceAffine };  // types.rs:99-117
AlignmentMode alignment_mode_from_params(const AlignmentParams& p);
awv_penalties to_penalties(const AlignmentParams& p);  // create_wfa_aligner, alignment.rs:263-289

struct AlignmentError : std::runtime_error {  // types.rs:121-131
  using std::runtime_error::runtime_error;
};

struct SparsificationStrategy {  // types.rs:78-95
  enum Kind { None, Random, Auto, Connectivity, TreeSampling } kind = None;
  double value = 0.0;                     // Random(frac) / Connectivity(prob)
  size_t k_nearest = 0, k_farthest = 0;   // TreeSampling(k_nearest, k_farthest, random_fraction, kmer_size)
  double random_fraction = 0.0;
  std::optional<size_t> kmer_size;
  // the CLI's -p syntax (main.rs:136-203): none | auto | random:<f> | giant:<p> | connectivity:<p> | tree:<n>:<f>:<r>[:<k>]
  static SparsificationStrategy parse(const std::string& s);  // throws std::invalid_argument with main.rs's messages
};

// lib.rs:116-153 -- throws std::invalid_argument with the reference's messages
AlignmentParams parse_scores(const std::string& scores_str);
// alignment.rs:347-376
std::string cigar_bytes_to_string(const uint8_t* ops, size_t n);
inline std::string cigar_bytes_to_string(const std::vector<uint8_t>& v) { return cigar_bytes_to_string(v.data(), v.size()); }
// alignment.rs:178-190
std::vector<uint8_t> reverse_complement(const std::vector<uint8_t>& seq);
// lib.rs:71-112
std::string alignment_to_paf(const AlignmentResult& r, const std::vector<Sequence>& sequences);
void append_paf(std::string& out, const AlignmentResult& r, const uint8_t* ops, size_t nops,
                const std::vector<Sequence>& sequences);

// Orientation of the query before the final alignment (alignment.rs:35-39).
enum class Orientation {
  Wfa,          // determine_orientation_wfa (alignment.rs:157-175): what AllPairIterator::new uses
  ForwardOnly,  // extension: skip orientation (all '+'); used when strands are known
  Mash          // determine_orientation_mash (alignment.rs:69-154), hoisted to per-sequence sketches
};

using Callback = std::function<void(AlignmentResult&&)>;  // may throw: first error aborts the run

// awv_engine_config.flags (AWV_F_*) for the per-device engines this library creates from now on
// (an engine lives for the rest of the process once created).
void set_engine_flags(int flags);
void set_engine_first_row_cols(int cols);  // awv_engine_config.first_row_cols, likewise (diagnostic / test hook)
// destroys the per-device engines this library holds (their HBM arenas are freed; the next run creates fresh ones with the
// flags then in force)
void release_engines();

class AllPairParallelIterator;

class AllPairIterator {  // iterator.rs:12-171
 public:
  AllPairIterator(const std::vector<Sequence>& sequences, AlignmentParams params);  // ::new
  static AllPairIterator with_options(const std::vector<Sequence>& sequences, AlignmentParams params,
                                      bool exclude_self, bool use_mash_orientation, SparsificationStrategy s);
  AllPairIterator& with_orientation_params(AlignmentParams p);
  // iterator.rs:101-110: a NEW iterator over the same sequences / params / exclude_self / orientation choice with the pair
  // list planned again under `strategy` -- through with_options, exactly like the reference, so the orientation params go
  // back to their default (edit_distance) there too; the device and thread settings (this build's extensions) carry over
  AllPairIterator with_sparsification(SparsificationStrategy strategy) const;
  // iterator.rs:113-125: the batch-parallel consumer (rayon's par_iter in the reference)
  AllPairParallelIterator into_par_iter() const;
  // iterator.rs:151-171 (`impl Iterator`): the next pair's alignment, in pair-list order; std::nullopt at the end.  A per-pair
  // call cannot feed a GPU, so the results are buffered: an empty buffer aligns the next `next_chunk` pairs in one engine call.
  std::optional<AlignmentResult> next();
  AllPairIterator& with_next_chunk(size_t pairs_per_engine_call);
  // host threads for sketching / orientation / formatting of THIS iterator's runs (the reference's `-t`; 0 = the
  // process-wide planner::host_threads())
  AllPairIterator& with_threads(int host_threads);
  AllPairIterator& with_orientation(Orientation o);
  AllPairIterator& with_device(int device);
  // keep pairs rank, rank + world, ... of the planned list (one process per GPU: every pair lands on
  // exactly one rank, per-rank cost stays even for a row-major all-pairs list)
  AllPairIterator& with_shard(size_t rank, size_t world);
  size_t pair_count() const { return pairs_.size(); }
  const std::vector<std::pair<size_t, size_t>>& get_pairs() const { return pairs_; }
  // iterator.rs:127-137,206-253: streams results (order unspecified in the reference for T>1;
  // here batches arrive in pair order)
  void for_each_with_callback(const Callback& cb);
  // formats every record with alignment_to_paf on a small thread pool and hands whole batches to
  // `sink` (replaces the single unbuffered writer thread of src/main.rs:347-367)
  void for_each_paf_batch(const std::function<void(const std::string&)>& sink, int format_threads = 8);
  awv_stats last_stats() const { return stats_; }

 private:
  friend class AllPairParallelIterator;
  void run(const std::function<void(int64_t first, int64_t n, const awv_result* res, const uint8_t* arena,
                                    const std::vector<uint8_t>& is_rev)>& batch_cb) { run_range(0, pairs_.size(), batch_cb); }
  // pairs_[first, first + count) through one orientation pass + one awv_align_pairs call; batch indices are relative to `first`
  void run_range(size_t first, size_t count,
                 const std::function<void(int64_t first, int64_t n, const awv_result* res, const uint8_t* arena,
                                          const std::vector<uint8_t>& is_rev)>& batch_cb);
  const std::vector<Sequence>& sequences_;
  AlignmentParams params_, orientation_params_;
  bool exclude_self_ = true;
  Orientation orientation_ = Orientation::Wfa;
  int device_ = 0;
  int threads_ = 0;
  std::vector<std::pair<size_t, size_t>> pairs_;
  awv_stats stats_{};
  // sequential iteration (next): position in the pair list and the buffered results of the running chunk
  size_t next_pos_ = 0, next_chunk_ = 16384;
  std::vector<AlignmentResult> next_buf_;
  size_t next_buf_pos_ = 0;
};

// iterator.rs:174-253.  The reference's parallel iterator maps align_pair over the pairs on rayon's workers and hands every
// result to a consumer that runs on those workers concurrently; here the pairs go through the engine in batches and every
// batch's results are handed to `callback` from `threads` host threads at once (same contract: the callback must be
// thread-safe; the first error it throws wins, stops the run and is rethrown -- iterator.rs:220-251).
class AllPairParallelIterator {
 public:
  size_t pair_count() const { return it_.pair_count(); }
  AllPairParallelIterator& with_threads(int threads) { threads_ = threads; return *this; }
  void for_each_with_callback(const Callback& cb);   // iterator.rs:206-253
  std::vector<AlignmentResult> collect();           // rayon's collect() on the parallel iterator: results in pair-list order
 private:
  friend class AllPairIterator;
  explicit AllPairParallelIterator(const AllPairIterator& it) : it_(it) {}
  AllPairIterator it_;
  int threads_ = 0;  // 0 = planner::host_threads()
};

// lib.rs:57-68: AllPairIterator::with_options(sequences, params, exclude_self = true, mash orientation = true, sparsification)
// .for_each_with_callback(callback); the callback may throw (first error aborts and is rethrown)
void process_alignments_with_callback(const std::vector<Sequence>& sequences, AlignmentParams params,
                                      SparsificationStrategy sparsification, const Callback& callback);

namespace wfa {  // src/wfa.rs
struct Penalties { int32_t mismatch, gap_opening1, gap_extension1, gap_opening2, gap_extension2; };
struct AlignmentResult {
  int32_t score;
  std::string cigar;
  size_t matches, mismatches, insertions, deletions, alignment_length;
};
// wfa.rs:105-176 -- returns "" when valid, else the reference's message
std::string validate_cigar_alignment(const uint8_t* cigar, size_t n, size_t query_len, size_t reference_len);
// wfa.rs:178-258 (fresh aligner per call in the reference; here one engine call)
AlignmentResult align_sequences(const std::vector<uint8_t>& pattern, const std::vector<uint8_t>& text,
                                const Penalties& p, AlignmentMode mode, int device = 0);
}  // namespace wfa

}  // namespace allwave
