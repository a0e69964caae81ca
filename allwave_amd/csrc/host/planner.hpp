// planner.hpp -- host-side pair planning around the hot path:
//   mash orientation      /root/reference/src/alignment.rs:69-154
//   hash sparsifiers      /root/reference/src/iterator.rs:256-334
//   mash sketches/matrix  /root/reference/src/mash.rs:78-184
//   kNN / stranger pairs  /root/reference/src/knn_graph.rs:12-174
// All hashing is Rust's std DefaultHasher = SipHash-1-3 with zero keys: a `[u8]` hashes its length as
// a little-endian u64 followed by the bytes, a `str` hashes its bytes followed by 0xFF.
#pragma once

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "allwave.hpp"

namespace allwave {
namespace planner {

uint64_t siphash(const uint8_t* data, size_t n, uint64_t k0, uint64_t k1, int c_rounds, int d_rounds);
uint64_t default_hash_bytes(const uint8_t* p, size_t n);  // impl Hash for [u8] through DefaultHasher
uint64_t default_hash_str(const std::string& s);          // impl Hash for str through DefaultHasher

// alignment.rs:96-122: all k-mers without non-ACGT (case-insensitive), hashed as raw bytes, sorted,
// truncated to sketch_size -- then deduplicated (the reference compares them as HashSets)
std::vector<uint64_t> sketch_sequence_stranded(const std::vector<uint8_t>& seq, size_t k, size_t sketch_size);
// mash.rs:78-107: canonical k-mer = min(hash(kmer), hash(reverse_complement_kmer))
std::vector<uint64_t> sketch_sequence_canonical(const std::vector<uint8_t>& seq, size_t k, size_t sketch_size);
double jaccard(const std::vector<uint64_t>& a, const std::vector<uint64_t>& b);  // sorted unique inputs
double mash_distance(double jaccard, size_t k);                                   // mash.rs:59-74

// alignment.rs:69-94 hoisted to per-sequence sketches (identical results, 2 sketches per sequence
// instead of 3 per pair): is_reverse[i] for pairs[i]
std::vector<uint8_t> orient_pairs_mash(const std::vector<Sequence>& seqs,
                                       const std::vector<std::pair<size_t, size_t>>& pairs, int threads);

// host threads used by sketching / orientation when the caller does not say (the CLI's -t; default 8)
void set_host_threads(int threads);
int host_threads();
std::vector<std::vector<double>> compute_distance_matrix(const std::vector<Sequence>& seqs, size_t k, size_t sketch_size);
std::string format_distance_matrix(const std::vector<Sequence>& seqs, const std::vector<std::vector<double>>& m);

double compute_connectivity_probability(size_t n, double connectivity_prob);  // iterator.rs:300-334
std::vector<std::pair<size_t, size_t>> apply_random_sparsification(std::vector<std::pair<size_t, size_t>> pairs,
                                                                   double keep_fraction, const std::vector<Sequence>& seqs);
std::vector<std::pair<size_t, size_t>> build_knn_graph(const std::vector<std::vector<double>>& d, size_t k, bool farthest);
std::vector<std::pair<size_t, size_t>> extract_tree_pairs(const std::vector<Sequence>& seqs, size_t k_nearest,
                                                          size_t k_farthest, double random_fraction, size_t kmer_size);

// ---- multi-GPU shards (what rayon's work stealing does for the reference, iterator.rs:222-233)
// Predicted cost of one pair: cell-steps grow with the square of the optimal penalty, estimated from
// the lengths -- a mismatch share proportional to the shorter sequence plus the gap a length
// difference forces (priced with the penalties in use).  Only ratios matter.
double predicted_pair_cost(size_t qlen, size_t tlen, const AlignmentParams& params);
// Longest-processing-time-first assignment of pairs to `world` shards: pairs in descending predicted
// cost (ties: list order), each to the currently lightest shard (ties: lowest rank).  Returns the
// shard of every pair; deterministic, so every process derives the same partition.  A list whose
// pairs all cost the same degenerates to the strided shard r, r + world, ...
std::vector<uint32_t> assign_shards_lpt(const std::vector<double>& cost, size_t world);

}  // namespace planner
}  // namespace allwave
