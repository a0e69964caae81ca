// capi.cpp -- extern "C" hooks over the C++ host mirror so the Python tests and bench.py can drive
// it through ctypes (plain pointers and sizes only).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "allwave.hpp"
#include "planner.hpp"

using namespace allwave;

namespace {
void set_err(char* err, size_t cap, const std::string& m) {
  if (err && cap) snprintf(err, cap, "%s", m.c_str());
}
std::vector<Sequence> make_seqs(int n, const char* const* ids, const uint8_t* bytes, const uint64_t* offs) {
  std::vector<Sequence> s((size_t)n);
  for (int i = 0; i < n; ++i) {
    s[i].id = ids[i];
    s[i].seq.assign(bytes + offs[i], bytes + offs[i + 1]);
  }
  return s;
}
}  // namespace

extern "C" {

int awh_parse_scores(const char* s, int32_t out[6], int* n, char* err, size_t cap) {
  try {
    const AlignmentParams p = parse_scores(s);
    out[0] = p.match_score; out[1] = p.mismatch_penalty; out[2] = p.gap_open; out[3] = p.gap_extend;
    *n = 4;
    if (p.gap2_open && p.gap2_extend) { out[4] = *p.gap2_open; out[5] = *p.gap2_extend; *n = 6; }
    const awv_penalties q = to_penalties(p);
    (void)q;
    return 0;
  } catch (const std::exception& e) { set_err(err, cap, e.what()); return -1; }
}

// mode of AlignmentMode::from_params: 0 edit, 1 single-piece, 2 two-piece; pen = penalties handed to the engine
int awh_mode_from_scores(const char* s, int* mode, int32_t pen[7], char* err, size_t cap) {
  try {
    const AlignmentParams p = parse_scores(s);
    *mode = (int)alignment_mode_from_params(p);
    const awv_penalties q = to_penalties(p);
    pen[0] = q.match; pen[1] = q.mismatch; pen[2] = q.gap_open1; pen[3] = q.gap_ext1; pen[4] = q.gap_open2; pen[5] = q.gap_ext2; pen[6] = q.two_piece;
    return 0;
  } catch (const std::exception& e) { set_err(err, cap, e.what()); return -1; }
}

int awh_cigar_to_string(const uint8_t* ops, size_t n, char* out, size_t cap) {
  const std::string s = cigar_bytes_to_string(ops, n);
  if (s.size() + 1 > cap) return -1;
  memcpy(out, s.c_str(), s.size() + 1);
  return (int)s.size();
}

int awh_reverse_complement(const uint8_t* in, size_t n, uint8_t* out) {
  const std::vector<uint8_t> r = reverse_complement(std::vector<uint8_t>(in, in + n));
  if (n) memcpy(out, r.data(), n);
  return 0;
}

// Formats one record from explicit fields (unit-tests alignment_to_paf without a GPU).
int awh_format_paf(const char* qid, size_t qlen, const char* tid, size_t tlen, size_t qs, size_t qe, size_t ts, size_t te,
                   int is_reverse, size_t num_matches, size_t alignment_length, const uint8_t* ops, size_t nops, char* out,
                   size_t cap) {
  std::vector<Sequence> seqs(2);
  seqs[0].id = qid; seqs[0].seq.assign(qlen, 'A');
  seqs[1].id = tid; seqs[1].seq.assign(tlen, 'A');
  AlignmentResult r;
  r.query_idx = 0; r.target_idx = 1; r.query_start = qs; r.query_end = qe; r.target_start = ts; r.target_end = te;
  r.is_reverse = is_reverse != 0; r.num_matches = num_matches; r.alignment_length = alignment_length;
  r.cigar_bytes.assign(ops, ops + nops);
  const std::string s = alignment_to_paf(r, seqs);
  if (s.size() + 1 > cap) return -1;
  memcpy(out, s.c_str(), s.size() + 1);
  return (int)s.size();
}

int awh_all_pairs_paf(int n, const char* const* ids, const uint8_t* bytes, const uint64_t* offs, const char* scores,
                      const char* sparsification, int orientation, int exclude_self, int device, char** out, size_t* out_len,
                      char* err, size_t cap) {
  try {
    const std::vector<Sequence> seqs = make_seqs(n, ids, bytes, offs);
    AllPairIterator it = AllPairIterator::with_options(seqs, parse_scores(scores), exclude_self != 0, orientation == 2,
                                                      SparsificationStrategy::parse(sparsification ? sparsification : "none"));
    it.with_orientation(orientation == 0 ? Orientation::ForwardOnly : orientation == 1 ? Orientation::Wfa : Orientation::Mash);
    it.with_device(device);
    std::string all;
    it.for_each_with_callback([&](AlignmentResult&& r) {  // the reference's own per-record path
      all += alignment_to_paf(r, seqs);
      all.push_back('\n');
    });
    *out = (char*)malloc(all.size() + 1);
    memcpy(*out, all.c_str(), all.size() + 1);
    *out_len = all.size();
    return 0;
  } catch (const std::exception& e) { set_err(err, cap, e.what()); return -1; }
}

// Every consumer of the pair list the reference offers, one per `mode` (tests/test_host_api.py):
//   0  AllPairIterator::for_each_with_callback                 iterator.rs:127-137
//   1  the sequential `impl Iterator` (next() until the end)    iterator.rs:151-171
//   2  into_par_iter().for_each_with_callback on `threads`      iterator.rs:113-125,206-253
//   3  into_par_iter().collect()                                iterator.rs:182-203 (rayon collect)
//   4  process_alignments_with_callback                         lib.rs:57-68 (mash orientation, exclude_self)
// `resparsify`: plan with -p none first, then call with_sparsification(strategy) (iterator.rs:101-110).
// `fail_at` >= 0: the callback throws at its fail_at-th record; the call must fail with that message (first error wins).
// out = PAF lines in the order the records arrived.
int awh_iterate(int n, const char* const* ids, const uint8_t* bytes, const uint64_t* offs, const char* scores,
                const char* sparsification, int orientation, int mode, int threads, int chunk, int resparsify, long fail_at,
                int device, char** out, size_t* out_len, size_t* n_records, char* err, size_t cap) {
  try {
    const std::vector<Sequence> seqs = make_seqs(n, ids, bytes, offs);
    const SparsificationStrategy strat = SparsificationStrategy::parse(sparsification ? sparsification : "none");
    const Orientation orient = orientation == 0 ? Orientation::ForwardOnly : orientation == 1 ? Orientation::Wfa : Orientation::Mash;
    std::mutex mu;
    std::string all;
    size_t seen = 0;
    auto record = [&](AlignmentResult&& r) {
      std::string line = alignment_to_paf(r, seqs);
      std::lock_guard<std::mutex> g(mu);
      if (fail_at >= 0 && (long)seen == fail_at) throw std::runtime_error("callback failed at record " + std::to_string(seen));
      ++seen;
      all += line;
      all.push_back('\n');
    };
    if (mode == 4) {
      process_alignments_with_callback(seqs, parse_scores(scores), strat, record);
    } else {
      AllPairIterator it0 = AllPairIterator::with_options(seqs, parse_scores(scores), true, orientation == 2,
                                                         resparsify ? SparsificationStrategy{} : strat);
      it0.with_orientation(orient).with_device(device);
      if (chunk > 0) it0.with_next_chunk((size_t)chunk);
      AllPairIterator it = resparsify ? it0.with_sparsification(strat) : it0;
      if (mode == 0) it.for_each_with_callback(record);
      else if (mode == 1) { while (auto r = it.next()) record(std::move(*r)); }
      else if (mode == 2) it.into_par_iter().with_threads(threads).for_each_with_callback(record);
      else if (mode == 3) { for (auto& r : it.into_par_iter().collect()) record(std::move(r)); }
      else throw std::invalid_argument("awh_iterate: unknown mode");
    }
    *out = (char*)malloc(all.size() + 1);
    memcpy(*out, all.c_str(), all.size() + 1);
    *out_len = all.size();
    *n_records = seen;
    return 0;
  } catch (const std::exception& e) { set_err(err, cap, e.what()); return -1; }
}

// End-to-end measurement: sequences -> GPU alignment -> D2H -> PAF text into a counting sink.
int awh_all_pairs_paf_count(int n, const char* const* ids, const uint8_t* bytes, const uint64_t* offs, const char* scores,
                            int orientation, int device, int format_threads, uint64_t* out_bytes, uint64_t* out_lines,
                            double* secs, awv_stats* st, char* err, size_t cap) {
  try {
    const std::vector<Sequence> seqs = make_seqs(n, ids, bytes, offs);
    AllPairIterator it(seqs, parse_scores(scores));
    it.with_orientation(orientation == 0 ? Orientation::ForwardOnly : orientation == 1 ? Orientation::Wfa : Orientation::Mash).with_device(device);
    it.with_threads(format_threads);  // this call's sketching / orientation threads: carried by the iterator, not a process-wide setting
    uint64_t nb = 0, nl = 0;
    const auto t0 = std::chrono::steady_clock::now();
    it.for_each_paf_batch([&](const std::string& s) {
      nb += s.size();
      for (char c : s) nl += c == '\n';
    }, format_threads);
    *secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    *out_bytes = nb;
    *out_lines = nl;
    if (st) *st = it.last_stats();
    return 0;
  } catch (const std::exception& e) { set_err(err, cap, e.what()); return -1; }
}

int awh_align_sequences(const uint8_t* pattern, size_t plen, const uint8_t* text, size_t tlen, const int32_t pen[5], int mode,
                        int device, int32_t* score, char* cigar, size_t ccap, uint64_t counts[5], char* err, size_t cap) {
  try {
    const wfa::Penalties p{pen[0], pen[1], pen[2], pen[3], pen[4]};
    const wfa::AlignmentResult r = wfa::align_sequences(std::vector<uint8_t>(pattern, pattern + plen),
                                                        std::vector<uint8_t>(text, text + tlen), p, (AlignmentMode)mode, device);
    *score = r.score;
    if (r.cigar.size() + 1 > ccap) { set_err(err, cap, "cigar buffer too small"); return -2; }
    memcpy(cigar, r.cigar.c_str(), r.cigar.size() + 1);
    counts[0] = r.matches; counts[1] = r.mismatches; counts[2] = r.insertions; counts[3] = r.deletions; counts[4] = r.alignment_length;
    return 0;
  } catch (const std::exception& e) { set_err(err, cap, e.what()); return -1; }
}

int awh_validate_cigar(const uint8_t* cigar, size_t n, size_t qlen, size_t rlen, char* err, size_t cap) {
  const std::string m = wfa::validate_cigar_alignment(cigar, n, qlen, rlen);
  if (m.empty()) return 0;
  set_err(err, cap, m);
  return -1;
}

// ---- planner hooks (no GPU needed) ----
uint64_t awh_siphash(const uint8_t* p, size_t n, uint64_t k0, uint64_t k1, int c, int d) { return planner::siphash(p, n, k0, k1, c, d); }
uint64_t awh_hash_bytes(const uint8_t* p, size_t n) { return planner::default_hash_bytes(p, n); }
uint64_t awh_hash_str(const char* s) { return planner::default_hash_str(s); }
double awh_connectivity_probability(size_t n, double x) { return planner::compute_connectivity_probability(n, x); }

// pair list of AllPairIterator::with_options(..., strategy) without aligning: out = malloc'ed (i, j) int64 pairs
int awh_plan_pairs(int n, const char* const* ids, const uint8_t* bytes, const uint64_t* offs, const char* sparsification,
                   int exclude_self, int64_t** out, size_t* npairs, char* err, size_t cap) {
  try {
    const std::vector<Sequence> seqs = make_seqs(n, ids, bytes, offs);
    // exclude_self & 2: plan with -p none first, then with_sparsification(strategy) (iterator.rs:101-110)
    const bool resparsify = (exclude_self & 2) != 0;
    AllPairIterator it0 = AllPairIterator::with_options(seqs, AlignmentParams{}, (exclude_self & 1) != 0, false,
                                                       resparsify ? SparsificationStrategy{} : SparsificationStrategy::parse(sparsification));
    AllPairIterator it = resparsify ? it0.with_sparsification(SparsificationStrategy::parse(sparsification)) : it0;
    const auto& p = it.get_pairs();
    *out = (int64_t*)malloc(sizeof(int64_t) * 2 * (p.size() + 1));
    for (size_t i = 0; i < p.size(); ++i) { (*out)[2 * i] = (int64_t)p[i].first; (*out)[2 * i + 1] = (int64_t)p[i].second; }
    *npairs = p.size();
    return 0;
  } catch (const std::exception& e) { set_err(err, cap, e.what()); return -1; }
}

int awh_knn_graph(const double* dist, int n, int k, int farthest, int64_t** out, size_t* npairs) {
  std::vector<std::vector<double>> d((size_t)n, std::vector<double>((size_t)n));
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) d[i][j] = dist[i * n + j];
  const auto p = planner::build_knn_graph(d, (size_t)k, farthest != 0);
  *out = (int64_t*)malloc(sizeof(int64_t) * 2 * (p.size() + 1));
  for (size_t i = 0; i < p.size(); ++i) { (*out)[2 * i] = (int64_t)p[i].first; (*out)[2 * i + 1] = (int64_t)p[i].second; }
  *npairs = p.size();
  return 0;
}

int awh_mash_matrix(int n, const char* const* ids, const uint8_t* bytes, const uint64_t* offs, int k, double* out) {
  const std::vector<Sequence> seqs = make_seqs(n, ids, bytes, offs);
  const auto m = planner::compute_distance_matrix(seqs, (size_t)k, 1000);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) out[i * n + j] = m[i][j];
  return 0;
}

int awh_orient_mash(int n, const char* const* ids, const uint8_t* bytes, const uint64_t* offs, const int64_t* pairs,
                    size_t npairs, uint8_t* is_rev) {
  const std::vector<Sequence> seqs = make_seqs(n, ids, bytes, offs);
  std::vector<std::pair<size_t, size_t>> p(npairs);
  for (size_t i = 0; i < npairs; ++i) p[i] = {(size_t)pairs[2 * i], (size_t)pairs[2 * i + 1]};
  const auto r = planner::orient_pairs_mash(seqs, p, 8);
  memcpy(is_rev, r.data(), npairs);
  return 0;
}

// shard of every pair under the cost-balanced (LPT) partition AllPairIterator::with_shard uses; cost_out (nullable)
// receives the predicted costs
int awh_shard_pairs(const int64_t* pairs, size_t npairs, const int64_t* lens, size_t nseq, const char* scores, size_t world, uint32_t* shard_out,
                    double* cost_out, char* err, size_t cap) {
  try {
    const AlignmentParams p = parse_scores(scores);
    if (world == 0) throw std::invalid_argument("shard_pairs: world must be at least 1");
    std::vector<double> cost(npairs);
    for (size_t i = 0; i < npairs; ++i) {
      const int64_t a = pairs[2 * i], b = pairs[2 * i + 1];
      if (a < 0 || b < 0 || (uint64_t)a >= nseq || (uint64_t)b >= nseq)
        throw std::invalid_argument("shard_pairs: pair " + std::to_string(i) + " names a sequence outside [0, " + std::to_string(nseq) + ")");
      cost[i] = planner::predicted_pair_cost((size_t)lens[a], (size_t)lens[b], p);
    }
    const std::vector<uint32_t> sh = planner::assign_shards_lpt(cost, world);
    for (size_t i = 0; i < npairs; ++i) shard_out[i] = sh[i];
    if (cost_out) for (size_t i = 0; i < npairs; ++i) cost_out[i] = cost[i];
    return 0;
  } catch (const std::exception& e) { set_err(err, cap, e.what()); return -1; }
}

// engine configuration of the engines the host library creates from now on; release = destroy the cached ones first
void awh_set_engine_config(int flags, int first_row_cols, int release) {
  if (release) release_engines();
  set_engine_flags(flags);
  set_engine_first_row_cols(first_row_cols);
}

void awh_free(void* p) { free(p); }

}  // extern "C"
