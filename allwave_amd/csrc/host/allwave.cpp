// allwave.cpp -- host-side mirror of allwave's API over the C ABI (see allwave.hpp).
#include "allwave.hpp"
#include "planner.hpp"

#include <algorithm>
#include <atomic>
#include <climits>
#include <cstdio>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <chrono>
#include <thread>

namespace allwave {

// ---- types.rs -------------------------------------------------------------------------------
AlignmentParams AlignmentParams::edit_distance() {  // types.rs:62-73
  AlignmentParams p;
  p.match_score = 0;
  p.mismatch_penalty = 1;
  p.gap_open = 1;
  p.gap_extend = 1;
  p.gap2_open.reset();
  p.gap2_extend.reset();
  p.max_divergence.reset();
  return p;
}

bool AlignmentParams::operator==(const AlignmentParams& o) const {
  return match_score == o.match_score && mismatch_penalty == o.mismatch_penalty && gap_open == o.gap_open &&
         gap_extend == o.gap_extend && gap2_open == o.gap2_open && gap2_extend == o.gap2_extend &&
         max_divergence == o.max_divergence;
}

AlignmentMode alignment_mode_from_params(const AlignmentParams& p) {  // types.rs:107-116
  if (p.gap2_open.has_value() && p.gap2_extend.has_value()) return AlignmentMode::TwoPieceAffine;
  if (p.gap_open == p.gap_extend && p.gap_open == p.mismatch_penalty) return AlignmentMode::EditDistance;
  return AlignmentMode::SinglePieceAffine;
}

awv_penalties to_penalties(const AlignmentParams& p) {  // alignment.rs:263-289
  awv_penalties q{};
  q.match = p.match_score;
  q.mismatch = p.mismatch_penalty;
  switch (alignment_mode_from_params(p)) {
    case AlignmentMode::EditDistance:  // "edit" is gap-affine (x, x, x)
      q.gap_open1 = p.mismatch_penalty;
      q.gap_ext1 = p.mismatch_penalty;
      break;
    case AlignmentMode::SinglePieceAffine:
      q.gap_open1 = p.gap_open;
      q.gap_ext1 = p.gap_extend;
      break;
    case AlignmentMode::TwoPieceAffine:
      q.gap_open1 = p.gap_open;
      q.gap_ext1 = p.gap_extend;
      q.gap_open2 = p.gap2_open.value_or(p.gap_open);
      q.gap_ext2 = p.gap2_extend.value_or(p.gap_extend);
      q.two_piece = 1;
      break;
  }
  return q;
}

// ---- lib.rs ---------------------------------------------------------------------------------
AlignmentParams parse_scores(const std::string& scores_str) {  // lib.rs:116-153
  std::vector<int32_t> scores;
  size_t pos = 0;
  while (true) {
    size_t comma = scores_str.find(',', pos);
    std::string tok = scores_str.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
    size_t b = tok.find_first_not_of(" \t\n\r"), e = tok.find_last_not_of(" \t\n\r");
    tok = b == std::string::npos ? "" : tok.substr(b, e - b + 1);
    size_t used = 0;
    long v = 0;
    bool ok = !tok.empty();
    if (ok) {
      try { v = std::stol(tok, &used, 10); } catch (...) { ok = false; }
      ok = ok && used == tok.size() && v >= INT32_MIN && v <= INT32_MAX;
    }
    if (!ok) throw std::invalid_argument("Failed to parse scores: invalid digit found in string");
    scores.push_back((int32_t)v);
    if (comma == std::string::npos) break;
    pos = comma + 1;
  }
  AlignmentParams p;
  p.max_divergence.reset();
  if (scores.size() == 4 || scores.size() == 6) {
    p.match_score = scores[0];
    p.mismatch_penalty = scores[1];
    p.gap_open = scores[2];
    p.gap_extend = scores[3];
    if (scores.size() == 6) { p.gap2_open = scores[4]; p.gap2_extend = scores[5]; }
    else { p.gap2_open.reset(); p.gap2_extend.reset(); }
    return p;
  }
  throw std::invalid_argument("Invalid number of scores: " + std::to_string(scores.size()) + ". Expected 4 or 6 values.");
}

static inline void append_uint(std::string& out, size_t v) {
  char buf[24];
  int n = 0;
  do { buf[n++] = (char)('0' + v % 10); v /= 10; } while (v);
  while (n) out.push_back(buf[--n]);
}

static void append_cigar(std::string& out, const uint8_t* ops, size_t n) {  // alignment.rs:347-376
  // Run-length encoding straight into the string's storage (a run costs at most two characters per op byte), runs
  // found eight op bytes at a time: formatting 65,280 CIGARs of 10 kbp alignments was 0.27 s of the 2.1 s end-to-end
  // call when it went byte by byte through push_back.
  const size_t base = out.size();
  out.resize(base + 2 * n + 24);
  char* const w0 = &out[0] + base;
  char* w = w0;
  size_t i = 0;
  while (i < n) {
    const uint8_t op = ops[i];
    const uint64_t pat = 0x0101010101010101ull * op;
    size_t j = i + 1;
    bool ended = false;
    while (j + 8 <= n) {
      uint64_t x;
      memcpy(&x, ops + j, 8);
      x ^= pat;
      if (x) { j += (size_t)(__builtin_ctzll(x) >> 3); ended = true; break; }
      j += 8;
    }
    if (!ended) while (j < n && ops[j] == op) ++j;
    size_t len = j - i;
    char buf[24];
    int k = 0;
    do { buf[k++] = (char)('0' + len % 10); len /= 10; } while (len);
    while (k) *w++ = buf[--k];
    *w++ = op == 'M' ? '=' : op == 'X' ? 'X' : op == 'I' ? 'D' : op == 'D' ? 'I' : '?';
    i = j;
  }
  out.resize(base + (size_t)(w - w0));
}

std::string cigar_bytes_to_string(const uint8_t* ops, size_t n) {
  std::string s;
  append_cigar(s, ops, n);
  return s;
}

std::vector<uint8_t> reverse_complement(const std::vector<uint8_t>& seq) {  // alignment.rs:178-190
  std::vector<uint8_t> out(seq.size());
  for (size_t i = 0; i < seq.size(); ++i) {
    uint8_t b = seq[seq.size() - 1 - i], c;
    switch (b) {
      case 'A': case 'a': c = 'T'; break;
      case 'T': case 't': c = 'A'; break;
      case 'C': case 'c': c = 'G'; break;
      case 'G': case 'g': c = 'C'; break;
      default: c = 'N'; break;
    }
    out[i] = c;
  }
  return out;
}

void append_paf(std::string& out, const AlignmentResult& r, const uint8_t* ops, size_t nops,
                const std::vector<Sequence>& sequences) {  // lib.rs:71-112
  const Sequence& q = sequences[r.query_idx];
  const Sequence& t = sequences[r.target_idx];
  const size_t qa = r.query_end - r.query_start, ta = r.target_end - r.target_start;
  const size_t block_len = std::max(ta, qa);
  const double identity = r.alignment_length > 0 ? (double)r.num_matches / (double)r.alignment_length : 0.0;
  out += q.id; out.push_back('\t');
  append_uint(out, q.seq.size()); out.push_back('\t');
  append_uint(out, r.query_start); out.push_back('\t');
  append_uint(out, r.query_end); out.push_back('\t');
  out.push_back(r.is_reverse ? '-' : '+'); out.push_back('\t');
  out += t.id; out.push_back('\t');
  append_uint(out, t.seq.size()); out.push_back('\t');
  append_uint(out, r.target_start); out.push_back('\t');
  append_uint(out, r.target_end); out.push_back('\t');
  append_uint(out, r.num_matches); out.push_back('\t');
  append_uint(out, block_len); out.push_back('\t');
  out += "60\tgi:f:";
  char buf[32];
  snprintf(buf, sizeof(buf), "%.6f", identity);
  out += buf;
  out += "\tcg:Z:";
  append_cigar(out, ops, nops);
}

std::string alignment_to_paf(const AlignmentResult& r, const std::vector<Sequence>& sequences) {
  std::string s;
  append_paf(s, r, r.cigar_bytes.data(), r.cigar_bytes.size(), sequences);
  return s;
}

// ---- iterator.rs ----------------------------------------------------------------------------
namespace {
// One engine per device and process, created on first use and kept (like the reference's cached
// per-thread aligners, alignment.rs:11-22): its HBM arenas are tens of GB, and allocating them right
// after a free can take the driver seconds.  A holder owns the device's engine for its lifetime.
std::atomic<int> g_engine_flags{0};  // awv_engine_config.flags of engines created from now on (set_engine_flags)
std::atomic<int> g_engine_first_row_cols{0};  // awv_engine_config.first_row_cols, likewise (diagnostic / test hook)
struct EngineTable {
  std::mutex mu;  // guards the table
  std::map<int, std::pair<awv_engine*, std::unique_ptr<std::mutex>>> engines;
};
EngineTable& engine_table() {
  static EngineTable t;
  return t;
}
struct EngineHolder {
  awv_engine* e = nullptr;
  std::unique_lock<std::mutex> lock;
  explicit EngineHolder(int device) {
    EngineTable& tb = engine_table();
    std::mutex* dev_mu;
    {
      std::lock_guard<std::mutex> g(tb.mu);
      auto& slot = tb.engines[device];
      if (!slot.second) slot.second.reset(new std::mutex());
      dev_mu = slot.second.get();
    }
    lock = std::unique_lock<std::mutex>(*dev_mu);  // one run at a time per device
    std::lock_guard<std::mutex> g(tb.mu);
    auto& slot = tb.engines[device];
    if (!slot.first) {
      awv_engine_config cfg{};
      cfg.device = device;
      cfg.flags = g_engine_flags.load();
      cfg.first_row_cols = g_engine_first_row_cols.load();
      if (awv_engine_create(&cfg, &slot.first) != AWV_OK) throw AlignmentError(std::string("engine: ") + awv_last_error());
    }
    e = slot.first;
  }
};

void upload(awv_engine* e, const std::vector<Sequence>& seqs) {
  std::vector<uint64_t> offs(seqs.size() + 1, 0);
  for (size_t i = 0; i < seqs.size(); ++i) offs[i + 1] = offs[i] + seqs[i].seq.size();
  std::vector<uint8_t> cat(offs.back() + 1);
  for (size_t i = 0; i < seqs.size(); ++i)
    if (!seqs[i].seq.empty()) memcpy(cat.data() + offs[i], seqs[i].seq.data(), seqs[i].seq.size());
  if (awv_engine_set_sequences(e, (int32_t)seqs.size(), cat.data(), offs.data()) != AWV_OK)
    throw AlignmentError(std::string("set_sequences: ") + awv_last_error());
}

// align_pair's result mapping (alignment.rs:42-65, 239-253)
AlignmentResult make_result(size_t qi, size_t ti, bool is_rev, const awv_result& r, const uint8_t* arena, bool copy_cigar) {
  AlignmentResult a;
  a.query_idx = qi;
  a.target_idx = ti;
  a.is_reverse = is_rev;
  if (r.status != AWV_ST_COMPLETED) {  // "empty" alignment on failure, still emitted
    a.score = INT32_MAX;
    return a;
  }
  a.query_end = (size_t)r.q_end;
  a.target_end = (size_t)r.t_end;
  a.score = r.score;
  a.num_matches = (size_t)r.num_matches;
  a.alignment_length = (size_t)r.num_matches + (size_t)r.num_mismatches;
  if (copy_cigar && arena) a.cigar_bytes.assign(arena + r.cigar_off, arena + r.cigar_off + r.cigar_len);
  return a;
}
}  // namespace

AllPairIterator::AllPairIterator(const std::vector<Sequence>& sequences, AlignmentParams params)
    : sequences_(sequences), params_(std::move(params)), orientation_params_(AlignmentParams::edit_distance()) {
  const size_t n = sequences.size();
  for (size_t i = 0; i < n; ++i)
    for (size_t j = 0; j < n; ++j)
      if (i != j) pairs_.emplace_back(i, j);  // iterator.rs:38-43 row-major, i != j
}

AllPairIterator AllPairIterator::with_options(const std::vector<Sequence>& sequences, AlignmentParams params,
                                              bool exclude_self, bool use_mash_orientation, SparsificationStrategy s) {
  AllPairIterator it(sequences, std::move(params));
  it.exclude_self_ = exclude_self;
  if (!exclude_self) {  // iterator.rs:44-46
    it.pairs_.clear();
    for (size_t i = 0; i < sequences.size(); ++i)
      for (size_t j = 0; j < sequences.size(); ++j) it.pairs_.emplace_back(i, j);
  }
  switch (s.kind) {  // iterator.rs:49-79
    case SparsificationStrategy::None: break;
    case SparsificationStrategy::Random:
      it.pairs_ = planner::apply_random_sparsification(std::move(it.pairs_), s.value, sequences);
      break;
    case SparsificationStrategy::Auto:
      it.pairs_ = planner::apply_random_sparsification(std::move(it.pairs_),
                                                      planner::compute_connectivity_probability(sequences.size(), 0.95), sequences);
      break;
    case SparsificationStrategy::Connectivity:
      it.pairs_ = planner::apply_random_sparsification(std::move(it.pairs_),
                                                      planner::compute_connectivity_probability(sequences.size(), s.value), sequences);
      break;
    case SparsificationStrategy::TreeSampling:
      it.pairs_ = planner::extract_tree_pairs(sequences, s.k_nearest, s.k_farthest, s.random_fraction, s.kmer_size.value_or(15));
      break;
  }
  it.orientation_ = use_mash_orientation ? Orientation::Mash : Orientation::Wfa;
  return it;
}

AllPairIterator& AllPairIterator::with_orientation_params(AlignmentParams p) { orientation_params_ = std::move(p); return *this; }
AllPairIterator& AllPairIterator::with_orientation(Orientation o) { orientation_ = o; return *this; }
AllPairIterator& AllPairIterator::with_device(int device) { device_ = device; return *this; }
void set_engine_flags(int flags) { g_engine_flags.store(flags); }
void set_engine_first_row_cols(int cols) { g_engine_first_row_cols.store(cols); }
void release_engines() {
  EngineTable& tb = engine_table();
  std::lock_guard<std::mutex> g(tb.mu);
  for (auto& kv : tb.engines) {
    std::lock_guard<std::mutex> busy(*kv.second.second);  // (waits for a run in flight on that device)
    if (kv.second.first) awv_engine_destroy(kv.second.first);
    kv.second.first = nullptr;
  }
}
AllPairIterator& AllPairIterator::with_shard(size_t rank, size_t world) {
  // cost-balanced shards (planner::assign_shards_lpt): every process derives the same partition and
  // keeps its own part, in list order; equal-cost lists (config 2 / 3) come out strided
  if (world <= 1) return *this;
  std::vector<double> cost(pairs_.size());
  for (size_t i = 0; i < pairs_.size(); ++i)
    cost[i] = planner::predicted_pair_cost(sequences_[pairs_[i].first].seq.size(), sequences_[pairs_[i].second].seq.size(), params_);
  const std::vector<uint32_t> shard = planner::assign_shards_lpt(cost, world);
  std::vector<std::pair<size_t, size_t>> mine;
  mine.reserve(pairs_.size() / world + 1);
  for (size_t i = 0; i < pairs_.size(); ++i)
    if (shard[i] == (uint32_t)rank) mine.push_back(pairs_[i]);
  pairs_.swap(mine);
  return *this;
}

AllPairIterator AllPairIterator::with_sparsification(SparsificationStrategy strategy) const {  // iterator.rs:101-110
  AllPairIterator it = with_options(sequences_, params_, exclude_self_, orientation_ == Orientation::Mash, std::move(strategy));
  if (orientation_ == Orientation::ForwardOnly) it.orientation_ = Orientation::ForwardOnly;  // (this build's extension survives)
  it.device_ = device_;
  it.threads_ = threads_;
  it.next_chunk_ = next_chunk_;
  return it;
}
AllPairIterator& AllPairIterator::with_next_chunk(size_t n) { next_chunk_ = std::max<size_t>(1, n); return *this; }
AllPairIterator& AllPairIterator::with_threads(int t) { threads_ = t; return *this; }
AllPairParallelIterator AllPairIterator::into_par_iter() const { return AllPairParallelIterator(*this); }

std::optional<AlignmentResult> AllPairIterator::next() {  // iterator.rs:151-171
  if (next_buf_pos_ >= next_buf_.size()) {
    next_buf_.clear();
    next_buf_pos_ = 0;
    if (next_pos_ >= pairs_.size()) return std::nullopt;
    const size_t first = next_pos_, cnt = std::min(next_chunk_, pairs_.size() - first);
    next_buf_.resize(cnt);
    run_range(first, cnt, [&](int64_t bf, int64_t bn, const awv_result* res, const uint8_t* arena, const std::vector<uint8_t>& rev) {
      for (int64_t i = 0; i < bn; ++i) {
        const auto& pr = pairs_[first + (size_t)(bf + i)];
        next_buf_[(size_t)(bf + i)] = make_result(pr.first, pr.second, rev[(size_t)(bf + i)] != 0, res[i], arena, true);
      }
    });
    next_pos_ += cnt;
  }
  return std::move(next_buf_[next_buf_pos_++]);
}

void AllPairParallelIterator::for_each_with_callback(const Callback& cb) {
  const int want = threads_ > 0 ? threads_ : (it_.threads_ > 0 ? it_.threads_ : planner::host_threads());
  it_.run([&](int64_t first, int64_t cnt, const awv_result* res, const uint8_t* arena, const std::vector<uint8_t>& rev) {
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(want, cnt));
    std::mutex mu;
    std::exception_ptr err;  // the first error wins (iterator.rs:236-242); the other workers stop at their next pair
    std::atomic<bool> stop{false};
    std::atomic<int64_t> cursor{0};
    auto work = [&]() {
      for (;;) {
        const int64_t i = cursor.fetch_add(1);
        if (i >= cnt || stop.load()) return;
        const auto& pr = it_.pairs_[(size_t)(first + i)];
        try {
          cb(make_result(pr.first, pr.second, rev[(size_t)(first + i)] != 0, res[i], arena, true));
        } catch (...) {
          std::lock_guard<std::mutex> g(mu);
          if (!err) err = std::current_exception();
          stop.store(true);
          return;
        }
      }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
    if (err) std::rethrow_exception(err);
  });
}

std::vector<AlignmentResult> AllPairParallelIterator::collect() {
  std::vector<AlignmentResult> out(it_.pairs_.size());
  it_.run([&](int64_t first, int64_t cnt, const awv_result* res, const uint8_t* arena, const std::vector<uint8_t>& rev) {
    for (int64_t i = 0; i < cnt; ++i) {
      const auto& pr = it_.pairs_[(size_t)(first + i)];
      out[(size_t)(first + i)] = make_result(pr.first, pr.second, rev[(size_t)(first + i)] != 0, res[i], arena, true);
    }
  });
  return out;
}

void process_alignments_with_callback(const std::vector<Sequence>& sequences, AlignmentParams params,
                                      SparsificationStrategy sparsification, const Callback& callback) {  // lib.rs:57-68
  AllPairIterator aligner = AllPairIterator::with_options(sequences, std::move(params), true, true, std::move(sparsification));
  aligner.for_each_with_callback(callback);
}

void AllPairIterator::run_range(size_t range_first, size_t range_count,
                                const std::function<void(int64_t, int64_t, const awv_result*, const uint8_t*,
                                                         const std::vector<uint8_t>&)>& batch_cb) {
#ifdef AWV_DEBUG_KNOBS
  const bool timing = getenv("AWH_TIMING") != nullptr;  // diagnostic: stage times on stderr
#else
  const bool timing = false;
#endif
  const auto tr0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (timing) fprintf(stderr, "[awh] %-18s %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - tr0).count());
  };
  EngineHolder eh(device_);
  lap("engine created");
  upload(eh.e, sequences_);
  lap("sequences uploaded");
  if (range_first > pairs_.size() || range_count > pairs_.size() - range_first) throw AlignmentError("pair range out of bounds");
  // (a sub-range -- sequential next() -- works on its own copy of the slice; the whole list is used in place)
  const bool whole = range_first == 0 && range_count == pairs_.size();
  std::vector<std::pair<size_t, size_t>> slice;
  if (!whole) slice.assign(pairs_.begin() + (ptrdiff_t)range_first, pairs_.begin() + (ptrdiff_t)(range_first + range_count));
  const std::vector<std::pair<size_t, size_t>>& plist = whole ? pairs_ : slice;
  const int64_t n = (int64_t)plist.size();
  const int host_thr = threads_ > 0 ? threads_ : planner::host_threads();
  std::vector<uint8_t> is_rev((size_t)n, 0);
  std::vector<awv_pair> ap((size_t)n);
  if (orientation_ == Orientation::Mash) {
    is_rev = planner::orient_pairs_mash(sequences_, plist, host_thr);  // alignment.rs:69-94 (host threads: the CLI's -t)
  } else if (orientation_ == Orientation::Wfa) {
    // determine_orientation_wfa (alignment.rs:157-175): align forward and reverse-complement with the
    // orientation params, compare #X+#I+#D; forward wins ties; a failed alignment counts as usize::MAX
    std::vector<awv_pair> op((size_t)2 * n);
    for (int64_t i = 0; i < n; ++i) {
      op[2 * i] = awv_pair{(int32_t)plist[i].first, (int32_t)plist[i].second, 0};
      op[2 * i + 1] = awv_pair{(int32_t)plist[i].first, (int32_t)plist[i].second, 1};
    }
    std::vector<awv_result> orr((size_t)2 * n);
    const awv_penalties open = to_penalties(orientation_params_);
    if (awv_align_pairs(eh.e, &open, op.data(), 2 * n, orr.data(), nullptr, nullptr) != AWV_OK)
      throw AlignmentError(std::string("orientation pass: ") + awv_last_error());
    for (int64_t i = 0; i < n; ++i) {
      auto dist = [](const awv_result& r) -> uint64_t {
        return r.status == AWV_ST_COMPLETED ? (uint64_t)r.num_mismatches + (uint64_t)r.num_ins + (uint64_t)r.num_del : UINT64_MAX;
      };
      is_rev[i] = dist(orr[2 * i]) <= dist(orr[2 * i + 1]) ? 0 : 1;
    }
  }
  for (int64_t i = 0; i < n; ++i) ap[i] = awv_pair{(int32_t)plist[i].first, (int32_t)plist[i].second, is_rev[i]};
  using BatchCb = std::function<void(int64_t, int64_t, const awv_result*, const uint8_t*, const std::vector<uint8_t>&)>;
  struct Ctx {
    const BatchCb* cb;
    const std::vector<uint8_t>* rev;
    std::exception_ptr err;
  } ctx{&batch_cb, &is_rev, nullptr};
  auto sink = [](void* user, int64_t first, int64_t cnt, const awv_result* res, const uint8_t* arena) -> int {
    Ctx* c = (Ctx*)user;
    try {
      (*c->cb)(first, cnt, res, arena, *c->rev);
    } catch (...) {
      c->err = std::current_exception();  // first error wins and aborts (iterator.rs:220-251)
      return 1;
    }
    return 0;
  };
  const awv_penalties pen = to_penalties(params_);
  lap("pairs oriented");
  const int rc = awv_align_pairs(eh.e, &pen, ap.data(), n, nullptr, sink, &ctx);
  lap("aligned + sunk");
  awv_engine_stats(eh.e, &stats_);
  if (ctx.err) std::rethrow_exception(ctx.err);
  if (rc != AWV_OK) throw AlignmentError(std::string("align_pairs: ") + awv_last_error());
}

void AllPairIterator::for_each_with_callback(const Callback& cb) {
  run([&](int64_t first, int64_t cnt, const awv_result* res, const uint8_t* arena, const std::vector<uint8_t>& rev) {
    for (int64_t i = 0; i < cnt; ++i) {
      const auto& pr = pairs_[(size_t)(first + i)];
      cb(make_result(pr.first, pr.second, rev[(size_t)(first + i)] != 0, res[i], arena, true));
    }
  });
}

void AllPairIterator::for_each_paf_batch(const std::function<void(const std::string&)>& sink, int format_threads) {
  run([&](int64_t first, int64_t cnt, const awv_result* res, const uint8_t* arena, const std::vector<uint8_t>& rev) {
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(format_threads, cnt / 64 + 1));
    std::vector<std::string> parts((size_t)T);
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) {
      th.emplace_back([&, t]() {
        const int64_t lo = cnt * t / T, hi = cnt * (t + 1) / T;
        std::string& out = parts[(size_t)t];
        out.reserve((size_t)(hi - lo) * 4096);
        for (int64_t i = lo; i < hi; ++i) {
          const auto& pr = pairs_[(size_t)(first + i)];
          const AlignmentResult a = make_result(pr.first, pr.second, rev[(size_t)(first + i)] != 0, res[i], arena, false);
          const bool ok = res[i].status == AWV_ST_COMPLETED;
          append_paf(out, a, ok ? arena + res[i].cigar_off : nullptr, ok ? res[i].cigar_len : 0, sequences_);
          out.push_back('\n');
        }
      });
    }
    for (auto& x : th) x.join();
#ifdef AWV_DEBUG_KNOBS
    if (getenv("AWH_TIMING")) fprintf(stderr, "[awh] formatted %lld pairs on %d threads\n", (long long)cnt, T);
#endif
    for (const auto& p : parts) sink(p);
  });
}

// ---- wfa.rs ---------------------------------------------------------------------------------
namespace wfa {

std::string validate_cigar_alignment(const uint8_t* cigar, size_t n, size_t query_len, size_t reference_len) {
  size_t q = 0, r = 0;
  char buf[160];
  for (size_t i = 0; i < n; ++i) {
    const uint8_t op = cigar[i];
    if (op == 'M' || op == '=' || op == 'X') {
      if (q >= query_len || r >= reference_len) {
        snprintf(buf, sizeof(buf), "CIGAR extends beyond sequences at M/=/X op: q_pos=%zu, r_pos=%zu, query_len=%zu, ref_len=%zu",
                 q, r, query_len, reference_len);
        return buf;
      }
      ++q; ++r;
    } else if (op == 'I') {  // WFA2: I consumes the reference
      if (r >= reference_len) {
        snprintf(buf, sizeof(buf), "CIGAR extends beyond reference at I op: r_pos=%zu, ref_len=%zu", r, reference_len);
        return buf;
      }
      ++r;
    } else if (op == 'D') {  // WFA2: D consumes the query
      if (q >= query_len) {
        snprintf(buf, sizeof(buf), "CIGAR extends beyond query at D op: q_pos=%zu, query_len=%zu", q, query_len);
        return buf;
      }
      ++q;
    } else {
      snprintf(buf, sizeof(buf), "Invalid CIGAR operation: %c (0x%02x)", (char)op, op);
      return buf;
    }
  }
  if (q != query_len) { snprintf(buf, sizeof(buf), "CIGAR doesn't cover full query: %zu vs %zu", q, query_len); return buf; }
  if (r != reference_len) { snprintf(buf, sizeof(buf), "CIGAR doesn't cover full reference: %zu vs %zu", r, reference_len); return buf; }
  return "";
}

AlignmentResult align_sequences(const std::vector<uint8_t>& pattern, const std::vector<uint8_t>& text,
                                const Penalties& p, AlignmentMode mode, int device) {
  awv_penalties q{};
  q.match = 0;
  q.mismatch = p.mismatch;
  switch (mode) {  // wfa.rs:185-218
    case AlignmentMode::EditDistance: q.gap_open1 = p.mismatch; q.gap_ext1 = p.mismatch; break;
    case AlignmentMode::SinglePieceAffine: q.gap_open1 = p.gap_opening1; q.gap_ext1 = p.gap_extension1; break;
    case AlignmentMode::TwoPieceAffine:
      q.gap_open1 = p.gap_opening1; q.gap_ext1 = p.gap_extension1;
      q.gap_open2 = p.gap_opening2; q.gap_ext2 = p.gap_extension2; q.two_piece = 1;
      break;
  }
  EngineHolder eh(device);
  std::vector<uint8_t> cig(pattern.size() + text.size() + 1);
  awv_result r{};
  if (awv_align_one(eh.e, &q, pattern.data(), (int32_t)pattern.size(), text.data(), (int32_t)text.size(), &r,
                    cig.data(), cig.size()) != AWV_OK)
    throw AlignmentError(std::string("Alignment failed: ") + awv_last_error());
  if (r.status != AWV_ST_COMPLETED) throw AlignmentError("Alignment failed with status: " + std::to_string(r.status));
  const std::string bad = validate_cigar_alignment(cig.data(), r.cigar_len, pattern.size(), text.size());
  if (!bad.empty()) throw AlignmentError("CIGAR validation failed: " + bad);
  AlignmentResult out;
  out.score = r.score;
  out.cigar = cigar_bytes_to_string(cig.data(), r.cigar_len);
  out.matches = (size_t)r.num_matches;
  out.mismatches = (size_t)r.num_mismatches;
  out.deletions = (size_t)r.num_ins;   // WFA2 'I' means standard 'D' (wfa.rs:94-96)
  out.insertions = (size_t)r.num_del;  // WFA2 'D' means standard 'I'
  out.alignment_length = out.matches + out.mismatches;
  return out;
}

}  // namespace wfa
}  // namespace allwave

namespace allwave {
SparsificationStrategy SparsificationStrategy::parse(const std::string& s) {  // main.rs:136-203
  SparsificationStrategy r;
  auto to_double = [](const std::string& t, const char* msg) {
    size_t used = 0;
    double v = 0;
    try { v = std::stod(t, &used); } catch (...) { throw std::invalid_argument(msg); }
    if (used != t.size() || t.empty()) throw std::invalid_argument(msg);
    return v;
  };
  auto to_usize = [](const std::string& t, const char* msg) {
    if (t.empty() || t.find_first_not_of("0123456789") != std::string::npos) throw std::invalid_argument(msg);
    return (size_t)std::stoull(t);
  };
  if (s == "none") return r;
  if (s == "auto") { r.kind = Auto; return r; }
  if (s.rfind("random:", 0) == 0) {
    r.kind = Random;
    r.value = to_double(s.substr(7), "Invalid random fraction");
    if (r.value <= 0.0 || r.value > 1.0) throw std::invalid_argument("Random fraction must be between 0 and 1");
    return r;
  }
  if (s.rfind("giant:", 0) == 0 || s.rfind("connectivity:", 0) == 0) {
    const bool giant = s[0] == 'g';
    r.kind = Connectivity;
    r.value = to_double(s.substr(giant ? 6 : 13), giant ? "Invalid giant component probability" : "Invalid connectivity probability");
    if (r.value <= 0.0 || r.value >= 1.0)
      throw std::invalid_argument(giant ? "Giant component probability must be between 0 and 1" : "Connectivity probability must be between 0 and 1");
    return r;
  }
  if (s.rfind("tree:", 0) == 0) {
    std::vector<std::string> parts;
    size_t pos = 5;
    while (true) {
      const size_t c = s.find(':', pos);
      parts.push_back(s.substr(pos, c == std::string::npos ? std::string::npos : c - pos));
      if (c == std::string::npos) break;
      pos = c + 1;
    }
    if (parts.size() < 3 || parts.size() > 4)
      throw std::invalid_argument("Invalid tree format. Use: tree:<k_nearest>:<k_farthest>:<random_fraction>[:<kmer_size>]");
    r.kind = TreeSampling;
    r.k_nearest = to_usize(parts[0], "Invalid k nearest count");
    r.k_farthest = to_usize(parts[1], "Invalid k farthest count");
    r.random_fraction = to_double(parts[2], "Invalid random fraction");
    if (r.k_nearest == 0 && r.k_farthest == 0)
      throw std::invalid_argument("At least one of k_nearest or k_farthest must be greater than 0");
    if (!(r.random_fraction >= 0.0 && r.random_fraction <= 1.0)) throw std::invalid_argument("Random fraction must be between 0 and 1");
    if (parts.size() == 4) {
      const size_t k = to_usize(parts[3], "Invalid k-mer size");
      if (k < 3 || k > 31) throw std::invalid_argument("K-mer size must be between 3 and 31");
      r.kmer_size = k;
    }
    return r;
  }
  throw std::invalid_argument("Invalid sparsification strategy. Use: none, auto, giant:<probability>, random:<fraction>, or tree:<near>:<far>:<random>[:<kmer>]");
}
}  // namespace allwave
