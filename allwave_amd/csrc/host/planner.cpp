// planner.cpp -- see planner.hpp
#include "planner.hpp"

#include <algorithm>
#include <atomic>
#include <queue>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>

namespace allwave {
namespace planner {

// ---- SipHash (Aumasson & Bernstein); Rust's DefaultHasher is SipHash-1-3 with k0 = k1 = 0 ----
static inline uint64_t rotl(uint64_t x, int b) { return (x << b) | (x >> (64 - b)); }

uint64_t siphash(const uint8_t* data, size_t n, uint64_t k0, uint64_t k1, int c_rounds, int d_rounds) {
  uint64_t v0 = 0x736f6d6570736575ULL ^ k0, v1 = 0x646f72616e646f6dULL ^ k1;
  uint64_t v2 = 0x6c7967656e657261ULL ^ k0, v3 = 0x7465646279746573ULL ^ k1;
  auto round = [&]() {
    v0 += v1; v1 = rotl(v1, 13); v1 ^= v0; v0 = rotl(v0, 32);
    v2 += v3; v3 = rotl(v3, 16); v3 ^= v2;
    v0 += v3; v3 = rotl(v3, 21); v3 ^= v0;
    v2 += v1; v1 = rotl(v1, 17); v1 ^= v2; v2 = rotl(v2, 32);
  };
  const size_t full = n / 8;
  for (size_t i = 0; i < full; ++i) {
    uint64_t m;
    memcpy(&m, data + 8 * i, 8);
    v3 ^= m;
    for (int r = 0; r < c_rounds; ++r) round();
    v0 ^= m;
  }
  uint64_t b = (uint64_t)n << 56;
  for (size_t i = 0; i < (n & 7); ++i) b |= (uint64_t)data[8 * full + i] << (8 * i);
  v3 ^= b;
  for (int r = 0; r < c_rounds; ++r) round();
  v0 ^= b;
  v2 ^= 0xff;
  for (int r = 0; r < d_rounds; ++r) round();
  return v0 ^ v1 ^ v2 ^ v3;
}

uint64_t default_hash_bytes(const uint8_t* p, size_t n) {
  // <[u8] as Hash>::hash: write_length_prefix(len) (a usize = u64 LE) then the bytes
  uint8_t buf[8 + 64];
  std::vector<uint8_t> big;
  uint8_t* b = buf;
  if (n > 64) { big.resize(n + 8); b = big.data(); }
  const uint64_t len = (uint64_t)n;
  memcpy(b, &len, 8);
  if (n) memcpy(b + 8, p, n);
  return siphash(b, n + 8, 0, 0, 1, 3);
}

uint64_t default_hash_str(const std::string& s) {
  // <str as Hash>::hash: the bytes followed by 0xFF
  std::vector<uint8_t> b(s.size() + 1);
  if (!s.empty()) memcpy(b.data(), s.data(), s.size());
  b[s.size()] = 0xFF;
  return siphash(b.data(), b.size(), 0, 0, 1, 3);
}

static inline bool is_dna_base(uint8_t b) {  // alignment.rs:151-154, mash.rs:116-119
  const uint8_t u = (b >= 'a' && b <= 'z') ? (uint8_t)(b - 32) : b;
  return u == 'A' || u == 'C' || u == 'G' || u == 'T';
}

template <bool CANONICAL>
static std::vector<uint64_t> sketch_impl(const std::vector<uint8_t>& seq, size_t k, size_t sketch_size) {
  std::vector<uint64_t> hashes;
  if (seq.size() < k || k == 0) return hashes;
  hashes.reserve(seq.size() - k + 1);
  // positions of the last non-ACGT byte seen, to skip k-mers containing one in O(1)
  size_t bad_until = 0;  // k-mers starting before this index contain a non-base
  for (size_t i = 0; i < k - 1 && i < seq.size(); ++i)
    if (!is_dna_base(seq[i])) bad_until = i + 1;
  std::vector<uint8_t> rc(k);
  for (size_t i = 0; i + k <= seq.size(); ++i) {
    if (!is_dna_base(seq[i + k - 1])) bad_until = i + k;
    if (i < bad_until) continue;
    uint64_t h = default_hash_bytes(seq.data() + i, k);
    if (CANONICAL) {  // mash.rs:121-133: upper-cases while complementing
      for (size_t j = 0; j < k; ++j) {
        const uint8_t b = seq[i + k - 1 - j];
        const uint8_t u = (b >= 'a' && b <= 'z') ? (uint8_t)(b - 32) : b;
        rc[j] = u == 'A' ? 'T' : u == 'T' ? 'A' : u == 'C' ? 'G' : u == 'G' ? 'C' : b;
      }
      h = std::min(h, default_hash_bytes(rc.data(), k));
    }
    hashes.push_back(h);
  }
  std::sort(hashes.begin(), hashes.end());
  if (hashes.size() > sketch_size) hashes.resize(sketch_size);  // truncate BEFORE deduplicating
  hashes.erase(std::unique(hashes.begin(), hashes.end()), hashes.end());
  return hashes;
}

std::vector<uint64_t> sketch_sequence_stranded(const std::vector<uint8_t>& seq, size_t k, size_t sketch_size) {
  return sketch_impl<false>(seq, k, sketch_size);
}
std::vector<uint64_t> sketch_sequence_canonical(const std::vector<uint8_t>& seq, size_t k, size_t sketch_size) {
  return sketch_impl<true>(seq, k, sketch_size);
}

double jaccard(const std::vector<uint64_t>& a, const std::vector<uint64_t>& b) {  // alignment.rs:125-139
  size_t i = 0, j = 0, inter = 0;
  while (i < a.size() && j < b.size()) {
    if (a[i] == b[j]) { ++inter; ++i; ++j; }
    else if (a[i] < b[j]) ++i;
    else ++j;
  }
  const size_t uni = a.size() + b.size() - inter;
  return uni == 0 ? 0.0 : (double)inter / (double)uni;
}

double mash_distance(double j, size_t k) {  // mash.rs:59-74
  if (j <= 0.0) return 1.0;
  const double ratio = (2.0 * j) / (1.0 + j);
  if (ratio <= 0.0) return 1.0;
  return (-1.0 / (double)k) * std::log(ratio);
}

template <typename F>
static void parallel_for(size_t n, int threads, F f) {
  const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(threads, 1), n));
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t) th.emplace_back([=]() { for (size_t i = n * t / T; i < n * (t + 1) / T; ++i) f(i); });
  for (auto& x : th) x.join();
}

std::vector<uint8_t> orient_pairs_mash(const std::vector<Sequence>& seqs,
                                       const std::vector<std::pair<size_t, size_t>>& pairs, int threads) {
  constexpr size_t K = 15, S = 1000;  // alignment.rs:70-75
  std::vector<std::vector<uint64_t>> fwd(seqs.size()), rev(seqs.size());
  parallel_for(seqs.size(), threads, [&](size_t i) {
    fwd[i] = sketch_sequence_stranded(seqs[i].seq, K, S);
    rev[i] = sketch_sequence_stranded(reverse_complement(seqs[i].seq), K, S);
  });
  std::vector<uint8_t> is_rev(pairs.size(), 0);
  parallel_for(pairs.size(), threads, [&](size_t p) {
    const size_t q = pairs[p].first, t = pairs[p].second;
    const double jf = jaccard(fwd[q], fwd[t]), jr = jaccard(rev[q], fwd[t]);
    is_rev[p] = jf >= jr ? 0 : 1;  // forward wins ties (alignment.rs:89)
  });
  return is_rev;
}

static std::atomic<int> g_host_threads{8};
void set_host_threads(int threads) { g_host_threads.store(std::max(1, threads)); }
int host_threads() { return g_host_threads.load(); }

std::vector<std::vector<double>> compute_distance_matrix(const std::vector<Sequence>& seqs, size_t k, size_t sketch_size) {
  const size_t n = seqs.size();
  std::vector<std::vector<uint64_t>> sk(n);
  parallel_for(n, host_threads(), [&](size_t i) { sk[i] = sketch_sequence_canonical(seqs[i].seq, k, sketch_size); });
  std::vector<std::vector<double>> m(n, std::vector<double>(n, 0.0));
  for (size_t i = 0; i < n; ++i)
    for (size_t j = i + 1; j < n; ++j) m[i][j] = m[j][i] = mash_distance(jaccard(sk[i], sk[j]), k);
  return m;
}

std::string format_distance_matrix(const std::vector<Sequence>& seqs, const std::vector<std::vector<double>>& m) {
  std::string out = "sequence";  // mash.rs:168-184
  for (const auto& s : seqs) { out += "\t"; out += s.id; }
  out += "\n";
  char buf[64];
  for (size_t i = 0; i < m.size(); ++i) {
    out += seqs[i].id;
    for (double d : m[i]) { snprintf(buf, sizeof(buf), "\t%.6f", d); out += buf; }
    out += "\n";
  }
  return out;
}

double compute_connectivity_probability(size_t n, double connectivity_prob) {  // iterator.rs:300-334
  if (n <= 1) return 1.0;
  const double x = std::min(std::max(connectivity_prob, 0.001), 0.999);
  if (n <= 10) return n == 2 ? 1.0 : n == 3 ? 0.8 : n == 4 ? 0.7 : n == 5 ? 0.6 : 0.5;
  const double nf = (double)n;
  const double c = -std::log(-std::log(x));
  const double p = (std::log(nf) + c) / nf;
  return std::min(std::max(p, 0.001), 1.0);
}

static inline bool keep_pair(const std::string& a, const std::string& b, double fraction) {  // iterator.rs:261-281
  const uint64_t h = default_hash_str(a + ":" + b);
  return (double)h / (double)UINT64_MAX < fraction;
}

std::vector<std::pair<size_t, size_t>> apply_random_sparsification(std::vector<std::pair<size_t, size_t>> pairs,
                                                                   double keep_fraction, const std::vector<Sequence>& seqs) {
  std::vector<std::pair<size_t, size_t>> out;
  for (const auto& p : pairs)
    if (keep_pair(seqs[p.first].id, seqs[p.second].id, keep_fraction)) out.push_back(p);
  return out;
}

std::vector<std::pair<size_t, size_t>> build_knn_graph(const std::vector<std::vector<double>>& d, size_t k, bool farthest) {
  const size_t n = d.size();  // knn_graph.rs:112-143 (Rust's sort_by is stable)
  std::vector<std::pair<size_t, size_t>> pairs;
  for (size_t i = 0; i < n; ++i) {
    std::vector<std::pair<double, size_t>> nb;
    for (size_t j = 0; j < n; ++j)
      if (j != i) nb.emplace_back(d[i][j], j);
    if (farthest) std::stable_sort(nb.begin(), nb.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
    else std::stable_sort(nb.begin(), nb.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    for (size_t t = 0; t < std::min(k, nb.size()); ++t) pairs.emplace_back(i, nb[t].second);
  }
  return pairs;
}

std::vector<std::pair<size_t, size_t>> extract_tree_pairs(const std::vector<Sequence>& seqs, size_t k_nearest,
                                                          size_t k_farthest, double random_fraction, size_t kmer_size) {
  std::vector<std::pair<size_t, size_t>> all;  // knn_graph.rs:12-52
  if (seqs.size() < 2) return all;
  const auto dm = compute_distance_matrix(seqs, kmer_size, 1000);
  if (k_nearest > 0) { auto p = build_knn_graph(dm, k_nearest, false); all.insert(all.end(), p.begin(), p.end()); }
  if (k_farthest > 0) { auto p = build_knn_graph(dm, k_farthest, true); all.insert(all.end(), p.begin(), p.end()); }
  if (random_fraction > 0.0)
    for (size_t i = 0; i < seqs.size(); ++i)
      for (size_t j = 0; j < seqs.size(); ++j)
        if (i != j && keep_pair(seqs[i].id, seqs[j].id, random_fraction)) all.emplace_back(i, j);
  std::sort(all.begin(), all.end());
  all.erase(std::unique(all.begin(), all.end()), all.end());
  return all;
}


double predicted_pair_cost(size_t qlen, size_t tlen, const AlignmentParams& params) {
  const double lo = (double)std::min(qlen, tlen), g = (double)(qlen > tlen ? qlen - tlen : tlen - qlen);
  double gap = 0;
  if (g > 0) {
    gap = params.gap_open + g * params.gap_extend;
    if (params.gap2_open && params.gap2_extend) gap = std::min(gap, (double)*params.gap2_open + g * (double)*params.gap2_extend);
  }
  const double s = 0.06 * params.mismatch_penalty * lo + gap + 16.0;  // ~6 % of the bases pay a mismatch's worth
  return s * s;
}

std::vector<uint32_t> assign_shards_lpt(const std::vector<double>& cost, size_t world) {
  const size_t n = cost.size();
  std::vector<uint32_t> shard(n, 0);
  if (world <= 1 || n == 0) return shard;
  bool uniform = true;
  for (size_t i = 1; i < n && uniform; ++i) uniform = cost[i] == cost[0];
  if (uniform) {  // LPT on equal costs is the strided shard
    for (size_t i = 0; i < n; ++i) shard[i] = (uint32_t)(i % world);
    return shard;
  }
  std::vector<uint32_t> order(n);
  for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
  // min-heap of (load, rank)
  typedef std::pair<double, uint32_t> LR;
  std::priority_queue<LR, std::vector<LR>, std::greater<LR>> heap;
  for (size_t r = 0; r < world; ++r) heap.push(LR(0.0, (uint32_t)r));
  for (size_t k = 0; k < n; ++k) {
    LR top = heap.top();
    heap.pop();
    shard[order[k]] = top.second;
    top.first += cost[order[k]];
    heap.push(top);
  }
  return shard;
}

}  // namespace planner
}  // namespace allwave
