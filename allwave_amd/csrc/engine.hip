// engine.hip -- host side of liballwave_hip.so: the C ABI of include/allwave_hip.h.
//
// Owns all device state for one GPU (what the reference keeps in per-thread cached WFA2
// aligners, /root/reference/src/alignment.rs:11-22,210-221): sequence copies, per-workgroup
// wavefront arenas sized for 288 GB of HBM, result/CIGAR arenas, one stream.  There is no CPU
// fallback: without a HIP device every entry point fails with AWV_ERR_NO_DEVICE.
#include "allwave_hip.h"
// the device code, once per workgroup size: awv:: one wave per pair (throughput), awvw:: four waves per pair,
// awvx:: sixteen waves per pair (one pair per CU: the few pairs a large length difference makes enormous)
#include "kernels_awv.hpp"  // (AWV_THRU_WG; the awv:: kernels themselves are instantiated in kernels_awv.hip -- here only the types)
#define AWV_NS awv
#define AWV_WG AWV_THRU_WG
#if AWV_THRU_WG == 128
#define AWV_DIRSPLIT 1
#endif
#include "biwfa_device.hpp"
#undef AWV_NS
#undef AWV_WG
#undef AWV_DIRSPLIT
#define AWV_NS awvw
#define AWV_WG 256
#include "biwfa_device.hpp"
#undef AWV_NS
#undef AWV_WG
#define AWV_NS awvx
#define AWV_WG 1024
#include "biwfa_device.hpp"
#undef AWV_NS
#undef AWV_WG
// awvw_m:: / awvx_m:: the same two with AWV_WIDE16: 16-bit rows for pairs of which only the shorter sequence fits 16
// bits (cells stored as min(h, v), 32-bit row metadata) -- only their 16-bit-row kernels are used
#define AWV_WIDE16 1
#define AWV_NS awvw_m
#define AWV_WG 256
#include "biwfa_device.hpp"
#undef AWV_NS
#undef AWV_WG
#define AWV_NS awvx_m
#define AWV_WG 1024
#include "biwfa_device.hpp"
#undef AWV_NS
#undef AWV_WG
#undef AWV_WIDE16

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <thread>
#include <vector>

namespace {
constexpr size_t EV_EXTRA = (awv::STACK_CAP * sizeof(awv::Task) + 3) / 4 + 16;  // uint32 words behind the events: the DFS stack

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess)                                                                      \
      return fail(_e == hipErrorOutOfMemory ? AWV_ERR_OOM : AWV_ERR_HIP,                       \
                  std::string(#expr) + ": " + hipGetErrorString(_e));                          \
  } while (0)

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;  // elements
  int reserve(size_t n) {
    if (n <= cap) return AWV_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = n + std::min<size_t>(n / 8, (size_t)64 << 20) + 64;  // growth slack, bounded for the big arenas
    hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
    if (e != hipSuccess) {
      want = n;
      e = hipMalloc((void**)&p, want * sizeof(T));
    }
    if (e != hipSuccess) {
      p = nullptr;
      return fail(AWV_ERR_OOM, std::string("hipMalloc ") + std::to_string(want * sizeof(T)) + " bytes: " + hipGetErrorString(e));
    }
    cap = want;
    return AWV_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  void adopt(T* q, size_t n) {
    release();
    p = q;
    cap = n;
  }
  size_t bytes() const { return cap * sizeof(T); }
};

struct SeqSet {
  int32_t n = 0;
  std::vector<uint64_t> off;   // padded device layout
  std::vector<int32_t> len;
  DevBuf<uint8_t> d_seq[4];
  DevBuf<uint64_t> d_off;
  DevBuf<int32_t> d_len;
  DevBuf<uint32_t> d_seq2[2];  // 2-bit packed forward / reverse-complement
  DevBuf<uint64_t> d_off2;     // word offsets
  DevBuf<uint8_t> d_ok2;       // bit0 forward packable, bit1 reverse-complement packable
  size_t total = 0;
  void release() {
    for (auto& b : d_seq) b.release();
    for (auto& b : d_seq2) b.release();
    d_off2.release();
    d_ok2.release();
    d_off.release();
    d_len.release();
    n = 0;
  }
};

// reverse_complement of /root/reference/src/alignment.rs:178-190
inline uint8_t rc_base(uint8_t b) {
  switch (b) {
    case 'A': case 'a': return 'T';
    case 'T': case 't': return 'A';
    case 'C': case 'c': return 'G';
    case 'G': case 'g': return 'C';
    default: return 'N';
  }
}

}  // namespace

struct awv_engine {
  awv_engine_config cfg{};
  int device = 0;
  int num_cus = 0;
  int wall_clock_khz = 100000;  // rate of s_memrealtime (hipDeviceAttributeWallClockRate)
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  SeqSet seqs;
  // scratch arenas (per persistent workgroup)
  DevBuf<uint8_t> ring_mem, hist_mem;
  DevBuf<uint32_t> ev_mem;
  // per-launch buffers
  DevBuf<int32_t> d_pair_q, d_pair_t, d_pair_rc;
  DevBuf<uint64_t> d_cigar_off;
  DevBuf<awv::DevResult> d_results;
  DevBuf<uint8_t> d_cigar;
  DevBuf<unsigned long long> d_counters;  // [0] work cursor, [1..] stats
  std::vector<uint8_t> h_cigar;
  awv_stats stats{};
};

namespace {

int upload_seqset(awv_engine* e, SeqSet& s, int32_t n, const uint8_t* bytes, const uint64_t* offsets) {
  if (n < 0 || (n > 0 && (!bytes || !offsets))) return fail(AWV_ERR_ARG, "set_sequences: null input");
  s.n = 0;  // the set counts as loaded (awv_align_pairs passes its AWV_ERR_STATE check) only once every copy below has landed
  s.off.assign((size_t)n + 1, 0);
  s.len.assign((size_t)n, 0);
  uint64_t run = 0;
  for (int i = 0; i < n; ++i) {
    if (offsets[i + 1] < offsets[i]) return fail(AWV_ERR_ARG, "set_sequences: offsets not monotone");
    const uint64_t l = offsets[i + 1] - offsets[i];
    if (l > (uint64_t)(INT32_MAX / 4)) return fail(AWV_ERR_ARG, "set_sequences: sequence too long");
    s.off[i] = run;
    s.len[i] = (int32_t)l;
    run += l + 8;  // extend reads 8 bytes at a time: keep over-reads inside the allocation
  }
  s.off[n] = run;
  s.total = run + 64;
  std::vector<uint8_t> h[4];
  for (auto& v : h) v.assign(s.total, 0);
  for (int i = 0; i < n; ++i) {
    const uint8_t* src = bytes + offsets[i];
    const size_t l = (size_t)s.len[i];
    uint8_t* f = h[0].data() + s.off[i];
    uint8_t* r = h[1].data() + s.off[i];
    uint8_t* c = h[2].data() + s.off[i];
    uint8_t* cr = h[3].data() + s.off[i];
    for (size_t j = 0; j < l; ++j) {
      const uint8_t b = src[j];
      f[j] = b;
      r[l - 1 - j] = b;
      const uint8_t cb = rc_base(b);
      c[l - 1 - j] = cb;  // reverse complement
      cr[j] = cb;         // its reversal
    }
  }
  // 2-bit packed copies (A,C,G,T -> 0..3, 16 bases per word) for sequences made of upper-case ACGT only;
  // bytes are compared verbatim by the reference, so anything else keeps the raw-byte path
  // (two pad words in front of the first sequence, one behind every sequence and four at the end: the kernels' probes of
  // packed words in place -- seq_mode 2 -- read up to two words below a sequence's first and two beyond its last)
  std::vector<uint64_t> off2((size_t)n + 1, 2);
  std::vector<uint8_t> ok2((size_t)std::max(n, 1), 0);
  for (int i = 0; i < n; ++i) off2[i + 1] = off2[i] + ((size_t)s.len[i] + 15) / 16 + 1;
  std::vector<uint32_t> h2[2];
  for (auto& v : h2) v.assign(off2[n] + 4, 0u);
  auto code = [](uint8_t b) -> int { return b == 'A' ? 0 : b == 'C' ? 1 : b == 'G' ? 2 : b == 'T' ? 3 : -1; };
  for (int i = 0; i < n; ++i) {
    for (int v = 0; v < 2; ++v) {
      const uint8_t* src = h[v ? 2 : 0].data() + s.off[i];
      uint32_t* dst = h2[v].data() + off2[i];
      bool ok = true;
      for (size_t j = 0; j < (size_t)s.len[i]; ++j) {
        const int c = code(src[j]);
        if (c < 0) { ok = false; break; }
        dst[j >> 4] |= (uint32_t)c << (2 * (j & 15));
      }
      if (ok) ok2[i] |= (uint8_t)(1 << v);
    }
  }
  HIP_TRY(hipSetDevice(e->device));
  for (int v = 0; v < 2; ++v) {
    if (int rc = s.d_seq2[v].reserve(h2[v].size())) return rc;
    HIP_TRY(hipMemcpyAsync(s.d_seq2[v].p, h2[v].data(), h2[v].size() * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
  }
  if (int rc = s.d_off2.reserve((size_t)n + 1)) return rc;
  if (int rc = s.d_ok2.reserve(ok2.size())) return rc;
  HIP_TRY(hipMemcpyAsync(s.d_off2.p, off2.data(), ((size_t)n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(s.d_ok2.p, ok2.data(), ok2.size(), hipMemcpyHostToDevice, e->stream));
  for (int v = 0; v < 4; ++v) {
    if (int rc = s.d_seq[v].reserve(s.total)) return rc;
    HIP_TRY(hipMemcpyAsync(s.d_seq[v].p, h[v].data(), s.total, hipMemcpyHostToDevice, e->stream));
  }
  if (int rc = s.d_off.reserve((size_t)n + 1)) return rc;
  if (int rc = s.d_len.reserve((size_t)std::max(n, 1))) return rc;
  HIP_TRY(hipMemcpyAsync(s.d_off.p, s.off.data(), ((size_t)n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, e->stream));
  if (n > 0) HIP_TRY(hipMemcpyAsync(s.d_len.p, s.len.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  s.n = n;
  return AWV_OK;
}

int check_penalties(const awv_penalties* p, awv::DevPenalties& d) {
  if (!p) return fail(AWV_ERR_ARG, "penalties: null");
  if (p->match != 0) return fail(AWV_ERR_PENALTIES, "match score must be 0 (WFA2 penalty transformation is out of scope)");
  if (p->mismatch <= 0 || p->gap_open1 < 0 || p->gap_ext1 <= 0) return fail(AWV_ERR_PENALTIES, "need x > 0, o >= 0, e > 0");
  if (p->two_piece && (p->gap_open2 < 0 || p->gap_ext2 <= 0)) return fail(AWV_ERR_PENALTIES, "need o2 >= 0, e2 > 0");
  d.x = p->mismatch;
  d.o1 = p->gap_open1;
  d.e1 = p->gap_ext1;
  d.two_piece = p->two_piece ? 1 : 0;
  d.o2 = d.two_piece ? p->gap_open2 : p->gap_open1;
  d.e2 = d.two_piece ? p->gap_ext2 : p->gap_ext1;
  int scope = std::max(d.x, d.o1 + d.e1);
  if (d.two_piece) scope = std::max(scope, d.o2 + d.e2);
  d.scope = scope + 1;
  if (d.scope + 2 > awv::MAX_RING) return fail(AWV_ERR_PENALTIES, "penalties too large: max(x, o+e) must be < 126");
  return AWV_OK;
}

// ---- arena placement --------------------------------------------------------------------------
// Where a multi-GB arena lands in HBM is worth up to +-6 % of the row traffic's rate (measured: the
// same kernel on five arenas allocated one after the other ran 1315..1465 ms, each arena keeping its
// rate; scratch/src/memplace.hip shows the same split with nothing but the traffic).  So the ring
// arena is chosen: a few candidates are allocated, each is timed with a few milliseconds of the step
// kernel's traffic pattern (rows 1, 2, 5, 10 and 25 steps old read, five rows written, per 248-column
// window, one wave per workgroup slot), the fastest is kept and the others are freed.
__global__ __launch_bounds__(64, 4) void arena_probe_kernel(unsigned char* arena, size_t slot_stride, int row_bytes, int steps, int width_cols,
                                                             unsigned long long* sink) {
  const int lane = threadIdx.x;
  awv::rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(arena + (size_t)blockIdx.x * slot_stride, 0, (int)std::min<size_t>(slot_stride, 0x7FFFFFFF), 0x00020000);
  auto off = [&](int dir, int comp, int score) { return ((dir * 5 + comp) * 32 + (score & 31)) * row_bytes; };
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  unsigned acc = 0;
  for (int s = 32; s < 32 + steps; ++s) {
    for (int dir = 0; dir < 2; ++dir) {
      const int lo = 1024 + (s & 7) * 4;
      for (int cb = lo; cb < lo + width_cols; cb += 248) {
        const int voff = (cb + lane * 4) * 2;
        const u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, off(dir, 0, s - 5), 0);
        const u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, off(dir, 0, s - 10), 0);
        const u32x2 c = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, off(dir, 0, s - 25), 0);
        const u32x2 d = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, off(dir, 1, s - 2), 2);
        const u32x2 f = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, off(dir, 3, s - 2), 2);
        const u32x2 g = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, off(dir, 2, s - 1), 2);
        const u32x2 h = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, off(dir, 4, s - 1), 2);
        const u32x2 m = a + b + c;
        acc += m[0] ^ m[1];
        __builtin_amdgcn_raw_buffer_store_b64(b + d, rs, voff, off(dir, 1, s), 0);
        __builtin_amdgcn_raw_buffer_store_b64(b + f, rs, voff, off(dir, 3, s), 0);
        __builtin_amdgcn_raw_buffer_store_b64(c + g, rs, voff, off(dir, 2, s), 0);
        __builtin_amdgcn_raw_buffer_store_b64(c + h, rs, voff, off(dir, 4, s), 0);
        __builtin_amdgcn_raw_buffer_store_b64(m, rs, voff, off(dir, 0, s), 0);
      }
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// worst-case penalty of an end-to-end alignment of two sequences of length <= n
long long worst_case_penalty(const awv::DevPenalties& d, long long n) {
  auto gap = [&](long long l) {
    long long g = d.o1 + l * d.e1;
    if (d.two_piece) g = std::min(g, (long long)d.o2 + l * d.e2);
    return g;
  };
  return std::min(2 * gap(n), n * (long long)d.x + gap(n));
}

int align_core(awv_engine* e, SeqSet& s, const awv_penalties* pen, const awv_pair* pairs, int64_t npairs,
               awv_result* out, awv_sink sink, void* user) {
  using namespace awv;
  if (npairs < 0 || (npairs > 0 && !pairs)) return fail(AWV_ERR_ARG, "align_pairs: null pairs");
  DevPenalties dp{};
  if (int rc = check_penalties(pen, dp)) return rc;
  HIP_TRY(hipSetDevice(e->device));
  e->stats = awv_stats{};
  if (npairs == 0) return AWV_OK;
  for (int64_t i = 0; i < npairs; ++i) {
    if (pairs[i].q_idx < 0 || pairs[i].q_idx >= s.n || pairs[i].t_idx < 0 || pairs[i].t_idx >= s.n)
      return fail(AWV_ERR_ARG, "align_pairs: sequence index out of range");
  }
  // multi-step passes (biwfa_device.hpp, compute_rows_multi): T steps per pass, T <= the nearest M source
  // (so that every M row a pass reads was written by an earlier pass); instantiated for the I/D depths
  // of the reference's presets (2-piece: e1 = 2 / e2 = 1; gap-affine: e = 1 or 2)
  int multi_T = std::min(std::min(dp.x, dp.o1 + dp.e1), awv::TMAX);
  if (dp.two_piece) multi_T = std::min(multi_T, dp.o2 + dp.e2);
  if (dp.two_piece ? (dp.e1 != 2 || dp.e2 != 1) : (dp.e1 != 1 && dp.e1 != 2)) multi_T = 0;
  if (multi_T < 2 || (e->cfg.flags & AWV_F_SINGLE_STEP)) multi_T = 0;
  // chained sweeps (compute_rows_multi, CHAIN): the previous one / two sweeps' M rows are this sweep's sources 5 and 10
  // scores back -- only with x = TMAX and o1 + e1 = 2 TMAX (the default scores); the third source must still come from
  // earlier passes: TMAX * chain_max <= o2 + e2
  int chain_max = 1;
  if (multi_T == awv::TMAX && dp.two_piece && dp.x == awv::TMAX && dp.o1 + dp.e1 == 2 * awv::TMAX && !(e->cfg.flags & AWV_F_NO_CHAIN))
    chain_max = std::max(1, std::min(awv::CHAIN_MAX, (dp.o2 + dp.e2) / awv::TMAX));
  int ring = 4;
  while (ring < dp.scope + 2 + (multi_T > 0 ? multi_T * chain_max - 1 : 0)) ring *= 2;
  const int64_t max_batch = e->cfg.max_batch_pairs > 0 ? e->cfg.max_batch_pairs : (int64_t)1 << 20;
  uint64_t max_arena = e->cfg.max_arena_bytes > 0 ? (uint64_t)e->cfg.max_arena_bytes : (uint64_t)8 << 30;
  // Experiment knobs read from the environment exist only in a -DAWV_DEBUG_KNOBS build; the product library's
  // behaviour depends on its arguments alone.
#ifdef AWV_DEBUG_KNOBS
  if (const char* env = getenv("AWV_MAX_ARENA_MB")) max_arena = std::max<uint64_t>(1, (uint64_t)atoll(env)) << 20;
  const bool inline_sink = getenv("AWV_INLINE_SINK") != nullptr;  // every sink on the calling thread
#else
  const bool inline_sink = false;
#endif
  // base-case capacities: score_remaining <= 250 or both lengths <= 100
  // A sub-problem that ends in an indel component pays that gap's open on top of the
  // score_remaining its parent hands down (the reverse aligner starts with the open pre-paid).
  const long long sb = std::max<long long>(FALLBACK_MIN_SCORE + std::max(dp.o1, dp.o2),
                                           worst_case_penalty(dp, FALLBACK_MIN_LENGTH));
  if (sb > 4000) return fail(AWV_ERR_PENALTIES, "penalties too large for the base-case history");
  const int sb_cap = (int)sb;
  const int wb_cap = ((2 * sb_cap + 9 + 2 * COL_PAD) + 63) & ~63;
  // dynamic LDS: metadata region (BiWFA ring metadata, aliased with the base-case table) + sequences
  auto lds_meta_bytes = [&](size_t meta_elem) {
    return ((size_t)2 * NCOMP * ring * meta_elem + (size_t)4 * ring * sizeof(int) + (size_t)dp.scope * NCOMP * sizeof(int) + 15) &
           ~(size_t)15;
  };

#ifdef AWV_DEBUG_KNOBS
  const bool timing = getenv("AWV_TIMING") != nullptr;  // diagnostic: host-side stage times on stderr
#else
  const bool timing = false;
#endif
  const auto tl0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (timing) fprintf(stderr, "[awv] %-22s %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - tl0).count());
  };
  std::vector<int32_t> hq, ht, hrc;
  std::vector<uint64_t> hoff;
  std::vector<awv_result> hres;
  int64_t first = 0;
  double kernel_ms = 0, h2d_ms = 0, d2h_ms = 0;
  unsigned long long stat_tot[STAT_N] = {0};
  uint64_t launches = 0;
  // With several batches in a call, the sink of batch i (the caller's formatting and output) runs on a
  // helper thread while batch i+1 is carved, launched and copied back.  It is joined before the next
  // sink starts: sinks never overlap, come in batch order, and a batch's result / CIGAR buffers stay
  // untouched until its sink has returned.  The last batch's sink runs on the calling thread.
  struct SinkRunner {
    std::thread th;
    int rc = 0;
    std::vector<awv_result> res;
    std::vector<uint8_t> cig;
    int wait() {
      if (th.joinable()) th.join();
      const int r = rc;
      rc = 0;
      return r;
    }
    ~SinkRunner() { if (th.joinable()) th.join(); }
  } runner;
  while (first < npairs) {
    // ---- carve a batch
    int64_t n = 0;
    uint64_t arena = 0;
    int maxsum = 0, maxlen = 0;
    hq.clear(); ht.clear(); hrc.clear(); hoff.clear();
    while (first + n < npairs && n < max_batch) {
      const awv_pair& p = pairs[first + n];
      const int ql = s.len[p.q_idx], tl = s.len[p.t_idx];
      const uint64_t need = ((uint64_t)ql + (uint64_t)tl + 7) & ~(uint64_t)7;
      if (n > 0 && arena + need > max_arena) break;
      hq.push_back(p.q_idx);
      ht.push_back(p.t_idx);
      hrc.push_back(p.q_revcomp ? 1 : 0);
      hoff.push_back(arena);
      arena += need;
      maxsum = std::max(maxsum, ql + tl);
      maxlen = std::max(maxlen, std::max(ql, tl));
      ++n;
    }
    // Dispatch order: workgroups take pairs from a shared cursor, so the most expensive pairs go
    // first (longest-processing-time-first; the work of a pair grows with the square of its score,
    // estimated here from the lengths and the gap a length difference forces).  Results and CIGARs
    // keep their caller-order slots through the index map.  Equal-cost batches keep caller order.
    std::vector<int64_t> amap;  // dispatch index -> batch index (empty = identity)
    bool skewed = false;  // the most expensive pair costs several times the median one
    {
      std::vector<int64_t> order((size_t)n);
      std::vector<uint64_t> cost((size_t)n);
      bool uniform = true;
      for (int64_t i = 0; i < n; ++i) {
        order[(size_t)i] = i;
        const int64_t ql = s.len[hq[(size_t)i]], tl = s.len[ht[(size_t)i]];
        cost[(size_t)i] = (uint64_t)(ql + tl + 4 * std::llabs(ql - tl));
        uniform = uniform && cost[(size_t)i] == cost[0];
      }
      if (!uniform) {
        std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return cost[(size_t)a] > cost[(size_t)b]; });
        std::vector<int32_t> q2((size_t)n), t2((size_t)n), rc2((size_t)n);
        std::vector<uint64_t> off2((size_t)n);
        for (int64_t i = 0; i < n; ++i) {
          const size_t o = (size_t)order[(size_t)i];
          q2[(size_t)i] = hq[o]; t2[(size_t)i] = ht[o]; rc2[(size_t)i] = hrc[o]; off2[(size_t)i] = hoff[o];
        }
        hq.swap(q2); ht.swap(t2); hrc.swap(rc2); hoff.swap(off2);
        skewed = cost[(size_t)order[0]] >= 3 * cost[(size_t)order[(size_t)n / 2]];
        amap.swap(order);
      }
    }
    if (int rc = e->d_pair_q.reserve((size_t)n)) return rc;
    if (int rc = e->d_pair_t.reserve((size_t)n)) return rc;
    if (int rc = e->d_pair_rc.reserve((size_t)n)) return rc;
    if (int rc = e->d_cigar_off.reserve((size_t)n)) return rc;
    if (int rc = e->d_results.reserve((size_t)n)) return rc;
    if (int rc = e->d_cigar.reserve((size_t)arena + 64)) return rc;
    if (int rc = e->d_counters.reserve(1 + STAT_N)) return rc;
    static_assert(sizeof(awv_result) == sizeof(DevResult), "result layout");
    hres.assign((size_t)n, awv_result{});
    // One group of the batch = one kernel flavour: `wide` pairs get a 256-thread workgroup each (four
    // waves deal a row's windows among themselves), the others one wave each.
    auto run_group = [&](std::vector<int32_t> hq, std::vector<int32_t> ht, std::vector<int32_t> hrc, std::vector<uint64_t> hoff,
                         std::vector<int64_t> amap, int waves, int width, int g_maxsum, int g_maxlen, bool reserve_only) -> int {
    // width: 0 = 16-bit rows (both lengths < 32760), 1 = 32-bit rows, 2 = 16-bit rows holding min(h, v) with 32-bit
    // row metadata (AWV_WIDE16: only the shorter length < 32760)
    const bool narrow = width != 1, wide_meta = width == 2;
    if (hq.empty()) return AWV_OK;
    const int wg = waves == 1 ? AWV_THRU_WG : 64 * waves;
    const int nslots_g = e->cfg.workgroups > 0 ? std::max(1, e->cfg.workgroups / (wg / 64)) : (WAVES_PER_SIMD * 256 / wg) * e->num_cus;
    const int wcap_full = ((g_maxsum + 9 + 256 + 2 * COL_PAD) + 255) & ~255;
    const int nslots_want = (int)std::min<int64_t>(nslots_g, (int64_t)hq.size());
    // 16-bit wavefront rows whenever every offset fits (halves the HBM/L2 traffic of the rings): the caller
    // groups pairs by row width, so one long sequence no longer drags a whole batch to 32-bit rows
    const size_t esz = narrow ? 2 : 4;
    // dynamic LDS = ring metadata (16-bit entries with 16-bit rows) + staging of the 2-bit packed
    // sequences: what the largest pair needs, within 160 KB / (16 waves per CU) per workgroup;
    // sub-problems that do not fit read global memory instead
    const size_t lds_meta = lds_meta_bytes(narrow && !wide_meta ? sizeof(RowMeta16) : sizeof(RowMeta));
    const size_t seq_need = ((((size_t)g_maxlen + 15) / 16 + 2) * 2 + 10) * 4;
    const size_t lds_budget = (size_t)(160 * 1024 / (WAVES_PER_SIMD * 256 / wg)) - STATIC_LDS_RESERVE;
    size_t lds_seq = (e->cfg.flags & AWV_F_NO_PACKED_SEQ) ? 0 : (lds_meta < lds_budget ? std::min(seq_need, lds_budget - lds_meta) : 0);
    lds_seq &= ~(size_t)15;
    const size_t dyn_lds = lds_meta + lds_seq;
    const size_t hist_rows = ((size_t)(sb_cap + 1) * NCOMP * wb_cap * esz + 63) & ~(size_t)63;
    const size_t hist_stride = hist_rows + (((size_t)(sb_cap + 1) * NCOMP * sizeof(RowMeta) + 63) & ~(size_t)63);
    // per-workgroup arenas as a function of the row capacity (columns)
    const size_t budget = e->cfg.max_scratch_bytes > 0 ? (size_t)e->cfg.max_scratch_bytes : (size_t)160 << 30;
    auto per_slot = [&](int wc) {
      return (size_t)2 * NCOMP * ring * wc * esz + hist_stride + (size_t)(wc + EV_EXTRA) * sizeof(uint32_t);
    };
    // Row capacity of the first attempt.  A wavefront at score s spans at most ~2 s / min(e) diagonals,
    // far fewer than plen + tlen for similar sequences; when full-width rows would not leave room for
    // every workgroup's arenas (long sequences), start with the widest rows that do and re-run only
    // the pairs whose wavefronts outgrow them (status CAPACITY) with wider rows.
    // Half-width rows (at least 8192 columns) already hold every pair whose optimal score is below
    // about a quarter of plen + tlen; they halve the arenas (allocating -- and the driver clearing --
    // tens of GB is the largest fixed cost of a call) and only unusually divergent pairs are re-run.
    // (a length difference forces a gap that long: the rows must span it in both directions)
    int g_maxdelta = 0;
    for (size_t i = 0; i < hq.size(); ++i) g_maxdelta = std::max(g_maxdelta, std::abs(s.len[hq[i]] - s.len[ht[i]]));
    int wcap = std::min(wcap_full, std::max(std::max(8192, (wcap_full / 2 + 255) & ~255), (2 * g_maxdelta + 4096 + 255) & ~255));
    if (per_slot(wcap) * (size_t)nslots_want > budget) {
      const size_t fixed = hist_stride + EV_EXTRA * sizeof(uint32_t);
      const size_t per_col = (size_t)2 * NCOMP * ring * esz + sizeof(uint32_t);
      const size_t share = budget / (size_t)nslots_want;
      long long wc = share > fixed ? (long long)((share - fixed) / per_col) : 0;
      wc = std::max<long long>(wc & ~255LL, 8192);
      wcap = (int)std::min<long long>(wcap, wc);
    }
    // (diagnostic / test hook: a narrower first attempt, so that the CAPACITY re-run path can be exercised on short sequences)
    if (e->cfg.first_row_cols > 0) wcap = std::min(wcap, std::max(2048, (e->cfg.first_row_cols + 255) & ~255));
    // ---- attempts: the whole batch at row capacity `wcap`, then only the pairs that outgrew it
    std::vector<awv_result> tres;
    float ms = 0;
    for (int wc = wcap;;) {
      const int64_t m = (int64_t)hq.size();
      const size_t ring_stride = (size_t)2 * NCOMP * ring * wc * esz;
      const size_t ev_stride = (size_t)wc + EV_EXTRA;  // run-length events + the DFS stack
      int nslots = (int)std::min<int64_t>(nslots_g, m);
      {  // keep the per-workgroup arenas inside the scratch budget (default 160 GiB of the 288 GB HBM)
        const size_t fit = std::max<size_t>(1, budget / per_slot(wc));
        if ((size_t)nslots > fit) nslots = (int)fit;
      }
      if (ring_stride * nslots > e->ring_mem.cap) {  // the ring arena grows: choose where it lies (see arena_probe_kernel)
        const size_t want = ring_stride * (size_t)nslots + ((size_t)64 << 20);
        size_t free_b = 0, total_b = 0;
        e->ring_mem.release();
        (void)hipMemGetInfo(&free_b, &total_b);
        int ncand = (e->cfg.flags & AWV_F_NO_ARENA_PROBE) ? 1 : 4;
        while (ncand > 1 && (size_t)ncand * want > free_b / 10 * 8) --ncand;  // all candidates are alive at once
        const bool probe_ok = ncand > 1 && want >= ((size_t)1 << 30) && esz == 2 && ring_stride >= (size_t)2 * 5 * 32 * 4096 &&
                              (size_t)wc * esz >= 4096;
        if (!probe_ok) ncand = 1;
        uint8_t* cand[4] = {nullptr, nullptr, nullptr, nullptr};
        float cms[4] = {0, 0, 0, 0};
        int best = -1;
        for (int c = 0; c < ncand; ++c) {
          if (hipMalloc((void**)&cand[c], want) != hipSuccess) { cand[c] = nullptr; (void)hipGetLastError(); break; }
          if (ncand == 1) { best = 0; break; }
          float tbest = 1e30f;
          hipError_t perr = hipSuccess;
          for (int rep = 0; rep < 2 && perr == hipSuccess; ++rep) {
            float pms = 0;
            if ((perr = hipEventRecord(e->ev0, e->stream)) != hipSuccess) break;
            hipLaunchKernelGGL(arena_probe_kernel, dim3(nslots), dim3(64), 0, e->stream, cand[c], ring_stride, (int)((size_t)wc * esz), 24,
                               std::min(1400, wc - 2048), e->d_counters.p);
            if ((perr = hipEventRecord(e->ev1, e->stream)) != hipSuccess) break;
            if ((perr = hipEventSynchronize(e->ev1)) != hipSuccess) break;
            if ((perr = hipEventElapsedTime(&pms, e->ev0, e->ev1)) != hipSuccess) break;
            tbest = std::min(tbest, pms);
          }
          if (perr != hipSuccess) {  // nothing may stay allocated behind an error return
            for (int k = 0; k < 4; ++k)
              if (cand[k]) (void)hipFree(cand[k]);
            return fail(AWV_ERR_HIP, std::string("ring arena probe: ") + hipGetErrorString(perr));
          }
          cms[c] = tbest;
          if (best < 0 || tbest < cms[best]) best = c;
          if (timing) fprintf(stderr, "[awv] ring arena candidate %d at %p: %.2f ms\n", c, (void*)cand[c], tbest);
        }
        if (best < 0) return fail(AWV_ERR_OOM, "hipMalloc " + std::to_string(want) + " bytes for the ring arena failed");
        for (int c = 0; c < 4; ++c)
          if (c != best && cand[c]) (void)hipFree(cand[c]);
        e->ring_mem.adopt(cand[best], want);
      }
      if (int rc = e->hist_mem.reserve(hist_stride * nslots)) return rc;
      if (int rc = e->ev_mem.reserve(ev_stride * nslots)) return rc;
      if (reserve_only) return AWV_OK;  // (first pass over the groups: one allocation covers them all)
      lap("arenas reserved");
      if (timing) fprintf(stderr, "[awv] ring arena %p (%zu MiB), hist %p, cigar %p\n", (void*)e->ring_mem.p, e->ring_mem.bytes() >> 20, (void*)e->hist_mem.p, (void*)e->d_cigar.p);
      // ---- H2D
      HIP_TRY(hipEventRecord(e->ev0, e->stream));
      HIP_TRY(hipMemcpyAsync(e->d_pair_q.p, hq.data(), (size_t)m * 4, hipMemcpyHostToDevice, e->stream));
      HIP_TRY(hipMemcpyAsync(e->d_pair_t.p, ht.data(), (size_t)m * 4, hipMemcpyHostToDevice, e->stream));
      HIP_TRY(hipMemcpyAsync(e->d_pair_rc.p, hrc.data(), (size_t)m * 4, hipMemcpyHostToDevice, e->stream));
      HIP_TRY(hipMemcpyAsync(e->d_cigar_off.p, hoff.data(), (size_t)m * 8, hipMemcpyHostToDevice, e->stream));
      HIP_TRY(hipMemsetAsync(e->d_counters.p, 0, (1 + STAT_N) * sizeof(unsigned long long), e->stream));
      HIP_TRY(hipEventRecord(e->ev1, e->stream));
      HIP_TRY(hipEventSynchronize(e->ev1));
      HIP_TRY(hipEventElapsedTime(&ms, e->ev0, e->ev1));
      h2d_ms += ms;
      // ---- launch
      KParams kp{};
      for (int v = 0; v < 4; ++v) kp.seq[v] = s.d_seq[v].p;
      kp.seq_off = s.d_off.p;
      kp.seq_len = s.d_len.p;
      kp.seq2[0] = s.d_seq2[0].p;
      kp.seq2[1] = s.d_seq2[1].p;
      kp.seq2_off = s.d_off2.p;
      kp.seq2_ok = s.d_ok2.p;
      kp.pair_q = e->d_pair_q.p;
      kp.pair_t = e->d_pair_t.p;
      kp.pair_rc = e->d_pair_rc.p;
      kp.npairs = m;
      kp.pen = dp;
      kp.ring = ring;
      // 32-bit rows: sweeps of at most TMAX32 scores, at most CHAIN_MAX32 of them chained (a lane vector is four registers)
      kp.multi_T = narrow ? multi_T : (std::min(multi_T, awv::TMAX32) >= 2 ? std::min(multi_T, awv::TMAX32) : 0);
      kp.deep_passes = (kp.multi_T > 0 && !(e->cfg.flags & AWV_F_NO_DEEP)) ? 1 : 0;
      kp.sub16 = (e->cfg.flags & AWV_F_FORCE_INT32) ? 0 : 1;  // (the pin means 32-bit rows throughout)
      // (32-bit rows: two sweeps chained through registers, a third when the kernel finds room for its chain rows in LDS -- the
      // kernel caps what it is offered: biwfa_device.hpp, AWV_LDS_CHAIN)
      kp.chain_max = narrow ? chain_max : (awv::TMAX32 == awv::TMAX ? std::min(chain_max, 3) : 1);
      kp.wcap = wc;
      kp.ring_mem = e->ring_mem.p;
      kp.ring_slot_stride = ring_stride;
      kp.lds_meta_bytes = (int)lds_meta;
      kp.lds_seq_bytes = (int)lds_seq;
      kp.sb_cap = sb_cap;
      kp.wb_cap = wb_cap;
      kp.hist_mem = e->hist_mem.p;
      kp.hist_slot_stride = hist_stride;
      kp.hist_meta_offset = hist_rows;
      kp.ev_mem = e->ev_mem.p;
      kp.ev_slot_stride = ev_stride;
      kp.cigar = e->d_cigar.p;
      kp.cigar_off = e->d_cigar_off.p;
      kp.results = e->d_results.p;
      kp.work_counter = e->d_counters.p;
      kp.stats = e->d_counters.p + 1;
      HIP_TRY(hipEventRecord(e->ev0, e->stream));
      auto launch = [&](auto kern, const auto& kparams) -> int {
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds));
        hipLaunchKernelGGL(kern, dim3(nslots), dim3(wg), dyn_lds, e->stream, kparams);
        return AWV_OK;
      };
      int lrc;
      static_assert(sizeof(awvw::KParams) == sizeof(awv::KParams) && sizeof(awvx::KParams) == sizeof(awv::KParams),
                    "same parameter block for every workgroup size");
      if (waves == 1) {
        HIP_TRY((hipError_t)awv_launch_one_wave(dp.two_piece, narrow ? 1 : 0, (unsigned)nslots, dyn_lds, e->stream, &kp));
        lrc = AWV_OK;
      } else if (wide_meta) {
        static_assert(sizeof(awvw_m::KParams) == sizeof(awv::KParams) && sizeof(awvx_m::KParams) == sizeof(awv::KParams), "same parameter block");
        if (waves == 4) {
          awvw_m::KParams kw;
          std::memcpy(&kw, &kp, sizeof(kw));
          lrc = dp.two_piece ? launch(awvw_m::biwfa_align_kernel<true, int16_t>, kw) : launch(awvw_m::biwfa_align_kernel<false, int16_t>, kw);
        } else {
          awvx_m::KParams kx;
          std::memcpy(&kx, &kp, sizeof(kx));
          lrc = dp.two_piece ? launch(awvx_m::biwfa_align_kernel<true, int16_t>, kx) : launch(awvx_m::biwfa_align_kernel<false, int16_t>, kx);
        }
      } else if (waves == 4) {
        awvw::KParams kw;
        std::memcpy(&kw, &kp, sizeof(kw));
        if (dp.two_piece) lrc = narrow ? launch(awvw::biwfa_align_kernel<true, int16_t>, kw) : launch(awvw::biwfa_align_kernel<true, int32_t>, kw);
        else lrc = narrow ? launch(awvw::biwfa_align_kernel<false, int16_t>, kw) : launch(awvw::biwfa_align_kernel<false, int32_t>, kw);
      } else {
        awvx::KParams kx;
        std::memcpy(&kx, &kp, sizeof(kx));
        if (dp.two_piece) lrc = narrow ? launch(awvx::biwfa_align_kernel<true, int16_t>, kx) : launch(awvx::biwfa_align_kernel<true, int32_t>, kx);
        else lrc = narrow ? launch(awvx::biwfa_align_kernel<false, int16_t>, kx) : launch(awvx::biwfa_align_kernel<false, int32_t>, kx);
      }
      if (lrc != AWV_OK) return lrc;
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipEventRecord(e->ev1, e->stream));
      // while the kernel runs: get the host CIGAR buffer's pages in place (0.2 s for config 2's 670 MB)
      if (sink && !(e->cfg.flags & AWV_F_KEEP_ON_DEVICE) && e->h_cigar.size() < (size_t)arena + 64) e->h_cigar.resize((size_t)arena + 64);
      HIP_TRY(hipEventSynchronize(e->ev1));
      HIP_TRY(hipEventElapsedTime(&ms, e->ev0, e->ev1));
      kernel_ms += ms;
      ++launches;
      lap("kernel done");
      // ---- D2H (results + counters)
      tres.resize((size_t)m);
      HIP_TRY(hipEventRecord(e->ev0, e->stream));
      HIP_TRY(hipMemcpyAsync(tres.data(), e->d_results.p, (size_t)m * sizeof(awv_result), hipMemcpyDeviceToHost, e->stream));
      unsigned long long hstat[1 + STAT_N];
      HIP_TRY(hipMemcpyAsync(hstat, e->d_counters.p, sizeof(hstat), hipMemcpyDeviceToHost, e->stream));
      HIP_TRY(hipEventRecord(e->ev1, e->stream));
      HIP_TRY(hipEventSynchronize(e->ev1));
      HIP_TRY(hipEventElapsedTime(&ms, e->ev0, e->ev1));
      d2h_ms += ms;
      for (int i = 0; i < STAT_N; ++i) stat_tot[i] += hstat[1 + i];
      // ---- scatter; collect the pairs to re-run with wider rows
      std::vector<int64_t> again;
      for (int64_t i = 0; i < m; ++i) {
        const int64_t bi = amap.empty() ? i : amap[(size_t)i];
        hres[(size_t)bi] = tres[(size_t)i];
        if (tres[(size_t)i].status == AWV_ST_CAPACITY && wc < wcap_full && !(e->cfg.flags & AWV_F_NO_RERUN)) again.push_back(i);
      }
      if (again.empty()) break;
      std::vector<int32_t> q2, t2, rc2;
      std::vector<uint64_t> off2;
      std::vector<int64_t> map2;
      for (int64_t i : again) {
        q2.push_back(hq[(size_t)i]);
        t2.push_back(ht[(size_t)i]);
        rc2.push_back(hrc[(size_t)i]);
        off2.push_back(hoff[(size_t)i]);
        map2.push_back(amap.empty() ? i : amap[(size_t)i]);
      }
      hq.swap(q2); ht.swap(t2); hrc.swap(rc2); hoff.swap(off2); amap.swap(map2);
      wc = (int)std::min<long long>(wcap_full, 4LL * wc);
    }
    return AWV_OK;
    };
    {
      // One group = one kernel flavour x one row width.  Flavour: all of a small batch goes four waves per pair
      // (latency), and so do pairs of long sequences (rows tens of windows wide; measured +17 % on 100 kbp
      // pairs) and pairs a length difference makes expensive -- pair by pair, so that the short pairs of a
      // mixed set (config 5) keep the throughput flavour.  Row width: 16-bit rows when both lengths fit.
      const bool never_wide = (e->cfg.flags & AWV_F_ONE_WAVE) != 0;
      // (a batch of uneven pairs that fits the machine about once is bound by its longest pairs, not by throughput)
      const bool all_wide = !never_wide && ((e->cfg.flags & AWV_F_FOUR_WAVES) || n <= (int64_t)(WAVES_PER_SIMD * e->num_cus) ||
                                            (skewed && n <= (int64_t)(4 * WAVES_PER_SIMD * e->num_cus)));
      constexpr int NG = 9;  // group = flavour (0 one wave, 1 four, 2 sixteen) * 3 + row width (0: 16-bit, 1: 32-bit, 2: 16-bit min(h, v) rows)
      std::vector<int32_t> q[NG], t[NG], rc[NG];
      std::vector<uint64_t> off[NG];
      std::vector<int64_t> map[NG];
      int gsum[NG] = {0}, glen[NG] = {0};
      // sixteen waves per pair only pay while such pairs are too few to fill the machine four waves at a time
      int64_t n_huge = 0;
      for (int64_t i = 0; i < n; ++i) n_huge += std::abs(s.len[hq[(size_t)i]] - s.len[ht[(size_t)i]]) >= 16384;
      const bool use_sixteen = !never_wide && !(e->cfg.flags & AWV_F_FOUR_WAVES) && n_huge > 0 && n_huge <= (int64_t)e->num_cus;
      const bool force32 = (e->cfg.flags & AWV_F_FORCE_INT32) != 0;
      const bool no_wide16 = (e->cfg.flags & AWV_F_NO_WIDE16) != 0;
      std::vector<uint8_t> fl((size_t)n);
      int64_t n_one = 0, n_four = 0;
      for (int64_t i = 0; i < n; ++i) {
        const int ql = s.len[hq[(size_t)i]], tl = s.len[ht[(size_t)i]];
        const int dl = std::abs(ql - tl);
        int f = (all_wide || (!never_wide && (dl >= 4096 || std::max(ql, tl) >= 32760))) ? 1 : 0;
        if (use_sixteen && dl >= 16384) f = 2;  // a forced gap that long: rows hundreds of windows wide
#ifdef AWV_DEBUG_KNOBS
        if (getenv("AWV_FORCE_SIXTEEN") && std::max(ql, tl) >= atoi(getenv("AWV_FORCE_SIXTEEN"))) f = 2;  // experiment: sixteen waves per pair for sequences at least that long
#endif
        fl[(size_t)i] = (uint8_t)f;
        n_one += f == 0;
        n_four += f == 1;
      }
      // One-wave pairs that cannot fill the machine even once, next to pairs that go four waves anyway, are bound by their
      // most expensive pair on a single wave while most of the machine idles: when their costs are uneven they go four
      // waves too (config 5: 3,315 such pairs, 2.9 s at 61 % of the CUs busy).  (Pairs come in descending cost order.)
      if (!never_wide && n_one > 0 && n_four > 0 && n_one <= (int64_t)(WAVES_PER_SIMD * 256 / AWV_THRU_WG) * e->num_cus) {
        auto cost_of = [&](int64_t i) {
          const int64_t ql = s.len[hq[(size_t)i]], tl = s.len[ht[(size_t)i]];
          return (uint64_t)(ql + tl + 4 * std::llabs(ql - tl));
        };
        int64_t first_one = -1, seen = 0, median_one = -1;
        for (int64_t i = 0; i < n && median_one < 0; ++i) {
          if (fl[(size_t)i] != 0) continue;
          if (first_one < 0) first_one = i;
          if (seen++ == n_one / 2) median_one = i;
        }
        if (2 * cost_of(first_one) >= 3 * cost_of(median_one))  // (work grows with the square of this length measure: 1.5 x = more than twice the work)
          for (int64_t i = 0; i < n; ++i)
            if (fl[(size_t)i] == 0) fl[(size_t)i] = 1;
      }
      for (int64_t i = 0; i < n; ++i) {
        const int ql = s.len[hq[(size_t)i]], tl = s.len[ht[(size_t)i]];
        const int f = fl[(size_t)i];
        // row width: every stored value must fit 16 bits -- text offsets (both lengths short), or min(h, v) when only
        // the shorter sequence is (the kernels for that exist in the four- and sixteen-wave flavours)
        int w = force32 || std::max(ql, tl) >= 32760 ? 1 : 0;
        if (w == 1 && !force32 && !no_wide16 && f >= 1 && std::min(ql, tl) < 32760) w = 2;
        const int g = f * 3 + w;
        q[g].push_back(hq[(size_t)i]); t[g].push_back(ht[(size_t)i]); rc[g].push_back(hrc[(size_t)i]); off[g].push_back(hoff[(size_t)i]);
        map[g].push_back(amap.empty() ? i : amap[(size_t)i]);
        gsum[g] = std::max(gsum[g], ql + tl);
        glen[g] = std::max(glen[g], std::max(ql, tl));
      }
      static const int waves_of[3] = {1, 4, 16};
      // the arenas are sized for the most demanding group first: growing them between two groups would
      // mean a free followed by a large allocation, which the driver can take seconds over
      {  // (reserve in descending order of estimated ring demand; DevBuf::reserve only ever grows)
        std::pair<size_t, int> demand[NG];
        for (int g = 0; g < NG; ++g) {
          demand[g] = {0, g};
          if (q[g].empty()) continue;
          const int wv = waves_of[g / 3];
          const int wgx = wv == 1 ? AWV_THRU_WG : 64 * wv;
          const int64_t slots = std::min<int64_t>((WAVES_PER_SIMD * 256 / wgx) * e->num_cus, (int64_t)q[g].size());
          demand[g].first = (size_t)slots * (size_t)gsum[g] * ((g % 3) == 1 ? 4 : 2);
        }
        std::sort(demand, demand + NG, [](const std::pair<size_t, int>& a, const std::pair<size_t, int>& b) { return a.first > b.first; });
        for (int k = 0; k < NG; ++k) {
          const int g = demand[k].second;
          if (q[g].empty()) continue;
          if (int rcg = run_group(q[g], t[g], rc[g], off[g], map[g], waves_of[g / 3], g % 3, gsum[g], glen[g], true)) return rcg;
        }
      }
      for (int g = NG - 1; g >= 0; --g)
        if (int rcg = run_group(std::move(q[g]), std::move(t[g]), std::move(rc[g]), std::move(off[g]), std::move(map[g]), waves_of[g / 3], g % 3, gsum[g], glen[g], false)) return rcg;
    }
    const bool want_cigar = sink && !(e->cfg.flags & AWV_F_KEEP_ON_DEVICE);
    lap("results on host");
    if (want_cigar) {
      e->h_cigar.resize((size_t)arena + 64);
      lap("host cigar buffer");
      HIP_TRY(hipEventRecord(e->ev0, e->stream));
      HIP_TRY(hipMemcpyAsync(e->h_cigar.data(), e->d_cigar.p, (size_t)arena, hipMemcpyDeviceToHost, e->stream));
      HIP_TRY(hipEventRecord(e->ev1, e->stream));
      HIP_TRY(hipEventSynchronize(e->ev1));
      float ms_c = 0;
      HIP_TRY(hipEventElapsedTime(&ms_c, e->ev0, e->ev1));
      d2h_ms += ms_c;
    }
    lap("cigars on host");
    if (out) std::memcpy(out + first, hres.data(), (size_t)n * sizeof(awv_result));
    if (sink) {
      if (const int prc = runner.wait()) return fail(AWV_ERR_SINK, "sink callback returned " + std::to_string(prc));
      if (first + n >= npairs || inline_sink) {  // the last (or only) batch: on the calling thread
        const int rc = sink(user, first, n, hres.data(), want_cigar ? e->h_cigar.data() : nullptr);
        if (rc != 0) return fail(AWV_ERR_SINK, "sink callback returned " + std::to_string(rc));
        lap("sink returned");
      } else {  // hand the batch's buffers to the helper thread and go on with the next batch
        runner.res.swap(hres);
        runner.cig.swap(e->h_cigar);
        const awv_result* rp = runner.res.data();
        const uint8_t* cp = want_cigar ? runner.cig.data() : nullptr;
        const int64_t f0 = first, n0 = n;
        SinkRunner* rn = &runner;
        try {
          runner.th = std::thread([rn, sink, user, f0, n0, rp, cp]() { rn->rc = sink(user, f0, n0, rp, cp); });
          lap("sink handed off");
        } catch (const std::exception&) {  // no thread to be had: this sink runs here, nothing may cross the C boundary
          const int rc = sink(user, f0, n0, rp, cp);
          if (rc != 0) return fail(AWV_ERR_SINK, "sink callback returned " + std::to_string(rc));
        }
      }
    }
    first += n;
  }
  e->stats.kernel_ms = kernel_ms;
  e->stats.h2d_ms = h2d_ms;
  e->stats.d2h_ms = d2h_ms;
  e->stats.launches = launches;
  e->stats.cell_steps = stat_tot[STAT_CELLS];
  e->stats.extend_steps = stat_tot[STAT_EXTEND];
  e->stats.n_breakpoints = stat_tot[STAT_BREAKPOINTS];
  e->stats.n_base = stat_tot[STAT_BASE];
  e->stats.overlap_scans = stat_tot[STAT_OVERLAP];
  e->stats.aligned_bp = stat_tot[STAT_ALIGNED_BP];
  e->stats.pairs_completed = stat_tot[STAT_PAIRS];
  e->stats.scratch_bytes = e->ring_mem.bytes() + e->hist_mem.bytes() + e->ev_mem.bytes();
  for (int i = 0; i < 14; ++i) e->stats.prof[i] = stat_tot[STAT_T_TOTAL + i];
  e->stats.restarts = stat_tot[STAT_RESTARTS];
  e->stats.multi_cell_steps = stat_tot[STAT_MULTI_CELLS];
  e->stats.windows[0] = stat_tot[STAT_WIN_SINGLE];
  e->stats.windows[1] = stat_tot[STAT_WIN_MULTI];
  e->stats.windows[2] = stat_tot[STAT_WIN_BASE];
  e->stats.windows[3] = stat_tot[STAT_WIN_BASE_MULTI];
  e->stats.clock_cycles = stat_tot[STAT_CLK_CYCLES];
  e->stats.clock_ticks = stat_tot[STAT_CLK_TICKS];
  e->stats.clock_tick_khz = (uint64_t)e->wall_clock_khz;
  e->stats.deep_cell_steps = stat_tot[STAT_DEEP_CELLS];
  return AWV_OK;
}

}  // namespace

extern "C" {

int awv_abi_version(void) { return AWV_ABI_VERSION; }

const char* awv_last_error(void) { return g_last_error.c_str(); }

int awv_engine_create(const awv_engine_config* cfg, awv_engine** out) {
  if (!out) return fail(AWV_ERR_ARG, "engine_create: null out");
  *out = nullptr;
  int ndev = 0;
  hipError_t err = hipGetDeviceCount(&ndev);
  if (err != hipSuccess || ndev <= 0)
    return fail(AWV_ERR_NO_DEVICE, std::string("no HIP device available (") + hipGetErrorString(err) +
                                       "): liballwave_hip has no CPU fallback");
  awv_engine* e = new awv_engine();
  if (cfg) e->cfg = *cfg;
  e->device = e->cfg.device;
  if (e->device < 0 || e->device >= ndev) {
    delete e;
    return fail(AWV_ERR_ARG, "engine_create: device ordinal out of range");
  }
  auto bail = [&](hipError_t he, const char* what) {
    std::string m = std::string(what) + ": " + hipGetErrorString(he);
    awv_engine_destroy(e);
    return fail(AWV_ERR_HIP, m);
  };
  hipError_t he;
  if ((he = hipSetDevice(e->device)) != hipSuccess) return bail(he, "hipSetDevice");
  hipDeviceProp_t prop;
  if ((he = hipGetDeviceProperties(&prop, e->device)) != hipSuccess) return bail(he, "hipGetDeviceProperties");
  e->num_cus = prop.multiProcessorCount;
  {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, e->device) == hipSuccess && khz > 0) e->wall_clock_khz = khz;
    else (void)hipGetLastError();
  }
  if ((he = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking)) != hipSuccess) return bail(he, "hipStreamCreate");
  if ((he = hipEventCreate(&e->ev0)) != hipSuccess) return bail(he, "hipEventCreate");
  if ((he = hipEventCreate(&e->ev1)) != hipSuccess) return bail(he, "hipEventCreate");
  *out = e;
  return AWV_OK;
}

void awv_engine_destroy(awv_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  e->seqs.release();
  e->ring_mem.release();
  e->hist_mem.release();
  e->ev_mem.release();
  e->d_pair_q.release();
  e->d_pair_t.release();
  e->d_pair_rc.release();
  e->d_cigar_off.release();
  e->d_results.release();
  e->d_cigar.release();
  e->d_counters.release();
  if (e->ev0) (void)hipEventDestroy(e->ev0);
  if (e->ev1) (void)hipEventDestroy(e->ev1);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

// Nothing C++ may cross the C boundary: a std::bad_alloc / std::length_error out of a host-side container (a caller that hands
// over billions of pairs, a box short of memory) would otherwise reach the foreign caller as std::terminate -> abort().
#define AWV_GUARDED(body)                                                                                         \
  try {                                                                                                           \
    body                                                                                                          \
  } catch (const std::bad_alloc&) {                                                                               \
    return fail(AWV_ERR_OOM, "host memory exhausted");                                                            \
  } catch (const std::exception& ex) {                                                                            \
    return fail(AWV_ERR_HIP, std::string("internal error: ") + ex.what());                                        \
  } catch (...) {                                                                                                 \
    return fail(AWV_ERR_HIP, "internal error: unknown exception");                                                \
  }

int awv_engine_set_sequences(awv_engine* e, int32_t n, const uint8_t* concat_bytes, const uint64_t* offsets) {
  if (!e) return fail(AWV_ERR_ARG, "null engine");
  AWV_GUARDED(return upload_seqset(e, e->seqs, n, concat_bytes, offsets);)
}

int awv_align_pairs(awv_engine* e, const awv_penalties* pen, const awv_pair* pairs, int64_t npairs, awv_result* out,
                    awv_sink sink, void* user) {
  if (!e) return fail(AWV_ERR_ARG, "null engine");
  if (e->seqs.n == 0 && npairs > 0) return fail(AWV_ERR_STATE, "align_pairs before set_sequences");
  AWV_GUARDED(return align_core(e, e->seqs, pen, pairs, npairs, out, sink, user);)
}

namespace {
int align_one_core(awv_engine* e, const awv_penalties* pen, const uint8_t* pattern, int32_t plen, const uint8_t* text,
                   int32_t tlen, awv_result* result, uint8_t* cigar_buf, size_t cigar_cap);
struct OneSink {
  uint8_t* buf;
  size_t cap;
  int rc;
};
int one_sink(void* user, int64_t, int64_t n, const awv_result* r, const uint8_t* arena) {
  OneSink* o = (OneSink*)user;
  if (n != 1 || !arena) return 1;
  if (r[0].status == AWV_ST_COMPLETED) {
    if (r[0].cigar_len > o->cap) { o->rc = AWV_ERR_ARG; return 2; }
    std::memcpy(o->buf, arena + r[0].cigar_off, r[0].cigar_len);
  }
  return 0;
}
}  // namespace

int awv_align_one(awv_engine* e, const awv_penalties* pen, const uint8_t* pattern, int32_t plen, const uint8_t* text,
                  int32_t tlen, awv_result* result, uint8_t* cigar_buf, size_t cigar_cap) {
  if (!e || !result || plen < 0 || tlen < 0 || (plen > 0 && !pattern) || (tlen > 0 && !text))
    return fail(AWV_ERR_ARG, "align_one: bad argument");
  if (cigar_cap < (size_t)plen + (size_t)tlen) return fail(AWV_ERR_ARG, "align_one: cigar buffer needs plen + tlen bytes");
  AWV_GUARDED(return align_one_core(e, pen, pattern, plen, text, tlen, result, cigar_buf, cigar_cap);)
}
}  // extern "C"

namespace {
int align_one_core(awv_engine* e, const awv_penalties* pen, const uint8_t* pattern, int32_t plen, const uint8_t* text,
                   int32_t tlen, awv_result* result, uint8_t* cigar_buf, size_t cigar_cap) {
  struct Scope {  // the temporary sequence set and the engine's flags go back whichever way this function is left
    awv_engine* e;
    int32_t saved_flags;
    SeqSet tmp;
    ~Scope() {
      e->cfg.flags = saved_flags;
      tmp.release();
    }
  } sc{e, e->cfg.flags, {}};
  std::vector<uint8_t> cat((size_t)plen + (size_t)tlen + 1);
  if (plen) std::memcpy(cat.data(), pattern, (size_t)plen);
  if (tlen) std::memcpy(cat.data() + plen, text, (size_t)tlen);
  const uint64_t offs[3] = {0, (uint64_t)plen, (uint64_t)plen + (uint64_t)tlen};
  int rc = upload_seqset(e, sc.tmp, 2, cat.data(), offs);
  if (rc == AWV_OK) {
    const awv_pair p{0, 1, 0};
    OneSink os{cigar_buf, cigar_cap, AWV_OK};
    e->cfg.flags &= ~AWV_F_KEEP_ON_DEVICE;
    rc = align_core(e, sc.tmp, pen, &p, 1, result, one_sink, &os);
    if (rc == AWV_ERR_SINK && os.rc != AWV_OK) rc = fail(os.rc, "align_one: cigar buffer too small");
  }
  return rc;
}
}  // namespace

extern "C" {

int awv_engine_stats(const awv_engine* e, awv_stats* out) {
  if (!e || !out) return fail(AWV_ERR_ARG, "stats: null argument");
  *out = e->stats;
  return AWV_OK;
}

}  // extern "C"
