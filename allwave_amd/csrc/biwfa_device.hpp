// biwfa_device.hpp -- hand-written HIP (gfx950 / CDNA4) kernels for allwave's per-pair hot path:
// end-to-end BiWFA under gap-affine / 2-piece gap-affine penalties with full CIGAR.
//
// Replaces what the reference does inside `wf.align(query, target)`
// (/root/reference/src/alignment.rs:231, src/wfa.rs:226) -> WFA2-lib [not in the container;
// semantics per SURVEY.md Appendix A].  Results are bit-exact against oracle/biwfa_oracle.c.
//
// Mapping onto the machine (integer DP: no MFMA):
//   * one sequence pair per 256-thread workgroup (4 wave64), persistent workgroups pull pairs
//     from an atomic cursor; the BiWFA recursion is an explicit DFS stack in LDS, so CIGAR ops
//     come out in order and no device recursion is needed;
//   * lanes <-> diagonals: a wave owns 64 consecutive diagonals ("chunk") per iteration, all
//     row loads/stores are coalesced; forward and reverse column spaces are mirrored on chunk
//     boundaries (colR = C - colF, C == 63 mod 64) so the meet-in-the-middle overlap test reads
//     both wavefronts coalesced;
//   * wavefront rows live in a per-workgroup arena in HBM (ring of `ring` rows per component
//     and direction; L2-resident when hot), row metadata (lo/hi, max antidiagonal) in LDS;
//   * trimming (first/last in-bounds diagonal) by wave ballots + LDS atomics, max-antidiagonal
//     by a wave reduction; one workgroup barrier per score step, forward and reverse steps fused.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>

namespace awv {

constexpr int WG = 256;
constexpr int MAX_RING = 128;
constexpr int NCOMP = 5;
constexpr int32_t OFF_NULL = INT32_MIN / 2;
constexpr int32_t NULLISH = INT32_MIN / 4;    // any value below is a NULL(+n)
enum { C_M = 0, C_I1 = 1, C_I2 = 2, C_D1 = 3, C_D2 = 4 };
constexpr int FALLBACK_MIN_SCORE = 250;
constexpr int FALLBACK_MIN_LENGTH = 100;
constexpr int STACK_CAP = 192;

// per-pair status (allwave_hip.h AWV_ST_*)
constexpr int ST_OK = 0, ST_CAPACITY = 1, ST_INTERNAL = 2, ST_MAX_STEPS = 3;

struct DevPenalties {
  int x, o1, e1, o2, e2, two_piece, scope;  // scope = max(x, o1+e1, o2+e2) + 1  (A.3)
};

struct DevResult {  // mirrors awv_result
  int32_t status, penalty, score;
  uint32_t cigar_len;
  uint64_t cigar_off;
  int32_t num_matches, num_mismatches, num_ins, num_del, q_end, t_end;
};

enum { STAT_CELLS = 0, STAT_EXTEND, STAT_BREAKPOINTS, STAT_BASE, STAT_OVERLAP, STAT_ALIGNED_BP, STAT_PAIRS, STAT_N };

struct KParams {
  const uint8_t* seq[4];  // 0 fwd, 1 reversed, 2 reverse-complement, 3 reversed reverse-complement
  const uint64_t* seq_off;
  const int32_t* seq_len;
  const int32_t* pair_q;
  const int32_t* pair_t;
  const int32_t* pair_rc;
  long long npairs;
  DevPenalties pen;
  int ring;        // power of two >= scope + 2
  int wcap;        // columns per ring row
  int32_t* ring_mem;
  size_t ring_slot_stride;  // int32 elements per workgroup slot
  int sb_cap;      // base-case score capacity
  int wb_cap;      // base-case columns per row
  int32_t* hist_mem;
  size_t hist_slot_stride;
  uint32_t* ev_mem;
  size_t ev_slot_stride;
  uint8_t* cigar;
  const uint64_t* cigar_off;
  DevResult* results;
  unsigned long long* work_counter;
  unsigned long long* stats;
};

struct RowMeta { int lo, hi; };
struct Acc { int hull_lo[NCOMP]; int hull_hi[NCOMP]; int maxak; int oob; };
struct Task { int pb, pe, tb, te, cb, ce, score_remaining; };
struct Breakpoint { int score, sf, sr, kf, kr, off_f, off_r, comp; };

struct Shared {
  RowMeta bi_meta[2][NCOMP][MAX_RING];
  int bi_A[2][MAX_RING];
  int bi_oob[2][MAX_RING];
  Acc acc[3][2];
  int firstk[MAX_RING * NCOMP];
  Task stack[STACK_CAP];
  int ext0[2];
  long long cur_pair;
  int error;
  int nev;
  int bt_total;
};

struct SubCtx {
  int plen, tlen;
  const uint8_t* P[2];
  const uint8_t* T[2];
  int kmin[2];
  int wcols;
};

__device__ __forceinline__ bool row_empty(const RowMeta& m) { return m.lo > m.hi; }

__device__ __forceinline__ uint64_t ld64u(const uint8_t* p) {
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

// bounded LCP of pattern[v..] / text[h..] (A.4), 8 bytes per iteration
__device__ __forceinline__ int extend_lcp(const uint8_t* P, const uint8_t* T, int v, int h, int plen, int tlen,
                                          unsigned& iters) {
  const int rem = min(plen - v, tlen - h);
  const uint8_t* pp = P + v;
  const uint8_t* tp = T + h;
  int n = 0;
  while (n < rem) {
    const uint64_t x = ld64u(pp + n) ^ ld64u(tp + n);
    ++iters;
    if (x) {
      n += (int)(__builtin_ctzll(x) >> 3);
      break;
    }
    n += 8;
  }
  return min(n, rem);
}

template <bool BASE>
__device__ __forceinline__ RowMeta get_meta(const KParams& kp, const Shared& sh, const RowMeta* base_meta, int dir,
                                            int comp, int score) {
  if (score < 0) return RowMeta{1, 0};
  if (BASE) return base_meta[score * NCOMP + comp];
  return sh.bi_meta[dir][comp][score & (kp.ring - 1)];
}

template <bool BASE>
__device__ __forceinline__ int32_t* row_ptr(const KParams& kp, int32_t* mem, int dir, int comp, int score) {
  if (score < 0) score = 0;  // null input rows are never dereferenced
  if (BASE) return mem + ((size_t)score * NCOMP + comp) * (size_t)kp.wb_cap;
  return mem + ((size_t)(dir * NCOMP + comp) * kp.ring + (size_t)(score & (kp.ring - 1))) * (size_t)kp.wcap;
}

__device__ __forceinline__ int32_t rd(const int32_t* p, const RowMeta& m, int k, int kmin) {
  return (k >= m.lo && k <= m.hi) ? p[k - kmin] : OFF_NULL;
}

__device__ __forceinline__ void acc_reset(Acc& a) {
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) { a.hull_lo[c] = INT_MAX; a.hull_hi[c] = INT_MIN; }
  a.maxak = 0;
  a.oob = 0;
}

// One compute-next + extend step of one direction (A.3 + A.4).  Output rows are written
// untrimmed; hull / max antidiagonal / oob land in `acc` (LDS atomics) and become the row
// metadata in finalize_row() after the workgroup barrier.  Returns the number of cells.
template <bool P2, bool BASE>
__device__ __forceinline__ int compute_row(const KParams& kp, Shared& sh, RowMeta* base_meta, const SubCtx& cx,
                                           int32_t* mem, int dir, int score, Acc& acc, unsigned& ext_iters) {
  const DevPenalties& pn = kp.pen;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kmin = cx.kmin[dir];
  const RowMeta mMx = get_meta<BASE>(kp, sh, base_meta, dir, C_M, score - pn.x);
  const RowMeta mO1 = get_meta<BASE>(kp, sh, base_meta, dir, C_M, score - pn.o1 - pn.e1);
  const RowMeta mI1 = get_meta<BASE>(kp, sh, base_meta, dir, C_I1, score - pn.e1);
  const RowMeta mD1 = get_meta<BASE>(kp, sh, base_meta, dir, C_D1, score - pn.e1);
  RowMeta mO2{1, 0}, mI2{1, 0}, mD2{1, 0};
  if (P2) {
    mO2 = get_meta<BASE>(kp, sh, base_meta, dir, C_M, score - pn.o2 - pn.e2);
    mI2 = get_meta<BASE>(kp, sh, base_meta, dir, C_I2, score - pn.e2);
    mD2 = get_meta<BASE>(kp, sh, base_meta, dir, C_D2, score - pn.e2);
  }
  int lo = INT_MAX, hi = INT_MIN;
  if (!row_empty(mMx)) { lo = min(lo, mMx.lo); hi = max(hi, mMx.hi); }
  if (!row_empty(mO1)) { lo = min(lo, mO1.lo - 1); hi = max(hi, mO1.hi + 1); }
  if (!row_empty(mI1)) { lo = min(lo, mI1.lo + 1); hi = max(hi, mI1.hi + 1); }
  if (!row_empty(mD1)) { lo = min(lo, mD1.lo - 1); hi = max(hi, mD1.hi - 1); }
  if (P2) {
    if (!row_empty(mO2)) { lo = min(lo, mO2.lo - 1); hi = max(hi, mO2.hi + 1); }
    if (!row_empty(mI2)) { lo = min(lo, mI2.lo + 1); hi = max(hi, mI2.hi + 1); }
    if (!row_empty(mD2)) { lo = min(lo, mD2.lo - 1); hi = max(hi, mD2.hi - 1); }
  }
  if (lo > hi) return 0;  // null step: acc stays reset -> all rows empty
  if (lo - 1 < kmin || hi + 1 > kmin + cx.wcols - 1) {
    sh.error = ST_CAPACITY;
    return 0;
  }
  const int32_t* pMx = row_ptr<BASE>(kp, mem, dir, C_M, score - pn.x);
  const int32_t* pO1 = row_ptr<BASE>(kp, mem, dir, C_M, score - pn.o1 - pn.e1);
  const int32_t* pI1 = row_ptr<BASE>(kp, mem, dir, C_I1, score - pn.e1);
  const int32_t* pD1 = row_ptr<BASE>(kp, mem, dir, C_D1, score - pn.e1);
  const int32_t* pO2 = P2 ? row_ptr<BASE>(kp, mem, dir, C_M, score - pn.o2 - pn.e2) : nullptr;
  const int32_t* pI2 = P2 ? row_ptr<BASE>(kp, mem, dir, C_I2, score - pn.e2) : nullptr;
  const int32_t* pD2 = P2 ? row_ptr<BASE>(kp, mem, dir, C_D2, score - pn.e2) : nullptr;
  int32_t* oM = row_ptr<BASE>(kp, mem, dir, C_M, score);
  int32_t* oI1 = row_ptr<BASE>(kp, mem, dir, C_I1, score);
  int32_t* oD1 = row_ptr<BASE>(kp, mem, dir, C_D1, score);
  int32_t* oI2 = P2 ? row_ptr<BASE>(kp, mem, dir, C_I2, score) : nullptr;
  int32_t* oD2 = P2 ? row_ptr<BASE>(kp, mem, dir, C_D2, score) : nullptr;
  const uint8_t* Pp = cx.P[dir];
  const uint8_t* Tp = cx.T[dir];
  const int plen = cx.plen, tlen = cx.tlen;
  const int colLo = lo - kmin, colHi = hi - kmin;
  int wlo[NCOMP], whi[NCOMP];
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) { wlo[c] = INT_MAX; whi[c] = INT_MIN; }
  int lane_maxak = 0;
  bool lane_oob = false;
  for (int cb = (colLo & ~63) + 64 * wave; cb <= colHi; cb += WG) {
    const int col = cb + lane;
    const int k = col + kmin;
    const bool act = col >= colLo && col <= colHi;
    int32_t m = OFF_NULL, ins1 = OFF_NULL, del1 = OFF_NULL, ins2 = OFF_NULL, del2 = OFF_NULL;
    if (act) {
      const int32_t mx = rd(pMx, mMx, k, kmin);
      const int32_t o1l = rd(pO1, mO1, k - 1, kmin), o1r = rd(pO1, mO1, k + 1, kmin);
      const int32_t i1e = rd(pI1, mI1, k - 1, kmin), d1e = rd(pD1, mD1, k + 1, kmin);
      ins1 = max(o1l, i1e) + 1;
      del1 = max(o1r, d1e);
      int32_t ins = ins1, del = del1;
      if (P2) {
        const int32_t o2l = rd(pO2, mO2, k - 1, kmin), o2r = rd(pO2, mO2, k + 1, kmin);
        const int32_t i2e = rd(pI2, mI2, k - 1, kmin), d2e = rd(pD2, mD2, k + 1, kmin);
        ins2 = max(o2l, i2e) + 1;
        del2 = max(o2r, d2e);
        ins = max(ins1, ins2);
        del = max(del1, del2);
      }
      m = max(del, max(mx + 1, ins));
      if ((uint32_t)m > (uint32_t)tlen || (uint32_t)(m - k) > (uint32_t)plen) {
        lane_oob |= (m > NULLISH);
        m = OFF_NULL;
      }
      if (m >= 0) {
        m += extend_lcp(Pp, Tp, m - k, m, plen, tlen, ext_iters);
        lane_maxak = max(lane_maxak, 2 * m - k);
      }
      oM[col] = m;
      oI1[col] = ins1;
      oD1[col] = del1;
      if (P2) { oI2[col] = ins2; oD2[col] = del2; }
    }
    // trimming (A.3): first / last in-bounds diagonal per component, via ballots
    auto inb = [&](int32_t v) { return (uint32_t)v <= (uint32_t)tlen && (uint32_t)(v - k) <= (uint32_t)plen; };
    const bool bI1 = act && inb(ins1), bD1 = act && inb(del1);
    const bool bI2 = P2 && act && inb(ins2), bD2 = P2 && act && inb(del2);
    lane_oob |= act && ((!bI1 && ins1 > NULLISH) || (!bD1 && del1 > NULLISH));
    if (P2) lane_oob |= act && ((!bI2 && ins2 > NULLISH) || (!bD2 && del2 > NULLISH));
    const uint64_t masks[NCOMP] = {__ballot(act && m >= 0), __ballot(bI1), __ballot(bI2), __ballot(bD1), __ballot(bD2)};
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) {
      if (!P2 && (c == C_I2 || c == C_D2)) continue;
      if (masks[c]) {
        wlo[c] = min(wlo[c], cb + (int)__builtin_ctzll(masks[c]));
        whi[c] = max(whi[c], cb + 63 - (int)__builtin_clzll(masks[c]));
      }
    }
  }
  const int wmax = wave_max_i32(lane_maxak);
  const bool woob = __any(lane_oob);
  if (lane == 0) {
    atomicMax(&acc.maxak, wmax);
    if (woob) acc.oob = 1;
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) {
      if (wlo[c] != INT_MAX) {
        atomicMin(&acc.hull_lo[c], wlo[c]);
        atomicMax(&acc.hull_hi[c], whi[c]);
      }
    }
  }
  return hi - lo + 1;
}

// after the barrier: every thread writes the same trimmed metadata (benign same-value stores)
template <bool BASE>
__device__ __forceinline__ void finalize_row(const KParams& kp, Shared& sh, RowMeta* base_meta, const SubCtx& cx,
                                             int dir, int score, const Acc& acc) {
  const int kmin = cx.kmin[dir];
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) {
    RowMeta m{1, 0};
    const int l = acc.hull_lo[c], h = acc.hull_hi[c];
    if (l != INT_MAX) { m.lo = l + kmin; m.hi = h + kmin; }
    if (BASE) base_meta[score * NCOMP + c] = m;
    else sh.bi_meta[dir][c][score & (kp.ring - 1)] = m;
  }
  if (!BASE) {
    sh.bi_A[dir][score & (kp.ring - 1)] = acc.maxak;
    sh.bi_oob[dir][score & (kp.ring - 1)] = acc.oob;
  }
}

// ---------------------------------------------------------------------------------------------
// CIGAR emission helpers (workgroup-wide, uniform arguments)
// ---------------------------------------------------------------------------------------------
struct Emit {
  uint8_t* cig;
  int n;
  int cnt[4];  // M X I D
};
__device__ __forceinline__ int op_index(uint8_t op) { return op == 'M' ? 0 : op == 'X' ? 1 : op == 'I' ? 2 : 3; }
__device__ __forceinline__ void emit_run(Emit& em, uint8_t op, int len) {
  uint8_t* p = em.cig + em.n;
  for (int i = threadIdx.x; i < len; i += WG) p[i] = op;
  em.n += len;
  em.cnt[op_index(op)] += len;
}

// ---------------------------------------------------------------------------------------------
// Base case: plain WFA with full history + backtrace (A.5), wavefront_bialign_base
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int bt_fetch(const KParams& kp, const RowMeta* base_meta, const int32_t* hist, int kmin,
                                        int max_score, int comp, int score, int k, int add, int type) {
  if (score < 0 || score > max_score) return -1;
  const RowMeta m = base_meta[score * NCOMP + comp];
  if (k < m.lo || k > m.hi) return -1;
  const int32_t v = hist[((size_t)score * NCOMP + comp) * (size_t)kp.wb_cap + (k - kmin)];
  if (v < 0) return -1;
  return ((v + add) << 4) | type;
}

enum { BT_I1_OPEN = 1, BT_I1_EXT = 2, BT_I2_OPEN = 3, BT_I2_EXT = 4, BT_D1_OPEN = 5, BT_D1_EXT = 6, BT_D2_OPEN = 7, BT_D2_EXT = 8, BT_M = 9 };

template <bool P2>
__device__ int base_align(const KParams& kp, Shared& sh, RowMeta* base_meta, SubCtx cx, int32_t* hist, uint32_t* events,
                          int cb, int ce, Emit& em, int& penalty_out, unsigned long long* lstats) {
  const DevPenalties& pn = kp.pen;
  const int tid = threadIdx.x, lane = tid & 63;
  const int plen = cx.plen, tlen = cx.tlen;
  const int kspan_lo = min(plen, kp.sb_cap), kspan_hi = min(tlen, kp.sb_cap);
  cx.kmin[0] = -kspan_lo - 4;
  cx.wcols = kspan_lo + kspan_hi + 9;
  if (cx.wcols > kp.wb_cap) return ST_CAPACITY;
  const int kmin = cx.kmin[0];
  // score 0
  for (int c = tid; c < NCOMP; c += WG) base_meta[c] = (c == cb) ? RowMeta{0, 0} : RowMeta{1, 0};
  if (tid == 0) {
    unsigned it = 0;
    int v0 = 0;
    if (cb == C_M) v0 = extend_lcp(cx.P[0], cx.T[0], 0, 0, plen, tlen, it);
    hist[(size_t)cb * kp.wb_cap + (0 - kmin)] = v0;
    acc_reset(sh.acc[0][0]);
    acc_reset(sh.acc[1][0]);
    acc_reset(sh.acc[2][0]);
  }
  __syncthreads();
  const int k_end = tlen - plen;
  int score = 0;
  unsigned ext_iters = 0;
  unsigned long long cells = 0;
  int pass = 0;
  for (;;) {
    // termination (wavefront_termination_end2end): end component reaches (plen, tlen)
    const RowMeta me = base_meta[score * NCOMP + ce];
    if (k_end >= me.lo && k_end <= me.hi) {
      const int32_t v = hist[((size_t)score * NCOMP + ce) * (size_t)kp.wb_cap + (k_end - kmin)];
      if (v >= tlen) break;
    }
    ++score;
    if (score > kp.sb_cap) return ST_CAPACITY;
    Acc& acc = sh.acc[pass % 3][0];
    cells += compute_row<P2, true>(kp, sh, base_meta, cx, hist, 0, score, acc, ext_iters);
    __syncthreads();
    if (sh.error) return sh.error;
    finalize_row<true>(kp, sh, base_meta, cx, 0, score, acc);
    if (tid == 0) acc_reset(sh.acc[(pass + 2) % 3][0]);
    ++pass;
    __syncthreads();  // base_meta lives in dynamic LDS shared by all waves: make it visible
  }
  penalty_out = score;
  if (tid == 0) {
    lstats[STAT_CELLS] += cells;
    lstats[STAT_BASE] += 1;
  }
  atomicAdd(&lstats[STAT_EXTEND], (unsigned long long)ext_iters);
  // ---- backtrace by wave 0 (candidates fetched by lanes 0..8, packed (offset<<4)|type, max wins)
  if (tid < 64) {
    int matrix = ce, sc = score, k = k_end, offset = tlen;
    int h = offset, v = offset - k;
    int nev = 0, total = 0;
    int last_op = -1, last_cnt = 0;
    int err = 0;
    auto push = [&](int op, int n) {
      if (n <= 0) return;
      if (op == last_op) { last_cnt += n; }
      else {
        if (last_op >= 0) { if (lane == 0) events[nev] = ((uint32_t)last_cnt << 2) | (uint32_t)last_op; ++nev; }
        last_op = op; last_cnt = n;
      }
      total += n;
    };
    int guard = 0;
    while (v > 0 && h > 0 && sc > 0) {
      if (++guard > 4 * (plen + tlen) + 16) { err = ST_INTERNAL; break; }
      const int mismatch = sc - pn.x, gap_open1 = sc - pn.o1 - pn.e1, gap_extend1 = sc - pn.e1;
      const int gap_open2 = sc - pn.o2 - pn.e2, gap_extend2 = sc - pn.e2;
      int cand = -1;
      if (matrix == C_M) {
        switch (lane) {
          case 0: cand = bt_fetch(kp, base_meta, hist, kmin, score, C_M, mismatch, k, 1, BT_M); break;
          case 1: cand = bt_fetch(kp, base_meta, hist, kmin, score, C_M, gap_open1, k - 1, 1, BT_I1_OPEN); break;
          case 2: cand = bt_fetch(kp, base_meta, hist, kmin, score, C_I1, gap_extend1, k - 1, 1, BT_I1_EXT); break;
          case 3: cand = bt_fetch(kp, base_meta, hist, kmin, score, C_M, gap_open1, k + 1, 0, BT_D1_OPEN); break;
          case 4: cand = bt_fetch(kp, base_meta, hist, kmin, score, C_D1, gap_extend1, k + 1, 0, BT_D1_EXT); break;
          case 5: if (P2) cand = bt_fetch(kp, base_meta, hist, kmin, score, C_M, gap_open2, k - 1, 1, BT_I2_OPEN); break;
          case 6: if (P2) cand = bt_fetch(kp, base_meta, hist, kmin, score, C_I2, gap_extend2, k - 1, 1, BT_I2_EXT); break;
          case 7: if (P2) cand = bt_fetch(kp, base_meta, hist, kmin, score, C_M, gap_open2, k + 1, 0, BT_D2_OPEN); break;
          case 8: if (P2) cand = bt_fetch(kp, base_meta, hist, kmin, score, C_D2, gap_extend2, k + 1, 0, BT_D2_EXT); break;
          default: break;
        }
      } else if (matrix == C_I1) {
        if (lane == 0) cand = bt_fetch(kp, base_meta, hist, kmin, score, C_I1, gap_extend1, k - 1, 1, BT_I1_EXT);
        if (lane == 1) cand = bt_fetch(kp, base_meta, hist, kmin, score, C_M, gap_open1, k - 1, 1, BT_I1_OPEN);
      } else if (matrix == C_I2) {
        if (lane == 0) cand = bt_fetch(kp, base_meta, hist, kmin, score, C_I2, gap_extend2, k - 1, 1, BT_I2_EXT);
        if (lane == 1) cand = bt_fetch(kp, base_meta, hist, kmin, score, C_M, gap_open2, k - 1, 1, BT_I2_OPEN);
      } else if (matrix == C_D1) {
        if (lane == 0) cand = bt_fetch(kp, base_meta, hist, kmin, score, C_D1, gap_extend1, k + 1, 0, BT_D1_EXT);
        if (lane == 1) cand = bt_fetch(kp, base_meta, hist, kmin, score, C_M, gap_open1, k + 1, 0, BT_D1_OPEN);
      } else {
        if (lane == 0) cand = bt_fetch(kp, base_meta, hist, kmin, score, C_D2, gap_extend2, k + 1, 0, BT_D2_EXT);
        if (lane == 1) cand = bt_fetch(kp, base_meta, hist, kmin, score, C_M, gap_open2, k + 1, 0, BT_D2_OPEN);
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) cand = max(cand, __shfl_xor(cand, o));
      const int max_all = __shfl(cand, 0);
      if (max_all < 0) { err = ST_INTERNAL; break; }
      if (matrix == C_M) {
        const int max_offset = max_all >> 4;
        const int num_matches = offset - max_offset;
        if (num_matches < 0) { err = ST_INTERNAL; break; }
        push(0, num_matches);
        offset = max_offset;
        v = offset - k;
        h = offset;
        if (v <= 0 || h <= 0) break;
      }
      const int bt = max_all & 0xF;
      switch (bt) {
        case BT_M: sc = mismatch; matrix = C_M; break;
        case BT_I1_OPEN: sc = gap_open1; matrix = C_M; break;
        case BT_I1_EXT: sc = gap_extend1; matrix = C_I1; break;
        case BT_I2_OPEN: sc = gap_open2; matrix = C_M; break;
        case BT_I2_EXT: sc = gap_extend2; matrix = C_I2; break;
        case BT_D1_OPEN: sc = gap_open1; matrix = C_M; break;
        case BT_D1_EXT: sc = gap_extend1; matrix = C_D1; break;
        case BT_D2_OPEN: sc = gap_open2; matrix = C_M; break;
        default: sc = gap_extend2; matrix = C_D2; break;
      }
      if (bt == BT_M) { push(1, 1); --offset; }
      else if (bt <= BT_I2_EXT) { push(2, 1); --k; --offset; }
      else { push(3, 1); ++k; }
      v = offset - k;
      h = offset;
    }
    if (!err) {
      if (matrix == C_M) {
        if (v > 0 && h > 0) {
          const int nm = min(v, h);
          push(0, nm);
          v -= nm;
          h -= nm;
        }
        push(3, v);
        push(2, h);
      } else if (v != 0 || h != 0 || sc != 0) {
        err = ST_INTERNAL;
      }
    }
    if (last_op >= 0) { if (lane == 0) events[nev] = ((uint32_t)last_cnt << 2) | (uint32_t)last_op; ++nev; }
    if (lane == 0) { sh.nev = nev; sh.bt_total = total; if (err) sh.error = err; }
  }
  __syncthreads();
  if (sh.error) return sh.error;
  // ---- emission: events were produced end -> start
  const int nev = sh.nev;
  const uint8_t opc[4] = {'M', 'X', 'I', 'D'};
  for (int e = nev - 1; e >= 0; --e) {
    const uint32_t ev = events[e];
    emit_run(em, opc[ev & 3u], (int)(ev >> 2));
  }
  __syncthreads();  // events / sh.nev are reused by the next base case
  return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// BiWFA breakpoint search (A.6)
// ---------------------------------------------------------------------------------------------
constexpr int BP_OK = 0, BP_END_REACHED = 100;

template <bool P2>
__device__ void bialign_overlap(const KParams& kp, Shared& sh, const SubCtx& cx, int32_t* ring_mem, int d0, int s0,
                                int s1, bool fwd, Breakpoint& bp, unsigned long long* lstats) {
  const DevPenalties& pn = kp.pen;
  const int d1 = 1 - d0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rmask = kp.ring - 1;
  const int plen = cx.plen, tlen = cx.tlen, L = plen + tlen, D = tlen - plen;
  const int slot0 = s0 & rmask;
  const int A0 = sh.bi_A[d0][slot0], oob0 = sh.bi_oob[d0][slot0];
  const int kmin0 = cx.kmin[d0], kmin1 = cx.kmin[d1];
  // exact pre-filter: an overlap needs antidiag0 + antidiag1 >= plen + tlen on some diagonal;
  // every in-bounds cell of any component at score s is <= the (extended) M cell, so the rows'
  // max M antidiagonals bound it unless an out-of-bounds value was seen (oob).
  auto group_pass = [&](int si) {
    const int slot1 = si & rmask;
    return oob0 || sh.bi_oob[d1][slot1] || (A0 + sh.bi_A[d1][slot1] >= L);
  };
  bool any = false;
  for (int i = 0; i < pn.scope; ++i) {
    const int si = s1 - i;
    if (si < 0) break;
    if (!group_pass(si)) continue;
    if (s0 + si - (P2 ? max(pn.o1, pn.o2) : pn.o1) < bp.score) { any = true; break; }
  }
  if (!any) return;
  for (int i = tid; i < pn.scope * NCOMP; i += WG) sh.firstk[i] = INT_MAX;
  __syncthreads();
  // stage 1: parallel scan of every candidate wavefront pair (superset of what the sequential
  // search visits: the best score only decreases within a call)
  auto scan = [&](int c, int i, int si) {
    const RowMeta r0 = sh.bi_meta[d0][c][slot0], r1 = sh.bi_meta[d1][c][si & rmask];
    if (row_empty(r0) || row_empty(r1)) return;
    const int a = max(r0.lo, D - r1.hi), b = min(r0.hi, D - r1.lo);
    if (a > b) return;
    if (tid == 0) lstats[STAT_OVERLAP] += 1;
    const int32_t* p0 = row_ptr<false>(kp, ring_mem, d0, c, s0);
    const int32_t* p1 = row_ptr<false>(kp, ring_mem, d1, c, si);
    const int ca = a - kmin0, cbn = b - kmin0;
    for (int cbase = (ca & ~63) + 64 * wave; cbase <= cbn; cbase += WG) {
      const int col0 = cbase + lane;
      const bool act = col0 >= ca && col0 <= cbn;
      const int k0 = col0 + kmin0, k1 = D - k0;
      const int32_t h0 = act ? p0[col0] : OFF_NULL;
      const int32_t h1 = act ? p1[k1 - kmin1] : OFF_NULL;
      bool cond = act && (h0 + h1 >= tlen);
      if (c != C_M) {  // indel2indel skips out-of-bounds forward coordinates
        const int kf = fwd ? k0 : k1, hf = fwd ? h0 : h1;
        cond = cond && !((hf - kf) > plen || hf > tlen);
      }
      const uint64_t mask = __ballot(cond);
      if (mask) {
        if (lane == 0) atomicMin(&sh.firstk[i * NCOMP + c], cbase + (int)__builtin_ctzll(mask) + kmin0);
        break;
      }
    }
  };
  for (int i = 0; i < pn.scope; ++i) {
    const int si = s1 - i;
    if (si < 0) break;
    if (!group_pass(si)) continue;
    if (P2 && s0 + si - pn.o2 < bp.score) { scan(C_D2, i, si); scan(C_I2, i, si); }
    if (s0 + si - pn.o1 < bp.score) { scan(C_D1, i, si); scan(C_I1, i, si); }
    if (s0 + si < bp.score) scan(C_M, i, si);
  }
  __syncthreads();
  // stage 2: replay in WFA2's order (per i: D2, I2, D1, I1, M; first k ascending)
  auto apply = [&](int c, int i, int si, int gap_open) {
    const int k0 = sh.firstk[i * NCOMP + c];
    if (k0 == INT_MAX) return;
    if (s0 + si - gap_open >= bp.score) return;
    const int k1 = D - k0;
    const int32_t h0 = row_ptr<false>(kp, ring_mem, d0, c, s0)[k0 - kmin0];
    const int32_t h1 = row_ptr<false>(kp, ring_mem, d1, c, si)[k1 - kmin1];
    if (fwd) { bp.sf = s0; bp.sr = si; bp.kf = k0; bp.kr = k1; bp.off_f = h0; bp.off_r = h1; }
    else { bp.sf = si; bp.sr = s0; bp.kf = k1; bp.kr = k0; bp.off_f = h1; bp.off_r = h0; }
    bp.score = s0 + si - gap_open;
    bp.comp = c;
  };
  for (int i = 0; i < pn.scope; ++i) {
    const int si = s1 - i;
    if (si < 0) break;
    if (!group_pass(si)) continue;
    if (P2 && s0 + si - pn.o2 < bp.score) { apply(C_D2, i, si, pn.o2); apply(C_I2, i, si, pn.o2); }
    if (s0 + si - pn.o1 < bp.score) { apply(C_D1, i, si, pn.o1); apply(C_I1, i, si, pn.o1); }
    if (s0 + si >= bp.score) continue;
    apply(C_M, i, si, 0);
  }
  __syncthreads();  // firstk is rewritten by the next call
}

template <bool P2>
__device__ int find_breakpoint(const KParams& kp, Shared& sh, SubCtx cx, int32_t* ring_mem, int cb, int ce,
                               int score_remaining, Breakpoint& bp, unsigned long long* lstats) {
  const DevPenalties& pn = kp.pen;
  const int tid = threadIdx.x;
  const int plen = cx.plen, tlen = cx.tlen;
  const int rmask = kp.ring - 1;
  // column spaces: forward kmin, reverse kmin mirrored on chunk boundaries (C == 63 mod 64)
  {
    long long bound = (long long)score_remaining + 2LL * pn.scope + 16;
    const int blo = (int)min((long long)plen, bound), bhi = (int)min((long long)tlen, bound);
    const int need = -blo - 4;
    cx.kmin[0] = need;
    const int D = tlen - plen;
    const int c0 = D - need - need;
    const int adj = ((63 - c0) % 64 + 64) % 64;
    cx.kmin[1] = need - adj;
    cx.wcols = blo + bhi + 9 + adj;
    if (cx.wcols > kp.wcap) return ST_CAPACITY;
  }
  // score-0 wavefronts (wavefront_unialign_init by begin component)
  for (int i = tid; i < 2 * NCOMP; i += WG) {
    const int dir = i / NCOMP, c = i % NCOMP;
    const int begin = dir == 0 ? cb : ce;
    sh.bi_meta[dir][c][0] = (c == begin) ? RowMeta{0, 0} : RowMeta{1, 0};
  }
  if (tid == 0 || tid == 64) {
    const int dir = tid >> 6;
    const int begin = dir == 0 ? cb : ce;
    unsigned it = 0;
    int v0 = 0;
    if (begin == C_M) v0 = extend_lcp(cx.P[dir], cx.T[dir], 0, 0, plen, tlen, it);
    row_ptr<false>(kp, ring_mem, dir, begin, 0)[0 - cx.kmin[dir]] = v0;
    sh.ext0[dir] = v0;
    sh.bi_A[dir][0] = (begin == C_M) ? 2 * v0 : 0;
    sh.bi_oob[dir][0] = 0;
    acc_reset(sh.acc[0][dir]);
    acc_reset(sh.acc[1][dir]);
    acc_reset(sh.acc[2][dir]);
  }
  __syncthreads();
  if (cb == C_M && ce == C_M && plen == tlen && (sh.ext0[0] >= tlen || sh.ext0[1] >= tlen)) return BP_END_REACHED;
  const int max_antidiagonal = plen + tlen - 1;
  int sf = 0, sr = 0;      // official scores
  int cf = 0, cr = 0;      // computed scores (may run one ahead: speculative row)
  int fmax = sh.bi_A[0][0], rmax = sh.bi_A[1][0];
  bp.score = INT_MAX;
  unsigned ext_iters = 0;
  unsigned long long cells = 0;
  int pass = 0;
  const long long max_steps = ((long long)pn.o1 + pn.o2 + 2LL * (pn.e1 + pn.e2) + pn.x) * ((long long)plen + tlen + 4) + 1024;
  long long steps = 0;
  int rc = BP_OK;
  // one fused pass: next forward row and next reverse row (whichever is missing)
  auto ensure_computed = [&]() {
    const bool needF = cf == sf, needR = cr == sr;
    if (!needF && !needR) return;
    Acc* a = sh.acc[pass % 3];
    if (needF) cells += compute_row<P2, false>(kp, sh, nullptr, cx, ring_mem, 0, sf + 1, a[0], ext_iters);
    if (needR) cells += compute_row<P2, false>(kp, sh, nullptr, cx, ring_mem, 1, sr + 1, a[1], ext_iters);
    __syncthreads();
    if (needF) { finalize_row<false>(kp, sh, nullptr, cx, 0, sf + 1, a[0]); cf = sf + 1; }
    if (needR) { finalize_row<false>(kp, sh, nullptr, cx, 1, sr + 1, a[1]); cr = sr + 1; }
    if (tid == 0) { acc_reset(sh.acc[(pass + 2) % 3][0]); acc_reset(sh.acc[(pass + 2) % 3][1]); }
    ++pass;
  };
  bool last_fwd = false;
  // phase 1: until the furthest points can collide
  for (;;) {
    if (fmax + rmax >= max_antidiagonal) break;
    ensure_computed();
    if (sh.error) { rc = sh.error; break; }
    ++sf;
    fmax = max(fmax, sh.bi_A[0][sf & rmask]);
    last_fwd = true;
    if (fmax + rmax >= max_antidiagonal) break;
    ++sr;
    rmax = max(rmax, sh.bi_A[1][sr & rmask]);
    last_fwd = false;
    if (++steps > max_steps) { rc = ST_MAX_STEPS; break; }
  }
  // phase 2: until no better overlap is possible
  const int gap_opening = P2 ? max(pn.o1, pn.o2) : pn.o1;
  while (rc == BP_OK) {
    ensure_computed();
    if (sh.error) { rc = sh.error; break; }
    if (last_fwd) {
      const int min_sr = (sr > pn.scope - 1) ? sr - (pn.scope - 1) : 0;
      if (sf + min_sr - gap_opening >= bp.score) break;
      bialign_overlap<P2>(kp, sh, cx, ring_mem, 0, sf, sr, true, bp, lstats);
      ++sr;
    }
    const int min_sf = (sf > pn.scope - 1) ? sf - (pn.scope - 1) : 0;
    if (min_sf + sr - gap_opening >= bp.score) break;
    bialign_overlap<P2>(kp, sh, cx, ring_mem, 1, sr, sf, false, bp, lstats);
    ++sf;
    last_fwd = true;
    if (++steps > max_steps) { rc = ST_MAX_STEPS; break; }
  }
  if (tid == 0) {
    lstats[STAT_CELLS] += cells;
    lstats[STAT_BREAKPOINTS] += 1;
  }
  atomicAdd(&lstats[STAT_EXTEND], (unsigned long long)ext_iters);
  __syncthreads();  // LDS metadata is rewritten by the next sub-problem
  if (rc == BP_OK && bp.score == INT_MAX) rc = ST_INTERNAL;
  return rc;
}

// ---------------------------------------------------------------------------------------------
// The kernel: persistent workgroups, one pair at a time, DFS over the BiWFA recursion
// ---------------------------------------------------------------------------------------------
template <bool P2>
__global__ __launch_bounds__(WG) void biwfa_align_kernel(KParams kp) {
  __shared__ Shared sh;
  __shared__ unsigned long long lstats[STAT_N];
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  RowMeta* base_meta = reinterpret_cast<RowMeta*>(dyn_smem);
  const int tid = threadIdx.x;
  const DevPenalties& pn = kp.pen;
  int32_t* ring_mem = kp.ring_mem + (size_t)blockIdx.x * kp.ring_slot_stride;
  int32_t* hist = kp.hist_mem + (size_t)blockIdx.x * kp.hist_slot_stride;
  uint32_t* events = kp.ev_mem + (size_t)blockIdx.x * kp.ev_slot_stride;
  if (tid < STAT_N) lstats[tid] = 0;
  __syncthreads();
  for (;;) {
    if (tid == 0) {
      sh.cur_pair = (long long)atomicAdd(kp.work_counter, 1ULL);
      sh.error = 0;
    }
    __syncthreads();
    const long long pair = sh.cur_pair;
    if (pair >= kp.npairs) break;
    const int qi = kp.pair_q[pair], ti = kp.pair_t[pair];
    const int qv = kp.pair_rc[pair] ? 2 : 0;
    const uint64_t qoff = kp.seq_off[qi], toff = kp.seq_off[ti];
    const int plenT = kp.seq_len[qi], tlenT = kp.seq_len[ti];
    const uint8_t* Pf = kp.seq[qv] + qoff;
    const uint8_t* Pr = kp.seq[qv + 1] + qoff;
    const uint8_t* Tf = kp.seq[0] + toff;
    const uint8_t* Tr = kp.seq[1] + toff;
    Emit em;
    em.cig = kp.cigar + kp.cigar_off[pair];
    em.n = 0;
    em.cnt[0] = em.cnt[1] = em.cnt[2] = em.cnt[3] = 0;
    int status = ST_OK;
    int penalty = -1;
    int sp = 0;
    if (tid == 0) {
      const bool min_length = max(plenT, tlenT) <= FALLBACK_MIN_LENGTH;
      sh.stack[0] = Task{0, plenT, 0, tlenT, C_M, C_M, min_length ? 0 : INT_MAX};
    }
    sp = 1;
    __syncthreads();
    bool top = true;
    while (sp > 0 && status == ST_OK) {
      const Task t = sh.stack[sp - 1];
      --sp;
      __syncthreads();  // everyone has read the entry before it can be overwritten
      const int plen = t.pe - t.pb, tlen = t.te - t.tb;
      if (tlen == 0) {
        emit_run(em, 'D', plen);
        if (top) penalty = plen > 0 ? min(pn.o1 + plen * pn.e1, P2 ? pn.o2 + plen * pn.e2 : INT_MAX) : 0;
        top = false;
        continue;
      } else if (plen == 0) {
        emit_run(em, 'I', tlen);
        if (top) penalty = min(pn.o1 + tlen * pn.e1, P2 ? pn.o2 + tlen * pn.e2 : INT_MAX);
        top = false;
        continue;
      }
      SubCtx cx;
      cx.plen = plen;
      cx.tlen = tlen;
      cx.P[0] = Pf + t.pb;
      cx.T[0] = Tf + t.tb;
      cx.P[1] = Pr + (plenT - t.pe);
      cx.T[1] = Tr + (tlenT - t.te);
      cx.kmin[0] = cx.kmin[1] = 0;
      cx.wcols = 0;
      bool do_base = t.score_remaining <= FALLBACK_MIN_SCORE;
      Breakpoint bp;
      if (!do_base) {
        const int rc = find_breakpoint<P2>(kp, sh, cx, ring_mem, t.cb, t.ce, t.score_remaining, bp, lstats);
        if (rc == BP_END_REACHED) do_base = true;  // wavefront_bialign_exception -> plain WFA
        else if (rc != BP_OK) { status = rc; break; }
      }
      if (do_base) {
        int pen_b = 0;
        const int rc = base_align<P2>(kp, sh, base_meta, cx, hist, events, t.cb, t.ce, em, pen_b, lstats);
        if (rc != ST_OK) { status = rc; break; }
        if (top) penalty = pen_b;
        top = false;
        continue;
      }
      const int bh = bp.off_f, bv = bp.off_f - bp.kf;
      if (bh < 0 || bh > tlen || bv < 0 || bv > plen || sp + 2 > STACK_CAP) { status = ST_INTERNAL; break; }
      if (tid == 0) {
        sh.stack[sp] = Task{t.pb + bv, t.pe, t.tb + bh, t.te, bp.comp, t.ce, bp.sr};      // right half
        sh.stack[sp + 1] = Task{t.pb, t.pb + bv, t.tb, t.tb + bh, t.cb, bp.comp, bp.sf};  // left half first
      }
      sp += 2;
      if (top) penalty = bp.score;
      top = false;
      __syncthreads();
    }
    if (tid == 0) {
      DevResult r;
      r.status = status;
      r.penalty = status == ST_OK ? penalty : 0;
      r.score = -r.penalty;
      r.cigar_len = status == ST_OK ? (uint32_t)em.n : 0u;
      r.cigar_off = kp.cigar_off[pair];
      r.num_matches = em.cnt[0];
      r.num_mismatches = em.cnt[1];
      r.num_ins = em.cnt[2];
      r.num_del = em.cnt[3];
      r.q_end = em.cnt[0] + em.cnt[1] + em.cnt[3];
      r.t_end = em.cnt[0] + em.cnt[1] + em.cnt[2];
      kp.results[pair] = r;
      if (status == ST_OK) {
        lstats[STAT_ALIGNED_BP] += (unsigned long long)plenT;
        lstats[STAT_PAIRS] += 1;
      }
    }
    __syncthreads();
  }
  __syncthreads();
  if (tid < STAT_N && lstats[tid]) atomicAdd(&kp.stats[tid], lstats[tid]);
}

}  // namespace awv
