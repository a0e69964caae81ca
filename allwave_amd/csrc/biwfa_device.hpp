// biwfa_device.hpp -- hand-written HIP (gfx950 / CDNA4) kernels for allwave's per-pair hot path:
// end-to-end BiWFA under gap-affine / 2-piece gap-affine penalties with full CIGAR.
//
// Replaces what the reference does inside `wf.align(query, target)`
// (/root/reference/src/alignment.rs:231, src/wfa.rs:226) -> WFA2-lib [not in the container;
// semantics per SURVEY.md Appendix A].  Results are bit-exact against oracle/biwfa_oracle.c.
//
// Mapping onto the machine (integer DP: no MFMA) -- DESIGN.md section 4:
//   * one sequence pair per workgroup, persistent workgroups pull pairs from an atomic cursor; the
//     BiWFA recursion is an explicit DFS stack, so CIGAR ops come out in order and no device
//     recursion is needed;
//   * lanes <-> diagonals: a lane owns 4 consecutive diagonals, a wave a 256-column window per
//     iteration; forward and reverse column spaces are mirrored on 256-column chunk boundaries so the
//     meet-in-the-middle overlap test reads both wavefronts with aligned vector loads;
//   * wavefront rows live in a per-workgroup arena in HBM (ring of `ring` rows per component and
//     direction), row metadata (lo/hi, max antidiagonal) in LDS;
//   * while the two searches are far apart a window advances 5 scores per pass (15 with chained sweeps
//     under the default scores) with the I/D rows -- and the previous sweeps' M rows -- in registers
//     (compute_rows_multi; the chain registers double as the first sweep's load targets, the next sweep's
//     remaining source rows are loaded a sweep ahead); from the safety margin before the meeting point on,
//     passes of 5 scores that also store every I/D row (deep_phase), until phase 1 of the search ends; phase 2
//     and trimmed rows go score by score (compute_row), forward and reverse steps fused, one barrier per step;
//   * long sequences (32-bit rows): three chained sweeps with the middle sweep's rows in LDS wherever the
//     sub-problem is too long for its packed sequences to be staged there; such a sub-problem probes the 2-bit
//     words where they lie in HBM (seq_mode 2), and every sub-problem that has become short enough is searched
//     with 16-bit rows inside the same launch (AWV_SUB16);
//   * the far-apart phase (multi_phase, which tail-calls deep_phase), the base case's passes (base_phase), the
//     breakpoint search (find_breakpoint_fn) and the trimmed-hull search (trim_pass_fn) are real functions with
//     register files of their own; their inputs travel through LDS (Shared::pctx); the passes' planning
//     (plan_multi) runs on the scalar unit without an LDS round trip per score.
//
//   * rows are 16-bit whenever every stored value fits: text offsets when both sequences are shorter than
//     32760, min(h, v) per cell when only the shorter one is (AWV_WIDE16: 32-bit row metadata); 32-bit rows
//     otherwise, in 5-score sweeps whose M sources are loaded one step ahead, two sweeps chained.
//
// The header is compiled once per workgroup size: AWV_WG = 64 (one wave per pair: the throughput
// kernel), AWV_WG = 256 (four waves share a pair's rows: small batches and very expensive pairs) and
// AWV_WG = 1024 (sixteen waves: a few enormous pairs), the latter two also with AWV_WIDE16, each into its
// own namespace AWV_NS.


#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>

#ifndef AWV_NS
#define AWV_NS awv
#endif
#ifndef AWV_WG
#define AWV_WG 64
#endif
namespace AWV_NS {

constexpr int WG = AWV_WG;  // threads per workgroup = per sequence pair (64 or 256)
// Direction split (two waves per pair): in the breakpoint search wave 0 computes the forward rows and
// wave 1 the reverse rows of a fused pass, each over all of its row's windows.
#ifdef AWV_DIRSPLIT
constexpr bool DIRSPLIT = true;
static_assert(AWV_WG == 128, "direction split = two waves per pair");
#else
constexpr bool DIRSPLIT = false;
#endif
constexpr int MAX_RING = 128;
constexpr int NCOMP = 5;
constexpr int32_t OFF_NULL = INT32_MIN / 2;
constexpr int32_t NULLISH = INT32_MIN / 4;    // any value below is a NULL(+n)
constexpr int32_t NULL16 = -16384;            // NULL as stored in 16-bit rows (NULL16 + NULL16 and NULL16 + tlen + 1 stay negative)
// AWV_WIDE16: 16-bit rows for pairs of which only the SHORTER sequence fits 16 bits (forced-gap pairs: 2 kbp against
// 45 kbp).  A cell on diagonal k then stores w = h - max(k, 0) = min(h, v) <= min(plen, tlen) instead of the text
// offset h, and the row metadata (diagonals up to the longer length) stays 32-bit.  Within a diagonal w orders like
// h, so every max() of the recurrences is unchanged; across diagonals an insertion adds 1 only where k <= 0 (source
// diagonal k - 1 < 0) and a deletion only where k >= 0; a cell is inside the matrix <=> 0 <= w <= min(plen + k,
// tlen - k, plen, tlen).  Everything outside the packed arithmetic sees h again (off_load1 / decode4 take k).
#ifdef AWV_WIDE16
constexpr bool WENC = true;
#else
constexpr bool WENC = false;
#endif
enum { C_M = 0, C_I1 = 1, C_I2 = 2, C_D1 = 3, C_D2 = 4 };
constexpr int FALLBACK_MIN_SCORE = 250;
constexpr int FALLBACK_MIN_LENGTH = 100;
constexpr int STACK_CAP = 48;
// 1: inside a launch with 32-bit rows, sub-problems of which both lengths are below SUB16_MAX_LEN are searched with 16-bit rows
#ifndef AWV_SUB16
#define AWV_SUB16 1
#endif
constexpr int SUB16_MAX_LEN = 32760;  // (the engine's own limit for 16-bit rows: engine.hip)
// Occupancy target: waves per SIMD (the register budget the kernel is compiled for) and the static
// LDS the kernel declares; the engine sizes the dynamic LDS so that WAVES_PER_SIMD * 4 waves fit a CU.
#ifndef AWV_WAVES_PER_SIMD
#define AWV_WAVES_PER_SIMD 4
#endif
constexpr int WAVES_PER_SIMD = AWV_WAVES_PER_SIMD;
constexpr int STATIC_LDS_RESERVE = 1024;
constexpr int COL_PAD = 576;  // columns of slack either side of a row: whole-wave vector loads stay inside it

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

enum { STAT_CELLS = 0, STAT_EXTEND, STAT_BREAKPOINTS, STAT_BASE, STAT_OVERLAP, STAT_ALIGNED_BP, STAT_PAIRS,
       // cycle stamps (s_memtime, wave 0) -- only filled by the -DAWV_PROF diagnostic build
       STAT_T_TOTAL, STAT_T_BI_COMPUTE, STAT_T_BI_BARRIER, STAT_T_BI_FINALIZE, STAT_T_OVERLAP, STAT_T_BASE_STEPS,
       STAT_T_BACKTRACE, STAT_T_EMIT, STAT_N_PASSES,
       STAT_T_CR_LOAD, STAT_T_CR_ALU, STAT_T_CR_EXTEND, STAT_T_CR_STORE, STAT_T_CR_REDUCE, STAT_RESTARTS, STAT_MULTI_CELLS, STAT_WIN_SINGLE, STAT_WIN_MULTI, STAT_WIN_BASE, STAT_WIN_BASE_MULTI,
       // the shader clock the launch actually ran at: every persistent workgroup stamps s_memtime (shader cycles) and
       // s_memrealtime (constant 100 MHz) once when it starts and once when it has drained the queue; the sums give
       // cycles / ticks x 100 MHz (MI355X_MICROARCH.md: the clock under load is not the nominal 2.4 GHz)
       STAT_CLK_CYCLES, STAT_CLK_TICKS,
       STAT_DEEP_CELLS,  // cells computed by deep_phase (passes that also store every I/D row); part of STAT_MULTI_CELLS
       STAT_N };

#ifdef AWV_PROF
#define PROF_DRAIN() __builtin_amdgcn_s_waitcnt(0)
#define PROF_NOW() __builtin_readcyclecounter()
#define PROF_ADD(slot, t0) do { if (threadIdx.x == 0) lstats[slot] += __builtin_readcyclecounter() - (t0); } while (0)
#define PROF_INC(slot) do { if (threadIdx.x == 0) lstats[slot] += 1; } while (0)
#define PROF_ADD_L(slot, t0) do { if (threadIdx.x == 0) sh.prof[(slot) - STAT_T_CR_LOAD] += __builtin_readcyclecounter() - (t0); } while (0)
#else
#define PROF_ADD_L(slot, t0) do { (void)(t0); } while (0)
#define PROF_DRAIN() do { } while (0)
#define PROF_NOW() 0ULL
#define PROF_ADD(slot, t0) do { (void)(t0); } while (0)
#define PROF_INC(slot) do { } while (0)
#endif

struct KParams {
  const uint8_t* seq[4];  // 0 fwd, 1 reversed, 2 reverse-complement, 3 reversed reverse-complement
  const uint64_t* seq_off;
  const int32_t* seq_len;
  const uint32_t* seq2[2];     // 2-bit packed forward / reverse-complement (16 bases per word, base i in bits 2(i%16))
  const uint64_t* seq2_off;    // word offset of each sequence in seq2[*]
  const uint8_t* seq2_ok;      // bit0: forward bytes are all upper-case ACGT, bit1: the reverse complement is
  const int32_t* pair_q;
  const int32_t* pair_t;
  const int32_t* pair_rc;
  long long npairs;
  DevPenalties pen;
  int ring;        // power of two >= scope + 2 (+ multi_T - 1 with multi-step passes)
  int chain_max;   // sweeps a multi-step pass may chain (1: none; > 1 needs x == TMAX and o1 + e1 == 2 TMAX, ring >= scope + 2 + TMAX * chain_max - 1)
  int multi_T;     // steps per multi-step pass (0: off): <= min(TMAX, x, o1+e1, o2+e2, ring - scope - 1), e1/e2 among the instantiated depths
  int deep_passes; // 1: the margin zone of phase 1 runs in passes that store every I/D row (deep_phase); 0: step by step there (round 2)
  int sub16;       // 32-bit launches: 1 = sub-problems whose lengths fit 16-bit rows are searched with 16-bit rows (AWV_SUB16)
  int wcap;        // columns per ring row
  void* ring_mem;
  size_t ring_slot_stride;  // bytes per workgroup slot
  int lds_meta_bytes;  // dynamic LDS: metadata region (16-byte multiple)
  int lds_seq_bytes;   // dynamic LDS: sequence staging region (0 = sequences stay in global memory)
  int sb_cap;      // base-case score capacity
  int wb_cap;      // base-case columns per row
  void* hist_mem;
  size_t hist_slot_stride;  // bytes per workgroup slot
  size_t hist_meta_offset;  // byte offset of the base-case metadata log inside a hist slot
  uint32_t* ev_mem;
  size_t ev_slot_stride;
  uint8_t* cigar;
  const uint64_t* cigar_off;
  DevResult* results;
  unsigned long long* work_counter;
  unsigned long long* stats;
};

struct RowMeta { int lo, hi; };
constexpr int K_BIG = 1 << 28;  // an empty row is {K_BIG, -K_BIG}: min/max hulls ignore it for free
#define ROW_EMPTY RowMeta{K_BIG, -K_BIG}
struct Acc { int hull_lo[NCOMP]; int hull_hi[NCOMP]; int maxak; int oob; int reach; };  // maxak_t: per step of a multi-step pass
struct Task { int pb, pe, tb, te, cb, ce, score_remaining, known; };  // known: the sub-problem's optimal score (INT_MAX at the top)
struct Breakpoint { int score, sf, sr, kf, kr, off_f, off_r, comp; };

// Scalar-unit arithmetic, spelled out.  Uniform min / max / add chains whose results end up in vector registers anyway (the data
// of an LDS store, a per-lane select) are otherwise selected as VALU code wholesale: plan_step's hull arithmetic was ~90 vector
// instructions per score and direction that way (v_min3 / v_max3 / v_cndmask on values that had just been read into scalar
// registers), 5 % of everything the kernel issues.  Operands come from v_readlane / scalar loads / constants.
__device__ __forceinline__ int s_min(int a, int b) { int r; asm("s_min_i32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc"); return r; }
__device__ __forceinline__ int s_max(int a, int b) { int r; asm("s_max_i32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc"); return r; }
__device__ __forceinline__ int s_add(int a, int b) { int r; asm("s_add_i32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc"); return r; }

struct alignas(4) RowMeta16 { int16_t lo, hi; };  // (one 32-bit LDS access, also where the compiler cannot see the base's alignment)  // |k| < 32760 whenever 16-bit rows are in use; empty = {1, 0}
template <typename OffT> struct MetaTraits;
#ifdef AWV_WIDE16
template <> struct MetaTraits<int16_t> { typedef RowMeta Stored; };
#else
template <> struct MetaTraits<int16_t> { typedef RowMeta16 Stored; };
#endif
template <> struct MetaTraits<int32_t> { typedef RowMeta Stored; };
__device__ __forceinline__ RowMeta meta_load(const RowMeta16* p) {
  const RowMeta16 m = *p;
  return m.lo > m.hi ? ROW_EMPTY : RowMeta{m.lo, m.hi};
}
__device__ __forceinline__ RowMeta meta_load(const RowMeta* p) { return *p; }
__device__ __forceinline__ void meta_store(RowMeta16* p, const RowMeta& m) {
  *p = m.lo > m.hi ? RowMeta16{1, 0} : RowMeta16{(int16_t)m.lo, (int16_t)m.hi};
}
// the same for a row whose lo / hi are in scalar registers (plan_step): the stored word(s) are put together on the scalar unit --
// lo > hi: the empty row -- and one vector move + LDS store per word remain
__device__ __forceinline__ void meta_store_scalar(RowMeta16* p, int lo, int hi) {
  int w;
  asm("s_pack_ll_b32_b16 %0, %1, %2\n\ts_cmp_gt_i32 %1, %2\n\ts_cselect_b32 %0, 1, %0" : "=&s"(w) : "s"(lo), "s"(hi) : "scc");
  *reinterpret_cast<int*>(p) = w;
}
__device__ __forceinline__ void meta_store_scalar(RowMeta* p, int lo, int hi) {
  int l, h;
  asm("s_cmp_gt_i32 %2, %3\n\ts_cselect_b32 %0, %4, %2\n\ts_cselect_b32 %1, %5, %3" : "=&s"(l), "=&s"(h) : "s"(lo), "s"(hi), "s"(K_BIG), "s"(-K_BIG) : "scc");
  *p = RowMeta{l, h};
}
__device__ __forceinline__ void meta_store(RowMeta* p, const RowMeta& m) { *p = m; }

// LDS: a small static part plus one dynamic region carved per launch:
//   [ ring_meta[2][NCOMP][ring] | bi_A[2][ring] | bi_oob[2][ring] | firstk[scope*NCOMP] | seq ]
// The base case (one direction) reuses ring_meta[0]; its full per-score metadata history, which
// only the backtrace reads, is logged to the workgroup's HBM arena instead of LDS.
template <typename OffT>
struct Lds {
  typename MetaTraits<OffT>::Stored* ring_meta;
  int* bi_A;
  int* bi_oob;
  int* firstk;
  uint32_t* seq;   // sequence staging: the sub-problem's 2-bit packed words (16 bases per word)
  RowMeta* meta_log;  // HBM: [score][NCOMP] of the running base case
};
// What a multi-step pass (multi_pass: a real function with registers of its own) needs to know about
// the kernel's parameters and the running sub-problem: written to LDS by find_breakpoint, read back
// into scalar registers inside the pass.
struct PassCtx {
  unsigned long long ring_mem;   // this workgroup's ring arena
  unsigned long long ring_bytes;
  unsigned long long P[2], T[2]; // SubCtx::P / T (raw-byte probes)
  int ring, wcap;
  int x, o1, e1, o2, e2;
  int lds_meta_bytes;
  int chain_max;
  int plen, tlen, kmin[2], wcols;
  int seq_mode, p_w0, t_w0, p_bit, t_bit;
  // base case (base_phase): history arena instead of the ring, the metadata log, capacities, the end cell
  unsigned long long meta_log;
  int wb_cap, sb_cap, end_comp;
  // the breakpoint search as a function (find_breakpoint_fn): launch constants and the staged-sequence bookkeeping of the sub-problem
  unsigned long long Pw, Tw;
  int multi_T, lds_seq_bytes, pb_abs, tb_abs;
  int deep_passes;
};
struct PhaseResult {
  int why, sc, fmax, rmax, npass;
  int sf, sr, last_fwd;  // the official scores and the side that advanced last where the phase stopped (sf = sr = sc unless deep_phase ended phase 1)
  int deep_from;         // every I/D row above this score is in memory (deep_phase began here; = sc when it did not run)
  unsigned long long cells, deep_cells;  // cells of all the passes / of deep_phase's
};
struct Shared {
  Acc acc[3][2];
  int chain_maxak[2][8];  // per direction: [0] the max antidiagonal of a far-apart pass; [t] of step t of a deep pass (TMAX <= 8)
  PassCtx pctx;
  PhaseResult pres;
  Breakpoint bp_out;  // find_breakpoint_fn's result
  unsigned long long ext_multi;  // extend probes counted by multi-step passes
  unsigned int win_single, win_multi, win_base, win_base_multi;  // windows processed (diagnostics)
  int ext0[2];
#ifdef AWV_PROF
  unsigned long long prof[5];  // (cycle stamps inside the step code: the -DAWV_PROF diagnostic build only; the static LDS reserve has no room to spare)
#endif
  long long cur_pair;
  int error;
  int nev;
  int bt_total;
};

// Pointers that arrive inside the KParams struct are generic (flat) to the compiler; the hot
// sequence reads go through explicit global-address-space pointers (global_load + SGPR base).
typedef const __attribute__((address_space(1))) uint8_t* gseq_t;
typedef uint64_t u64_unaligned __attribute__((aligned(1)));
typedef const __attribute__((address_space(1))) uint32_t* gwords_t;
__device__ __forceinline__ gseq_t to_global(const uint8_t* p) { return (gseq_t)(uintptr_t)p; }

struct SubCtx {
  int plen, tlen;
  gseq_t P[2];
  gseq_t T[2];
  int kmin[2];
  int wcols;
  // 2-bit packed staging in LDS (only when both sequences are pure upper-case ACGT and the words fit)
  int seq_mode;            // 0: probes read raw bytes from global memory, 1: packed words staged in LDS, 2: packed words read from global memory (P[0] / T[0] point at them)
  int p_w0, t_w0;          // first staged word of pattern / text in Lds::seq (2 pad words before, 3 after)
  int p_bit, t_bit;        // position (0..15) of the sub-problem's first base inside that word
  gwords_t Pw, Tw;         // the whole sequences' packed words in HBM (nullptr: not packable)
  int pb_abs, tb_abs;      // sub-problem start inside the sequences
};

__device__ __forceinline__ bool row_empty(const RowMeta& m) { return m.lo > m.hi; }

// Values read from LDS / computed from them are uniform across the workgroup by construction but
// arrive in vector registers; readfirstlane moves them to SGPRs so the step planning, row offsets
// and control flow run on the scalar unit (and stop eating the VGPR budget).
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
// clamp(x, -1, hi) for hi >= -1 in one instruction (the compiler cannot prove -1 <= hi for a runtime bound)
__device__ __forceinline__ int clamp_from_m1(int x, int hi) {
  int r;
  asm("v_med3_i32 %0, %1, -1, %2" : "=v"(r) : "v"(x), "s"(hi));
  return r;
}
// threadIdx.x for the cold phases: opaque, so that what is derived from it (strided loop counters,
// per-thread pointers) is recomputed there instead of being hoisted into registers that stay
// allocated across the hot row loop
__device__ __forceinline__ int cold_tid() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}
__device__ __forceinline__ RowMeta uni(const RowMeta& m) { return RowMeta{uni(m.lo), uni(m.hi)}; }

__device__ __forceinline__ uint64_t ld64u(gseq_t p) {
  return *(const __attribute__((address_space(1))) u64_unaligned*)p;
}

// wave64 max-reduction on the DPP network (no LDS traffic): row_shr 1/2/4/8 leaves each row's max
// in its lane 15, row_bcast15 / row_bcast31 fold the four rows; lane 63 holds the result.
__device__ __forceinline__ int wave_max_i32(int v) {
  v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x111, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x112, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x114, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x118, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x142, 0xa, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x143, 0xc, 0xf, false));
  return __builtin_amdgcn_readlane(v, 63);
}

// ---------------------------------------------------------------------------------------------
// Row storage: OffT = int16_t (used when every offset fits, i.e. lengths < 32760) or int32_t; a lane
// vector is 4 diagonals (8 or 16 bytes).  32-bit rows compute in int32 with WFA2's NULL =
// INT32_MIN/2 (A.1); 16-bit rows compute on packed halves.  16-bit rows store every NULL(+n) as
// NULL16 = -16384 and clamp offsets past the text end to tlen+1: both are invisible to every
// comparison the algorithm makes (a NULL(+n) only ever meets max() against real offsets or the
// out-of-bounds test; an h > tlen offset stays out of bounds under +1 / diagonal shifts and only
// ever wins max() or fails the bounds test).
// ---------------------------------------------------------------------------------------------
template <typename OffT> struct OffTraits;
template <> struct OffTraits<int32_t> { static constexpr int VEC = 4; };
template <> struct OffTraits<int16_t> { static constexpr int VEC = 4; };

template <typename OffT> constexpr bool wenc_of() { return WENC && sizeof(OffT) == 2; }
// one stored cell of diagonal k as a text offset (OFF_NULL for a NULL)
template <typename OffT>
__device__ __forceinline__ int32_t off_load1(const OffT* p, int k) {
  const int32_t v = (int32_t)*p;
  if (sizeof(OffT) == 2) return v < 0 ? OFF_NULL : (wenc_of<OffT>() ? v + max(k, 0) : v);
  return v;
}
// an unpacked lane vector whose element j lies on diagonal k0 + j: stored form -> text offsets (negative = NULL stays)
template <typename OffT>
__device__ __forceinline__ void decode4(int32_t (&a)[4], int k0) {
  if constexpr (wenc_of<OffT>()) {
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = a[j] < 0 ? a[j] : a[j] + max(k0 + j, 0);
  }
}
// the largest stored value inside the matrix on diagonal k (-1: the diagonal lies outside), `cap` = min(plen, tlen)
__device__ __forceinline__ int wenc_max(int k, int plen, int tlen, int cap) {
  int r;
  const int x = min(plen + k, tlen - k);
  asm("v_med3_i32 %0, %1, -1, %2" : "=v"(r) : "v"(x), "s"(cap));
  return r;
}

// Buffer addressing for the hot row accesses: one SGPR descriptor per arena, an SGPR byte offset
// per row, one shared VGPR column offset per lane and the k-1/k/k+1 shift as an immediate.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)min(bytes, (size_t)0x7FFFFFFF), 0x00020000);
}

template <typename OffT> struct RawVec;
template <> struct RawVec<int16_t> { unsigned int w[2]; };
template <> struct RawVec<int32_t> { unsigned int w[4]; };

// AUX = cache policy bits of the buffer instruction (0 default, 2 = nt: a line read for the last time)
template <typename OffT, int AUX = 0>
__device__ __forceinline__ RawVec<OffT> buf_load_raw(rsrc_t r, int voff, int soff) {
  RawVec<OffT> o;
  if constexpr (sizeof(OffT) == 2) {
    const auto raw = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX);
    o.w[0] = raw[0]; o.w[1] = raw[1];
  } else {
    const auto raw = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX);
    o.w[0] = raw[0]; o.w[1] = raw[1]; o.w[2] = raw[2]; o.w[3] = raw[3];
  }
  return o;
}

template <typename OffT>
__device__ __forceinline__ void unpack_raw(const RawVec<OffT>& r, int32_t (&out)[4]) {
  if constexpr (sizeof(OffT) == 2) {
    const int32_t a = (int32_t)r.w[0], b = (int32_t)r.w[1];
    out[0] = (a << 16) >> 16;
    out[1] = a >> 16;
    out[2] = (b << 16) >> 16;
    out[3] = b >> 16;  // a stored NULL is NULL16 (negative): every consumer treats any negative as NULL
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = (int32_t)r.w[j];
  }
}

// Canonical 16-bit stored form of two M cells (each either OFF_NULL or an offset >= 0): v < 0 ? NULL16 : min(v, tlen + 1),
// packed low | high.  One v_med3 per cell (OFF_NULL clamps up to NULL16, an offset past the text end down to tlen + 1)
// and one v_perm for the pair.
__device__ __forceinline__ unsigned pack_canon16(int32_t lo, int32_t hi, int tlen1, int& clo, int& chi) {
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(clo) : "v"(lo), "v"((int)NULL16), "s"(tlen1));
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(chi) : "v"(hi), "v"((int)NULL16), "s"(tlen1));
  return __builtin_amdgcn_perm((unsigned)chi, (unsigned)clo, 0x05040100u);
}
__device__ __forceinline__ unsigned pack_canon16(int32_t lo, int32_t hi, int tlen1) {
  int a, b;
  return pack_canon16(lo, hi, tlen1, a, b);
}

template <typename OffT>
__device__ __forceinline__ void buf_store_vec(rsrc_t r, int voff, int soff, const int32_t (&v)[4], int tlen) {
  if (sizeof(OffT) == 2) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    u32x2 raw;
    raw[0] = pack_canon16(v[0], v[1], tlen + 1);
    raw[1] = pack_canon16(v[2], v[3], tlen + 1);
    __builtin_amdgcn_raw_buffer_store_b64(raw, r, voff, soff, 0);
  } else {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 raw;
#pragma unroll
    for (int j = 0; j < 4; ++j) raw[j] = (unsigned)v[j];
    __builtin_amdgcn_raw_buffer_store_b128(raw, r, voff, soff, 0);
  }
}

template <bool BASE, typename OffT>
__device__ __forceinline__ int row_off(const KParams& kp, int dir, int comp, int score) {
  if (score < 0) score = 0;
  if (BASE) return (int)(((unsigned)score * NCOMP + comp) * (unsigned)kp.wb_cap * (unsigned)sizeof(OffT));
  return (int)((((unsigned)(dir * NCOMP + comp)) * kp.ring + (unsigned)(score & (kp.ring - 1))) * (unsigned)kp.wcap *
               (unsigned)sizeof(OffT));
}

// bounded LCP of pattern[v..] / text[h..] (A.4), 8 bytes per iteration
__device__ __forceinline__ int extend_lcp(gseq_t P, gseq_t T, int v, int h, int plen, int tlen,
                                          unsigned& iters) {
  const int rem = min(plen - v, tlen - h);
  gseq_t pp = P + (unsigned)v;
  gseq_t tp = T + (unsigned)h;
  int n = 0;
  while (n < rem) {
    const uint64_t x = ld64u(pp + (unsigned)n) ^ ld64u(tp + (unsigned)n);
    ++iters;
    if (x) {
      n += (int)(__builtin_ctzll(x) >> 3);
      break;
    }
    n += 8;
  }
  return min(n, rem);
}

// ---- sequences staged in LDS as 2-bit codes (A,C,G,T = 0..3; 16 bases per 32-bit word).  One probe
// compares 32 bases.  Only forward words are staged: the reverse aligner reads the same words from
// the far end, where the match count is the number of leading (instead of trailing) zero pairs.
constexpr int PROBE_PACKED = 32, PROBE_BYTES = 8;

__device__ __forceinline__ uint64_t lds_bits64(const uint32_t* base, int pos) {
  const uint32_t* w = base + (pos >> 4);
  const unsigned d0 = w[0], d1 = w[1], d2 = w[2];
  const unsigned sh = ((unsigned)pos & 15u) * 2u;
  const unsigned lo = __builtin_amdgcn_alignbit(d1, d0, sh);
  const unsigned hi = __builtin_amdgcn_alignbit(d2, d1, sh);
  return ((uint64_t)hi << 32) | lo;
}

// 16-base version for the first probe of a cell: most cells stop within a few bases, and the
// occasional longer run continues with the 32-base probes of extend_lcp_packed
constexpr int PROBE_FIRST = 16;
__device__ __forceinline__ unsigned lds_bits32(const uint32_t* base, int pos) {
  const uint32_t* w = base + (pos >> 4);
  return __builtin_amdgcn_alignbit(w[1], w[0], ((unsigned)pos & 15u) * 2u);
}
template <int DIR>
__device__ __forceinline__ int packed_first_count(const uint32_t* seq, const SubCtx& cx, int v, int h) {
  if (DIR == 0) {
    const unsigned x = lds_bits32(seq + cx.p_w0, cx.p_bit + v) ^ lds_bits32(seq + cx.t_w0, cx.t_bit + h);
    return (x ? __builtin_ctz(x) : 32) >> 1;  // (v_ffbl + v_min: 16 when all sixteen bases match)
  }
  // the 16 bases that END at forward position (len - v): start = bit + len - v - 16, biased by one pad word
  const unsigned x = lds_bits32(seq + cx.p_w0 - 1, cx.p_bit + cx.plen - v) ^ lds_bits32(seq + cx.t_w0 - 1, cx.t_bit + cx.tlen - h);
  return (x ? __builtin_clz(x) : 32) >> 1;
}

// The first probes of the four cells of a lane vector with their LDS reads issued together: left to itself the compiler
// reuses one register pair for every read and waits for each before issuing the next -- eight LDS round trips per step, in
// series, about 40 % of a wave's step time.  Here the reads of NB cells (2 NB ds_read2_b32) go out back to back and are
// waited for once.  (Inline asm: the compiler places no waits around what it cannot see, so the wait is part of the block;
// it also drains whatever LDS / scalar-memory operation of the compiler's own was in flight, which is harmless.)
// sub-problems whose lengths add up to no more than this run step by step throughout (a pass function's call + planning do not pay)
#ifndef AWV_MIN_PASS_LEN
#define AWV_MIN_PASS_LEN 1024
#endif
// 1: plan_multi plans a pass's steps without an LDS round trip per step (I/D metadata forwarded in scalar registers, M rows
// fetched an iteration ahead); 0: one plan_step per score
#ifndef AWV_PLAN_PIPELINED
#define AWV_PLAN_PIPELINED 1
#endif
// 1: the first sweep's x- and (o1 + e1)-lag sources are loaded straight into the chain registers (compute_rows_multi, ALIAS)
#ifndef AWV_CHAIN_ALIAS
#define AWV_CHAIN_ALIAS 1
#endif
// 1: 32-bit rows (long sequences): the (o1 + e1)-lag M rows of a chained pass's later sweeps live in LDS -- a lane vector is four
// registers there and a second set of chain registers does not fit; the staging region of the packed sequences is free whenever
// the sub-problem is too long to be staged (the top BiWFA levels of 100 kbp reads), and five scores x 64 lanes x 16 B per wave
// fit it.  Passes then chain three sweeps (15 scores) instead of two: 52 row vectors per 15 scores move through HBM instead of 70.
#ifndef AWV_LDS_CHAIN
#define AWV_LDS_CHAIN 1
#endif
// 1: the last sweep of a window issues the NEXT window's first-sweep loads, step by step, into the chain registers its own
// steps have just finished with (compute_rows_multi, XPREF; needs AWV_TAP_PREFETCH)
#ifndef AWV_WINDOW_PREFETCH
#define AWV_WINDOW_PREFETCH 0  // bit-exact, measured twice on config 2: -1.0 % and 0.0 % (1795 -> 1777 / 1797 ms): a window's first burst is not where the waves wait; off
#endif
// 1: chained sweeps load their (o2 + e2)-lag M rows one sweep ahead (compute_rows_multi, PREF)
#ifndef AWV_TAP_PREFETCH
#define AWV_TAP_PREFETCH 1
#endif
#ifndef AWV_PROBE_BATCH
#define AWV_PROBE_BATCH 2  // cells per batch of LDS reads: 0 = the compiler's own order, 2, 4 (config 2, same box: 1645 / 1629 / 1660 ms)
#endif
// experiment (off): wave priority raised while a window issues its row loads (1) and also while a step issues its probe reads (2)
#ifndef AWV_SETPRIO
#define AWV_SETPRIO 0
#endif
#if AWV_SETPRIO
#define AWV_PRIO(n) __builtin_amdgcn_s_setprio(n)
#else
#define AWV_PRIO(n) ((void)0)
#endif
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
template <int DIR, int NB>
__device__ __forceinline__ void packed_first_counts4(const uint32_t* seq, const SubCtx& cx, const int (&vv)[4], const int (&hh)[4], int (&nn)[4]) {
  static_assert(NB == 2 || NB == 4, "batch of two or four cells");
  // LDS byte address of word (pos >> 4) of the staged pattern / text (reverse: the 16 bases that END at len - v, one pad word back)
  const unsigned baseP = (unsigned)(uintptr_t)(seq + cx.p_w0 - (DIR ? 1 : 0)), baseT = (unsigned)(uintptr_t)(seq + cx.t_w0 - (DIR ? 1 : 0));
  const int offP = DIR ? cx.p_bit + cx.plen : cx.p_bit, offT = DIR ? cx.t_bit + cx.tlen : cx.t_bit;
  unsigned aP[4], aT[4], sP[4], sT[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int pp = DIR ? offP - vv[j] : offP + vv[j], pt = DIR ? offT - hh[j] : offT + hh[j];
    aP[j] = baseP + (unsigned)((pp >> 4) << 2);
    aT[j] = baseT + (unsigned)((pt >> 4) << 2);
    sP[j] = (unsigned)pp << 1;  // (v_alignbit takes the low five bits: ((pos & 15) * 2))
    sT[j] = (unsigned)pt << 1;
  }
  u32x2_t wP[4], wT[4];
  if constexpr (NB == 4) {
    asm volatile("ds_read2_b32 %0, %8 offset1:1\n\tds_read2_b32 %1, %9 offset1:1\n\tds_read2_b32 %2, %10 offset1:1\n\tds_read2_b32 %3, %11 offset1:1\n\t"
                 "ds_read2_b32 %4, %12 offset1:1\n\tds_read2_b32 %5, %13 offset1:1\n\tds_read2_b32 %6, %14 offset1:1\n\tds_read2_b32 %7, %15 offset1:1\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(wP[0]), "=&v"(wT[0]), "=&v"(wP[1]), "=&v"(wT[1]), "=&v"(wP[2]), "=&v"(wT[2]), "=&v"(wP[3]), "=&v"(wT[3])
                 : "v"(aP[0]), "v"(aT[0]), "v"(aP[1]), "v"(aT[1]), "v"(aP[2]), "v"(aT[2]), "v"(aP[3]), "v"(aT[3]));
  } else {
#pragma unroll
    for (int g = 0; g < 4; g += 2)
      asm volatile("ds_read2_b32 %0, %4 offset1:1\n\tds_read2_b32 %1, %5 offset1:1\n\tds_read2_b32 %2, %6 offset1:1\n\tds_read2_b32 %3, %7 offset1:1\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(wP[g]), "=&v"(wT[g]), "=&v"(wP[g + 1]), "=&v"(wT[g + 1])
                   : "v"(aP[g]), "v"(aT[g]), "v"(aP[g + 1]), "v"(aT[g + 1]));
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned x = __builtin_amdgcn_alignbit(wP[j][1], wP[j][0], sP[j]) ^ __builtin_amdgcn_alignbit(wT[j][1], wT[j][0], sT[j]);
    nn[j] = DIR == 0 ? ((x ? __builtin_ctz(x) : 32) >> 1) : ((x ? __builtin_clz(x) : 32) >> 1);
  }
}

// XOR of the next 32 pattern/text bases at (v, h) of direction DIR
template <int DIR>
__device__ __forceinline__ uint64_t packed_probe(const uint32_t* seq, const SubCtx& cx, int v, int h) {
  if (DIR == 0) return lds_bits64(seq + cx.p_w0, cx.p_bit + v) ^ lds_bits64(seq + cx.t_w0, cx.t_bit + h);
  // the 32 bases that END at forward position (len - v): start = bit + len - v - 32, biased by the 2 pad words
  return lds_bits64(seq + cx.p_w0 - 2, cx.p_bit + cx.plen - v) ^ lds_bits64(seq + cx.t_w0 - 2, cx.t_bit + cx.tlen - h);
}
template <int DIR>
__device__ __forceinline__ int packed_count(uint64_t x) {
  if (x == 0) return PROBE_PACKED;
  return DIR == 0 ? (int)(__builtin_ctzll(x) >> 1) : (int)(__builtin_clzll(x) >> 1);
}

template <int DIR>
__device__ __forceinline__ int extend_lcp_packed(const uint32_t* seq, const SubCtx& cx, int v, int h, unsigned& iters) {
  const int rem = min(cx.plen - v, cx.tlen - h);
  int n = 0;
  while (n < rem) {
    const int c = packed_count<DIR>(packed_probe<DIR>(seq, cx, v + n, h + n));
    ++iters;
    n += c;
    if (c < PROBE_PACKED) break;
  }
  return min(n, rem);
}

// ---- seq_mode 2 (round 3): a sub-problem too long to be staged probes the SAME 2-bit words where they lie in HBM instead of
// the raw bytes.  A 100 kbp pair's two sequences are 50 KB that way instead of 200 KB, and a 64-byte line holds 256 bases: a
// step's probes -- one pair of positions per diagonal, neighbouring diagonals a few bases apart -- touch a quarter of the
// lines, and the 128 pairs an XCD has in flight fit its 4 MB L2 four times better.  SubCtx::P[0] / T[0] then point at the word
// holding the sub-problem's first base (p_bit / t_bit = its position in that word, as in LDS); the raw pointers are not kept.
// The engine pads the packed arrays with two words in front (reverse probes of a sequence's first bases read up to two words
// below its first) and four behind; what a probe reads outside the sub-problem is cut off by the remaining length.
#ifndef AWV_GLOBAL_PACKED
#define AWV_GLOBAL_PACKED 1
#endif
typedef unsigned int u32x3_t __attribute__((ext_vector_type(3)));
typedef u32x2_t __attribute__((aligned(4))) u32x2_a4;
typedef u32x3_t __attribute__((aligned(4))) u32x3_a4;
__device__ __forceinline__ u32x2_t gw_load2(gseq_t base, int pos) {
  return *(const __attribute__((address_space(1))) u32x2_a4*)(base + (unsigned)((pos >> 4) << 2));
}
__device__ __forceinline__ uint64_t gw_bits64(gseq_t base, int pos) {
  const u32x3_t w = *(const __attribute__((address_space(1))) u32x3_a4*)(base + (unsigned)((pos >> 4) << 2));
  const unsigned sh = ((unsigned)pos & 15u) * 2u;
  return ((uint64_t)__builtin_amdgcn_alignbit(w.z, w.y, sh) << 32) | __builtin_amdgcn_alignbit(w.y, w.x, sh);
}
// word bases and bit offsets of the probes of direction DIR (reverse: the bases that END at forward position len - v, one pad
// word back for the 16-base probe, two for the 32-base one)
template <int DIR, int BACK>
__device__ __forceinline__ void gw_frame(const SubCtx& cx, gseq_t& bP, gseq_t& bT, int& oP, int& oT) {
  bP = cx.P[0] - (DIR ? 4 * BACK : 0);
  bT = cx.T[0] - (DIR ? 4 * BACK : 0);
  oP = DIR ? cx.p_bit + cx.plen : cx.p_bit;
  oT = DIR ? cx.t_bit + cx.tlen : cx.t_bit;
}
template <int DIR>
__device__ __forceinline__ void gw_first_counts4(const SubCtx& cx, const int (&vv)[4], const int (&hh)[4], int (&nn)[4]) {
  gseq_t bP, bT;
  int oP, oT;
  gw_frame<DIR, 1>(cx, bP, bT, oP, oT);
  u32x2_t wP[4], wT[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {  // eight loads in flight, one wait
    wP[j] = gw_load2(bP, DIR ? oP - vv[j] : oP + vv[j]);
    wT[j] = gw_load2(bT, DIR ? oT - hh[j] : oT + hh[j]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned sP = (unsigned)(DIR ? oP - vv[j] : oP + vv[j]) << 1, sT = (unsigned)(DIR ? oT - hh[j] : oT + hh[j]) << 1;
    const unsigned x = __builtin_amdgcn_alignbit(wP[j].y, wP[j].x, sP) ^ __builtin_amdgcn_alignbit(wT[j].y, wT[j].x, sT);
    nn[j] = DIR == 0 ? ((x ? __builtin_ctz(x) : 32) >> 1) : ((x ? __builtin_clz(x) : 32) >> 1);
  }
}
template <int DIR>
__device__ __forceinline__ int extend_lcp_gw(const SubCtx& cx, int v, int h, unsigned& iters) {
  gseq_t bP, bT;
  int oP, oT;
  gw_frame<DIR, 2>(cx, bP, bT, oP, oT);
  const int rem = min(cx.plen - v, cx.tlen - h);
  int n = 0;
  while (n < rem) {
    const uint64_t x = gw_bits64(bP, DIR ? oP - v - n : oP + v + n) ^ gw_bits64(bT, DIR ? oT - h - n : oT + h + n);
    ++iters;
    const int c = packed_count<DIR>(x);
    n += c;
    if (c < PROBE_PACKED) break;
  }
  return min(n, rem);
}

// Copies the sub-problem's packed words into LDS (all threads; ends with a barrier).  Falls back
// to probes from global memory -- of the packed words (seq_mode 2) when the pair is packable and the words do not fit, of the
// raw bytes when it is not packable.
template <typename OffT>
__device__ __forceinline__ void stage_sequences(const KParams& kp, const Lds<OffT>& lds, SubCtx& cx) {
  cx.seq_mode = 0;
  if (cx.Pw == nullptr || cx.Tw == nullptr) return;
  const int pw_first = cx.pb_abs >> 4, tw_first = cx.tb_abs >> 4;
  const int npw = ((cx.pb_abs + cx.plen + 15) >> 4) - pw_first;
  const int ntw = ((cx.tb_abs + cx.tlen + 15) >> 4) - tw_first;
  cx.p_bit = cx.pb_abs & 15;
  cx.t_bit = cx.tb_abs & 15;
  cx.p_w0 = 2;
  cx.t_w0 = cx.p_w0 + npw + 3 + 2;
  const int total_words = cx.t_w0 + ntw + 3;
  if (total_words * 4 > kp.lds_seq_bytes) {
    if (AWV_GLOBAL_PACKED) {
      cx.seq_mode = 2;
      cx.P[0] = cx.P[1] = (gseq_t)(cx.Pw + pw_first);
      cx.T[0] = cx.T[1] = (gseq_t)(cx.Tw + tw_first);
    }
    return;
  }
  cx.seq_mode = 1;
  const int t0 = cold_tid();
  for (int i = t0; i < npw; i += WG) lds.seq[cx.p_w0 + i] = cx.Pw[pw_first + i];
  for (int i = t0; i < ntw; i += WG) lds.seq[cx.t_w0 + i] = cx.Tw[tw_first + i];
  __syncthreads();
}

template <typename OffT>
__device__ __forceinline__ RowMeta get_meta(const KParams& kp, const Lds<OffT>& lds, int dir, int comp, int score) {
  if (score < 0) return ROW_EMPTY;
  return uni(meta_load(&lds.ring_meta[(dir * NCOMP + comp) * kp.ring + (score & (kp.ring - 1))]));
}

template <bool BASE, typename OffT>
__device__ __forceinline__ OffT* row_ptr(const KParams& kp, void* mem, int dir, int comp, int score) {
  if (score < 0) score = 0;  // null input rows are never dereferenced
  if (BASE) return (OffT*)mem + ((size_t)score * NCOMP + comp) * (size_t)kp.wb_cap;
  return (OffT*)mem + ((size_t)(dir * NCOMP + comp) * kp.ring + (size_t)(score & (kp.ring - 1))) * (size_t)kp.wcap;
}

__device__ __forceinline__ void acc_reset(Acc& a) {
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) { a.hull_lo[c] = INT_MAX; a.hull_hi[c] = INT_MIN; }
  a.maxak = 0;
  a.oob = 0;
  a.reach = 0;
}

struct StepPlan {  // uniform description of one compute-next call
  RowMeta src[7];  // Mx, O1, I1e, D1e, O2, I2e, D2e
  RowMeta hull[7]; // hull of the step that wrote each source row (= the M row's metadata at that score)
  int lo, hi;      // hull of the cells to compute (= predicted hull of the M row)
};

__device__ __forceinline__ void hull_add(RowMeta& h, const RowMeta& s, int dlo, int dhi) {
  h.lo = min(h.lo, s.lo + dlo);  // empty rows ({K_BIG, -K_BIG}) drop out of min/max by themselves
  h.hi = max(h.hi, s.hi + dhi);
}

// SCALAR: the hull arithmetic spelled out for the scalar unit (the passes' planning loop, plan_multi); the step-by-step loop of
// find_breakpoint_fn keeps the compiler's own selection -- there the extra scalar registers spill (425 spill instructions against 73)
template <bool P2, bool BASE, typename OffT, bool SCALAR = false>
__device__ __forceinline__ void plan_step(const KParams& kp, const Lds<OffT>& lds, int dir, int score, StepPlan& pl) {
  const DevPenalties& pn = kp.pen;
  // The nine metadata entries are fetched by nine lanes in one LDS round trip and handed out with
  // readlane: lanes 0..6 the sources (Mx, O1, I1e, D1e, O2, I2e, D2e), lanes 7 / 8 the M rows of the
  // I/D sources' scores (their steps' hulls).
  {
    const int lane = threadIdx.x & 63;
    const int lag = lane == 0 ? pn.x : lane == 1 ? pn.o1 + pn.e1 : (lane == 2 || lane == 3 || lane == 7) ? pn.e1 : lane == 4 ? pn.o2 + pn.e2 : pn.e2;
    const int comp = lane == 2 ? C_I1 : lane == 3 ? C_D1 : lane == 5 ? C_I2 : lane == 6 ? C_D2 : C_M;
    const int sc_l = score - lag;
    RowMeta mine = ROW_EMPTY;
    if (lane < 9 && sc_l >= 0 && (P2 || (lane != 4 && lane != 5 && lane != 6 && lane != 8)))
      mine = meta_load(&lds.ring_meta[(dir * NCOMP + comp) * kp.ring + (sc_l & (kp.ring - 1))]);
    auto hand = [&](int l) { return RowMeta{__builtin_amdgcn_readlane(mine.lo, l), __builtin_amdgcn_readlane(mine.hi, l)}; };
    pl.src[0] = hand(0);
    pl.src[1] = hand(1);
    pl.src[2] = hand(2);
    pl.src[3] = hand(3);
    pl.src[4] = pl.src[5] = pl.src[6] = ROW_EMPTY;
    if (P2) {
      pl.src[4] = hand(4);
      pl.src[5] = hand(5);
      pl.src[6] = hand(6);
    }
    // Every step stores whole lane vectors over its hull for all components (cells outside a
    // component's own range hold NULL), so while no row has been trimmed a source row can be masked
    // by whole lane vectors against the hull of the step that wrote it.
    pl.hull[0] = pl.src[0];
    pl.hull[1] = pl.src[1];
    pl.hull[2] = pl.hull[3] = hand(7);
    pl.hull[4] = pl.src[4];
    pl.hull[5] = pl.hull[6] = P2 ? hand(8) : ROW_EMPTY;
  }
  // (the score-0 row of a sub-problem that begins in an indel component has that cell and no M cell)
  if (score - pn.e1 == 0) { pl.hull[2] = pl.src[2]; pl.hull[3] = pl.src[3]; }
  if (P2 && score - pn.e2 == 0) { pl.hull[5] = pl.src[5]; pl.hull[6] = pl.src[6]; }
  // Predicted hulls (exact unless some value goes out of bounds, which the step detects and then
  // repairs in trim_pass): a cell of I (D) is non-NULL iff its left (right) source cell is, M iff
  // any source is (A.3).  They are written to the row metadata right away -- the slot of the new
  // score is not read by anyone during this step -- so nothing but lo/hi outlives the barrier.
  if constexpr (SCALAR) {
    // (on the scalar unit by construction -- s_min / s_max / s_add above; an empty source row {K_BIG, -K_BIG} drops out of the
    // minima / maxima by itself and leaves lo > hi behind where nothing feeds a component)
    int plo[NCOMP], phi[NCOMP];
    plo[C_I1] = s_add(s_min(pl.src[1].lo, pl.src[2].lo), 1);
    phi[C_I1] = s_add(s_max(pl.src[1].hi, pl.src[2].hi), 1);
    plo[C_D1] = s_add(s_min(pl.src[1].lo, pl.src[3].lo), -1);
    phi[C_D1] = s_add(s_max(pl.src[1].hi, pl.src[3].hi), -1);
    plo[C_I2] = plo[C_D2] = K_BIG;
    phi[C_I2] = phi[C_D2] = -K_BIG;
    int mlo = s_min(pl.src[0].lo, s_min(plo[C_I1], plo[C_D1])), mhi = s_max(pl.src[0].hi, s_max(phi[C_I1], phi[C_D1]));
    if (P2) {
      plo[C_I2] = s_add(s_min(pl.src[4].lo, pl.src[5].lo), 1);
      phi[C_I2] = s_add(s_max(pl.src[4].hi, pl.src[5].hi), 1);
      plo[C_D2] = s_add(s_min(pl.src[4].lo, pl.src[6].lo), -1);
      phi[C_D2] = s_add(s_max(pl.src[4].hi, pl.src[6].hi), -1);
      mlo = s_min(mlo, s_min(plo[C_I2], plo[C_D2]));
      mhi = s_max(mhi, s_max(phi[C_I2], phi[C_D2]));
    }
    plo[C_M] = mlo;
    phi[C_M] = mhi;
    pl.lo = mlo;
    pl.hi = mhi;
  #pragma unroll
    for (int c = 0; c < NCOMP; ++c) {
      meta_store_scalar(&lds.ring_meta[(dir * NCOMP + c) * kp.ring + (score & (kp.ring - 1))], plo[c], phi[c]);
      if (BASE && (threadIdx.x & 63) == 0) lds.meta_log[score * NCOMP + c] = plo[c] > phi[c] ? ROW_EMPTY : RowMeta{plo[c], phi[c]};  // history for the backtrace
    }
    return;
  }
  RowMeta pred[NCOMP];
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) pred[c] = ROW_EMPTY;
  hull_add(pred[C_I1], pl.src[1], 1, 1);
  hull_add(pred[C_I1], pl.src[2], 1, 1);
  hull_add(pred[C_D1], pl.src[1], -1, -1);
  hull_add(pred[C_D1], pl.src[3], -1, -1);
  if (P2) {
    hull_add(pred[C_I2], pl.src[4], 1, 1);
    hull_add(pred[C_I2], pl.src[5], 1, 1);
    hull_add(pred[C_D2], pl.src[4], -1, -1);
    hull_add(pred[C_D2], pl.src[6], -1, -1);
  }
  hull_add(pred[C_M], pl.src[0], 0, 0);
  hull_add(pred[C_M], pred[C_I1], 0, 0);
  hull_add(pred[C_M], pred[C_D1], 0, 0);
  if (P2) {
    hull_add(pred[C_M], pred[C_I2], 0, 0);
    hull_add(pred[C_M], pred[C_D2], 0, 0);
  }
  pl.lo = pred[C_M].lo;
  pl.hi = pred[C_M].hi;
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) {
    if (pred[c].lo > pred[c].hi) pred[c] = ROW_EMPTY;
    meta_store(&lds.ring_meta[(dir * NCOMP + c) * kp.ring + (score & (kp.ring - 1))], pred[c]);
    if (BASE && (threadIdx.x & 63) == 0) lds.meta_log[score * NCOMP + c] = pred[c];  // history for the backtrace
  }
}

// Neighbour-shifted copies of a lane vector, built in registers: lane l's vector holds diagonals
// k0..k0+3; shift_from_left gives k0-1..k0+2 (the missing element is lane l-1's last one), and
// shift_from_right gives k0+1..k0+4 (lane l+1's first one).  Full-wave DPP shifts (wave_shr:1 /
// wave_shl:1) move the halo element; lanes 0 / 63 receive nothing useful and are not productive.
// (bound_ctrl on: the lane without a source reads 0 and no "old" value has to be materialised)
__device__ __forceinline__ unsigned dpp_from_lower_lane(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ unsigned dpp_from_upper_lane(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x130, 0xf, 0xf, true); }
__device__ __forceinline__ RawVec<int16_t> shift_from_left(const RawVec<int16_t>& c) {
  RawVec<int16_t> o;
  o.w[0] = __builtin_amdgcn_alignbit(c.w[0], dpp_from_lower_lane(c.w[1]), 16);
  o.w[1] = __builtin_amdgcn_alignbit(c.w[1], c.w[0], 16);
  return o;
}
__device__ __forceinline__ RawVec<int16_t> shift_from_right(const RawVec<int16_t>& c) {
  RawVec<int16_t> o;
  o.w[0] = __builtin_amdgcn_alignbit(c.w[1], c.w[0], 16);
  o.w[1] = __builtin_amdgcn_alignbit(dpp_from_upper_lane(c.w[0]), c.w[1], 16);
  return o;
}
__device__ __forceinline__ RawVec<int32_t> shift_from_left(const RawVec<int32_t>& c) {
  RawVec<int32_t> o;
  o.w[0] = dpp_from_lower_lane(c.w[3]); o.w[1] = c.w[0]; o.w[2] = c.w[1]; o.w[3] = c.w[2];
  return o;
}
__device__ __forceinline__ RawVec<int32_t> shift_from_right(const RawVec<int32_t>& c) {
  RawVec<int32_t> o;
  o.w[0] = c.w[1]; o.w[1] = c.w[2]; o.w[2] = c.w[3]; o.w[3] = dpp_from_upper_lane(c.w[0]);
  return o;
}

// extend (A.4) of N = 4 * steps M cells of one lane (cell i lies on diagonal k0 + i % 4; m[i] < 0: NULL, left
// alone): the first probe (16 bases) of all N cells is issued together -- one LDS round trip for the lot,
// invalid cells probe offset 0, always readable -- then the few longer runs continue in a loop each
// PERCELL: the rare longer runs branch per cell on the lanes' condition itself instead of through a per-lane bit mask
// (fewer vector instructions; used by the multi-step passes -- in the step-by-step loop, whose register file is full,
// the grouped form keeps the allocation it has)
template <typename OffT, int N, bool PERCELL = false>
__device__ __forceinline__ void extend_cells_n(const Lds<OffT>& lds, const SubCtx& cx, int dir, int k0, int32_t (&m)[N], unsigned& ext_iters) {
  static_assert(N % 4 == 0 && N <= 32, "whole lane vectors");
  const gseq_t Pp = dir ? cx.P[1] : cx.P[0];  // (selects: `dir` may be a run-time value, SubCtx lives in registers)
  const gseq_t Tp = dir ? cx.T[1] : cx.T[0];
  const int plen = cx.plen, tlen = cx.tlen;
  int rr[N], vv[N], hh[N];
  const bool packed = cx.seq_mode == 1, gpacked = cx.seq_mode == 2;
  const uint32_t* seq = lds.seq;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const bool ok = m[j] >= 0;
    if constexpr (wenc_of<OffT>()) {  // m holds w = min(h, v): v = w + max(-k, 0), h = w + max(k, 0)
      const int k = k0 + (j & 3);
      vv[j] = ok ? m[j] + max(-k, 0) : 0;
      hh[j] = ok ? m[j] + max(k, 0) : 0;
    } else {
      vv[j] = ok ? m[j] - (k0 + (j & 3)) : 0;
      hh[j] = ok ? m[j] : 0;
    }
    rr[j] = ok ? min(plen - vv[j], tlen - hh[j]) : 0;
  }
  // uniform branches hoisted out of the per-cell code so the probes stay back to back
  unsigned cont = 0;
  if (packed) {
    int nn[N];
    if (dir == 0) {
#pragma unroll
      for (int j = 0; j < N; ++j) nn[j] = packed_first_count<0>(seq, cx, vv[j], hh[j]);
    } else {
#pragma unroll
      for (int j = 0; j < N; ++j) nn[j] = packed_first_count<1>(seq, cx, vv[j], hh[j]);
    }
    if (PERCELL) {
      ext_iters += N;
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const bool more = nn[j] == PROBE_FIRST && rr[j] > PROBE_FIRST;
        m[j] += min(nn[j], rr[j]);  // rr == 0 for NULL cells: unchanged
        if (more) {
          const int v = wenc_of<OffT>() ? m[j] + max(-(k0 + (j & 3)), 0) : m[j] - (k0 + (j & 3));
          const int h = wenc_of<OffT>() ? m[j] + max(k0 + (j & 3), 0) : m[j];
          m[j] += dir == 0 ? extend_lcp_packed<0>(seq, cx, v, h, ext_iters) : extend_lcp_packed<1>(seq, cx, v, h, ext_iters);
        }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
      cont |= (nn[j] == PROBE_FIRST && rr[j] > PROBE_FIRST) ? (1u << j) : 0u;
      m[j] += min(nn[j], rr[j]);  // rr == 0 for NULL cells: unchanged
    }
  } else if (gpacked) {
#pragma unroll
    for (int g = 0; g < N; g += 4) {
      int v4[4], h4[4], n4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v4[j] = vv[g + j]; h4[j] = hh[g + j]; }
      if (dir == 0) gw_first_counts4<0>(cx, v4, h4, n4);
      else gw_first_counts4<1>(cx, v4, h4, n4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        cont |= (n4[j] == PROBE_FIRST && rr[g + j] > PROBE_FIRST) ? (1u << (g + j)) : 0u;
        m[g + j] += min(n4[j], rr[g + j]);
      }
    }
  } else {
    uint64_t xx[N];
#pragma unroll
    for (int j = 0; j < N; ++j) xx[j] = ld64u(Pp + (unsigned)vv[j]) ^ ld64u(Tp + (unsigned)hh[j]);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      int n = xx[j] ? (int)(__builtin_ctzll(xx[j]) >> 3) : PROBE_BYTES;
      cont |= (n == PROBE_BYTES && rr[j] > PROBE_BYTES) ? (1u << j) : 0u;
      m[j] += min(n, rr[j]);
    }
  }
  ext_iters += N;
  if (cont) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      if (cont & (1u << j)) {
        const int v = wenc_of<OffT>() ? m[j] + max(-(k0 + (j & 3)), 0) : m[j] - (k0 + (j & 3));
        const int h = wenc_of<OffT>() ? m[j] + max(k0 + (j & 3), 0) : m[j];
        if (packed) m[j] += dir == 0 ? extend_lcp_packed<0>(seq, cx, v, h, ext_iters) : extend_lcp_packed<1>(seq, cx, v, h, ext_iters);
        else if (gpacked) m[j] += dir == 0 ? extend_lcp_gw<0>(cx, v, h, ext_iters) : extend_lcp_gw<1>(cx, v, h, ext_iters);
        else m[j] += extend_lcp(Pp, Tp, v, h, plen, tlen, ext_iters);
      }
    }
  }
}
template <typename OffT>
__device__ __forceinline__ void extend_cells(const Lds<OffT>& lds, const SubCtx& cx, int dir, int k0, int32_t (&m)[4], unsigned& ext_iters) {
  extend_cells_n<OffT, 4>(lds, cx, dir, k0, m, ext_iters);
}

// The passes' form of the extension (round 3): bounds test, NULL handling and extension in one go.  cand[j] is the M candidate
// of diagonal k0 + j as the recurrences leave it (any value; NULLs are negative), ok[j] says whether it is a cell of the
// wavefront (inside the step's hull and inside the matrix: hmin <= cand <= hmaxv[j], the largest offset inside the matrix on
// that diagonal).  The remaining length is hmaxv[j] - cand (= min(plen - v, tlen - h)), cells that are not ok probe position 0
// and come back OFF_NULL: 23 vector instructions per cell where bounds test + extend_cells_n took 34.
#ifndef AWV_LEAN_EXT
#define AWV_LEAN_EXT 1
#endif
// 1: a search's passes start at score 0 (sources of negative scores are empty rows to plan_multi); 0: round 2's rule, step by
// step until `scope` - 1 rows exist
#ifndef AWV_EARLY_PASSES
#define AWV_EARLY_PASSES 1
#endif
// 1: cells that are not part of the wavefront probe wherever their candidate value points instead of position 0 -- an LDS read
// outside the workgroup's allocation returns 0 on gfx950 (scratch/src/ldsoob.hip: every address tried, no fault), a raw-byte
// probe cannot do that (global memory): packed probes only
#ifndef AWV_OOB_PROBES
#define AWV_OOB_PROBES 0  // measured: 1686 ms against 1673 ms on config 2 (the scattered addresses cost more than the two selects per cell save)
#endif
// NBATCH: cells whose first-probe LDS reads go out together (packed_first_counts4; 0: the compiler's own order) -- chosen per
// instantiation by the caller: only where the registers are there (no spill inside the window loop, scratch/spill_audit.py)
template <typename OffT, int NBATCH>
__device__ __forceinline__ void extend_cells_lean(const Lds<OffT>& lds, const SubCtx& cx, int dir, int k0, const int32_t (&cand)[4], const bool (&ok)[4],
                                                  const int (&hmaxv)[4], int32_t (&m)[4], unsigned& ext_iters) {
  const uint32_t* seq = lds.seq;
  int rr[4], vv[4], hh[4], nn[4];
  // positions of the probes: PROBE_ANYWHERE -- the cell's candidate value as it is (cells that are not ok read somewhere, possibly
  // outside the LDS allocation, and their count is dropped); otherwise such cells probe position (0, 0)
  auto positions = [&](bool anywhere) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + j;
      const bool use = anywhere || ok[j];
      if constexpr (wenc_of<OffT>()) {  // cand holds w = min(h, v): v = w + max(-k, 0), h = w + max(k, 0)
        vv[j] = use ? cand[j] + max(-k, 0) : 0;
        hh[j] = use ? cand[j] + max(k, 0) : 0;
      } else {
        vv[j] = use ? cand[j] - k : 0;
        hh[j] = use ? cand[j] : 0;
      }
    }
  };
#pragma unroll
  for (int j = 0; j < 4; ++j) rr[j] = hmaxv[j] - cand[j];
  if (cx.seq_mode == 1) {
    positions(AWV_OOB_PROBES != 0);
    if (AWV_SETPRIO >= 2) AWV_PRIO(2);
    if constexpr (NBATCH != 0) {
      if (dir == 0) packed_first_counts4<0, NBATCH ? NBATCH : 4>(seq, cx, vv, hh, nn);
      else packed_first_counts4<1, NBATCH ? NBATCH : 4>(seq, cx, vv, hh, nn);
    } else if (dir == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) nn[j] = packed_first_count<0>(seq, cx, vv[j], hh[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) nn[j] = packed_first_count<1>(seq, cx, vv[j], hh[j]);
    }
    if (AWV_SETPRIO >= 2) AWV_PRIO(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = min(nn[j], rr[j]);
      const bool more = ok[j] && t == PROBE_FIRST;  // all sixteen bases matched and at least that many were left (exactly sixteen left: the loop below ends at once)
      int mo = cand[j] + t;
      if (more) {
        const int k = k0 + j;
        const int v = wenc_of<OffT>() ? mo + max(-k, 0) : mo - k;
        const int h = wenc_of<OffT>() ? mo + max(k, 0) : mo;
        mo += dir == 0 ? extend_lcp_packed<0>(seq, cx, v, h, ext_iters) : extend_lcp_packed<1>(seq, cx, v, h, ext_iters);
      }
      m[j] = ok[j] ? mo : OFF_NULL;
    }
    return;
  }
  positions(false);
  if (cx.seq_mode == 2) {  // packed words from global memory (too long for the LDS staging)
    if (dir == 0) gw_first_counts4<0>(cx, vv, hh, nn);
    else gw_first_counts4<1>(cx, vv, hh, nn);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = min(nn[j], rr[j]);
      const bool more = ok[j] && t == PROBE_FIRST;
      int mo = cand[j] + t;
      if (more) {
        const int k = k0 + j;
        const int v = wenc_of<OffT>() ? mo + max(-k, 0) : mo - k;
        const int h = wenc_of<OffT>() ? mo + max(k, 0) : mo;
        mo += dir == 0 ? extend_lcp_gw<0>(cx, v, h, ext_iters) : extend_lcp_gw<1>(cx, v, h, ext_iters);
      }
      m[j] = ok[j] ? mo : OFF_NULL;
    }
    return;
  }
  // raw-byte probes from global memory (sequences that are not pure upper-case ACGT)
  const gseq_t Pp = dir ? cx.P[1] : cx.P[0];
  const gseq_t Tp = dir ? cx.T[1] : cx.T[0];
  uint64_t xx[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) xx[j] = ld64u(Pp + (unsigned)vv[j]) ^ ld64u(Tp + (unsigned)hh[j]);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = xx[j] ? (int)(__builtin_ctzll(xx[j]) >> 3) : PROBE_BYTES;
    const int t = min(n, rr[j]);
    const bool more = ok[j] && t == PROBE_BYTES;
    int mo = cand[j] + t;
    if (more) {
      const int k = k0 + j;
      const int v = wenc_of<OffT>() ? mo + max(-k, 0) : mo - k;
      const int h = wenc_of<OffT>() ? mo + max(k, 0) : mo;
      mo += extend_lcp(Pp, Tp, v, h, cx.plen, cx.tlen, ext_iters);
    }
    m[j] = ok[j] ? mo : OFF_NULL;
  }
}

// One compute-next + extend step of one direction (A.3 + A.4).  Every lane owns VEC consecutive
// diagonals and loads ONE naturally aligned vector per source row; the k-1 / k+1 neighbours are
// assembled in registers from the adjacent lanes (shift_from_left / shift_from_right), so lanes 0 and
// 63 of a window only feed their neighbours: a window produces 62 * VEC = 248 columns and windows
// overlap by one lane vector on each side.  Output rows are written untrimmed; max antidiagonal /
// oob land in `acc`.  Returns the number of cells.
template <bool P2, bool BASE, typename OffT>
__device__ __forceinline__ int compute_row(const KParams& kp, Shared& sh, const Lds<OffT>& lds, const SubCtx& cx, rsrc_t rs,
                                           int dir, int score, const StepPlan& pl, bool dirty, Acc& acc, unsigned& ext_iters) {
  static_assert(WG % 64 == 0, "whole waves");
  constexpr int NWAVES = (DIRSPLIT && !BASE) ? 1 : WG / 64;  // a row's windows are dealt round-robin to the workgroup's waves (direction split: a row belongs to one wave)
  constexpr int VEC = OffTraits<OffT>::VEC;
  static_assert(VEC == 4, "lane vectors are 4 diagonals wide");
  constexpr int WSPAN = 64 * VEC;
  static_assert(WSPAN == 256, "window = 256 columns (64 lane vectors)");
  constexpr int PROD = 62;                // productive lanes 1..62
  constexpr int WSTRIDE = PROD * VEC;     // 248 new columns per window
  constexpr int ESZ = (int)sizeof(OffT);
  const DevPenalties& pn = kp.pen;
  const int lane = threadIdx.x & 63;
  const int kmin = cx.kmin[dir];
  const int lo = pl.lo, hi = pl.hi;
  if (lo > hi) return 0;  // null step: all rows empty
  if (lo - kmin < WSPAN + 1 || hi - kmin + WSPAN + VEC + 2 > cx.wcols) {  // whole-wave vector accesses stay in the row
    sh.error = ST_CAPACITY;
    return 0;
  }
  const int sMx = row_off<BASE, OffT>(kp, dir, C_M, score - pn.x);
  const int sO1 = row_off<BASE, OffT>(kp, dir, C_M, score - pn.o1 - pn.e1);
  const int sI1 = row_off<BASE, OffT>(kp, dir, C_I1, score - pn.e1);
  const int sD1 = row_off<BASE, OffT>(kp, dir, C_D1, score - pn.e1);
  const int sO2 = P2 ? row_off<BASE, OffT>(kp, dir, C_M, score - pn.o2 - pn.e2) : 0;
  const int sI2 = P2 ? row_off<BASE, OffT>(kp, dir, C_I2, score - pn.e2) : 0;
  const int sD2 = P2 ? row_off<BASE, OffT>(kp, dir, C_D2, score - pn.e2) : 0;
  const int tM = row_off<BASE, OffT>(kp, dir, C_M, score);
  const int tI1 = row_off<BASE, OffT>(kp, dir, C_I1, score);
  const int tD1 = row_off<BASE, OffT>(kp, dir, C_D1, score);
  const int tI2 = P2 ? row_off<BASE, OffT>(kp, dir, C_I2, score) : 0;
  const int tD2 = P2 ? row_off<BASE, OffT>(kp, dir, C_D2, score) : 0;
  const int plen = cx.plen, tlen = cx.tlen;
  const int vcap = wenc_of<OffT>() ? min(plen, tlen) : tlen;  // largest stored value inside the matrix
  const int colLo = lo - kmin, colHi = hi - kmin;
  int lane_maxak = 0;
  bool lane_oob = false;
  const bool productive = lane >= 1 && lane <= PROD;
  // Windows that need no masking at all, as a range of window origins [int_lo, int_hi], found once
  // per row.  With trimmed rows around (`dirty`): the productive columns and their +-1 halo lie
  // inside every source's own range; otherwise: all 64 lane vectors lie inside the (lane-aligned)
  // hull of every source's step.
  int int_lo = INT_MIN, int_hi = INT_MAX;
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    if (!P2 && r >= 4) break;
    if (dirty) {
      int_lo = max(int_lo, pl.src[r].lo - kmin + 1 - VEC);          // src.lo <= cb + VEC + kmin - 1
      int_hi = min(int_hi, pl.src[r].hi - kmin - (PROD + 1) * VEC);  // cb + (PROD + 1) * VEC + kmin <= src.hi
    } else {
      int_lo = max(int_lo, (pl.hull[r].lo - kmin) & ~(VEC - 1));
      int_hi = min(int_hi, ((pl.hull[r].hi - kmin) | (VEC - 1)) - 63 * VEC);
    }
  }
  for (int cb = (colLo & ~(VEC - 1)) - VEC + (NWAVES > 1 ? (int)(threadIdx.x >> 6) * WSTRIDE : 0); cb + VEC <= colHi;
       cb += WSTRIDE * NWAVES) {  // lane 1 owns columns cb+4..cb+7
    const int c0 = cb + lane * VEC;
    const int k0 = c0 + kmin;
    const int voff = c0 * ESZ;  // naturally aligned lane vector
    const bool interior = cb >= int_lo && cb <= int_hi;  // wave-uniform
    // All row loads are issued back to back with no control flow in between (absent rows read a
    // valid dummy row), so the wave waits for memory once; mask / shift afterwards.
    const unsigned long long tc0 = PROF_NOW();
    // lanes whose four diagonals lie wholly outside [lo, hi] neither compute nor store (their values
    // are NULL by construction and nobody reads outside a row's range); their neighbours still load
    const bool lane_on = productive && c0 + VEC > colLo && c0 <= colHi;
    const bool load_on = c0 + 2 * VEC > colLo && c0 - VEC <= colHi;
    RawVec<OffT> cMx{}, cO1{}, cI1{}, cD1{}, cO2{}, cI2{}, cD2{};
    if (load_on) {
      cMx = buf_load_raw<OffT>(rs, voff, sMx);
      cO1 = buf_load_raw<OffT>(rs, voff, sO1);
      // an I/D row is read by exactly one later step (score + e): non-temporal, so that the dead
      // lines leave L2 / Infinity Cache first (the M rows are read three times, up to `scope` steps later)
      cI1 = buf_load_raw<OffT, 2>(rs, voff, sI1);
      cD1 = buf_load_raw<OffT, 2>(rs, voff, sD1);
      if (P2) {
        cO2 = buf_load_raw<OffT>(rs, voff, sO2);
        cI2 = buf_load_raw<OffT, 2>(rs, voff, sI2);
        cD2 = buf_load_raw<OffT, 2>(rs, voff, sD2);
      }
    }
    PROF_DRAIN();
    PROF_ADD_L(STAT_T_CR_LOAD, tc0);
    const unsigned long long tc1 = PROF_NOW();
    int32_t m[VEC];
    if (!interior && !dirty) {  // edge windows: lane vectors outside the hull of a source's step hold nothing (or stale data)
      auto lmask = [&](const RowMeta& h, RawVec<OffT>& v) {
        const bool keep = c0 >= ((h.lo - kmin) & ~(VEC - 1)) && c0 <= ((h.hi - kmin) | (VEC - 1));  // (empty: never)
        const unsigned nullw = sizeof(OffT) == 2 ? ((unsigned)(unsigned short)NULL16) * 0x00010001u : (unsigned)OFF_NULL;
#pragma unroll
        for (int r = 0; r < (int)(sizeof(v.w) / sizeof(v.w[0])); ++r) v.w[r] = keep ? v.w[r] : nullw;
      };
      lmask(pl.hull[0], cMx);
      lmask(pl.hull[1], cO1);
      lmask(pl.hull[2], cI1);
      lmask(pl.hull[3], cD1);
      if (P2) {
        lmask(pl.hull[4], cO2);
        lmask(pl.hull[5], cI2);
        lmask(pl.hull[6], cD2);
      }
    } else if (!interior) {  // some row was trimmed: NULL out, element by element, what lies outside each row's own range
      auto mask = [&](const RowMeta& mm, RawVec<OffT>& v) {
        if constexpr (sizeof(OffT) == 2) {
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int ka = k0 + 2 * r;
            const unsigned keep = ((ka >= mm.lo && ka <= mm.hi) ? 0x0000FFFFu : 0u) |
                                  ((ka + 1 >= mm.lo && ka + 1 <= mm.hi) ? 0xFFFF0000u : 0u);
            const unsigned nullw = ((unsigned)(unsigned short)NULL16) * 0x00010001u;
            v.w[r] = (v.w[r] & keep) | (nullw & ~keep);
          }
        } else {
#pragma unroll
          for (int j = 0; j < VEC; ++j) v.w[j] = (k0 + j >= mm.lo && k0 + j <= mm.hi) ? v.w[j] : (unsigned)OFF_NULL;
        }
      };
      mask(pl.src[0], cMx);
      mask(pl.src[1], cO1);
      mask(pl.src[2], cI1);
      mask(pl.src[3], cD1);
      if (P2) {
        mask(pl.src[4], cO2);
        mask(pl.src[5], cI2);
        mask(pl.src[6], cD2);
      }
    }
    const int hbase = plen + k0;
    if constexpr (sizeof(OffT) == 2) {
      // ---- 16-bit rows: the DP runs on packed pairs (v_pk_max_i16 / v_pk_add_u16 / v_pk_min_i16),
      // two diagonals per instruction.  All values fit 16 bits (offsets <= tlen + 1, NULL16 = -16384,
      // at most +1 per step before the store re-canonicalises), so the ordering and results are
      // those of the 32-bit arithmetic used for 32-bit rows below.
      typedef short s2 __attribute__((ext_vector_type(2)));
      typedef unsigned short us2 __attribute__((ext_vector_type(2)));
      auto as2 = [](unsigned w) { return __builtin_bit_cast(s2, w); };
      auto asu = [](s2 v) { return __builtin_bit_cast(unsigned, v); };
      const s2 one = {1, 1};
      const short tl1 = (short)(vcap + 1);
      const s2 tlen1 = {tl1, tl1};
      const us2 null_u = {(unsigned short)NULL16, (unsigned short)NULL16};
      // what buf_store_vec does per element, v < 0 ? NULL16 : min(v, tlen + 1), in two packed ops:
      // every negative value is NULL16 + n (n >= 0), i.e. >= 0xC000 as an unsigned half
      auto canon = [&](s2 v) {
        const s2 c = __builtin_elementwise_min(v, tlen1);
        return __builtin_bit_cast(s2, __builtin_elementwise_min(__builtin_bit_cast(us2, c), null_u));
      };
      // max first, neighbour shift after: max(O[k-1], I[k-1]) = (max(O, I))[k-1] -- one shift per gap kind instead of two
      auto pkmax = [&](const RawVec<OffT>& a, const RawVec<OffT>& b) {
        RawVec<OffT> o;
        o.w[0] = asu(__builtin_elementwise_max(as2(a.w[0]), as2(b.w[0])));
        o.w[1] = asu(__builtin_elementwise_max(as2(a.w[1]), as2(b.w[1])));
        return o;
      };
      const RawVec<OffT> sI1 = shift_from_left(pkmax(cO1, cI1)), sD1 = shift_from_right(pkmax(cO1, cD1));
      RawVec<OffT> sI2{}, sD2{};
      if (P2) {
        sI2 = shift_from_left(pkmax(cO2, cI2));
        sD2 = shift_from_right(pkmax(cO2, cD2));
      }
      RawVec<OffT> oI1, oD1, oI2, oD2;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        // AWV_WIDE16 (stored w = h - max(k, 0)): an insertion into diagonal k adds 1 only for k <= 0, a deletion only for k >= 0
        const int ka = k0 + 2 * r;
        const s2 incI = WENC ? s2{(short)(ka <= 0), (short)(ka + 1 <= 0)} : one;
        const s2 incD = {(short)(ka >= 0), (short)(ka + 1 >= 0)};
        const s2 ins1 = as2(sI1.w[r]) + incI;
        const s2 del1 = WENC ? as2(sD1.w[r]) + incD : as2(sD1.w[r]);
        s2 ins = ins1, del = del1;
        oI1.w[r] = asu(canon(ins1));
        oD1.w[r] = asu(canon(del1));
        if (P2) {
          const s2 ins2 = as2(sI2.w[r]) + incI;
          const s2 del2 = WENC ? as2(sD2.w[r]) + incD : as2(sD2.w[r]);
          ins = __builtin_elementwise_max(ins, ins2);
          del = __builtin_elementwise_max(del, del2);
          oI2.w[r] = asu(canon(ins2));
          oD2.w[r] = asu(canon(del2));
        }
        const s2 mm2 = __builtin_elementwise_max(del, __builtin_elementwise_max(as2(cMx.w[r]) + one, ins));
        // bounds on sign-extended halves, as the 32-bit path below: in bounds <=> 0 <= mm <= hmax
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int mm = (int)mm2[e];
          const int hmax = WENC ? wenc_max(ka + e, plen, tlen, vcap) : clamp_from_m1(hbase + 2 * r + e, tlen);
          lane_oob |= lane_on && mm > hmax;
          m[2 * r + e] = (mm > hmax || mm < 0 || !lane_on) ? OFF_NULL : mm;
        }
      }
      if (lane_on) {
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        auto st = [&](int soff, const RawVec<OffT>& v) {
          u32x2 w2;
          w2[0] = v.w[0];
          w2[1] = v.w[1];
          __builtin_amdgcn_raw_buffer_store_b64(w2, rs, voff, soff, 0);
        };
        st(tI1, oI1);
        st(tD1, oD1);
        if (P2) { st(tI2, oI2); st(tD2, oD2); }
      }
    } else {
      const RawVec<OffT> rO1l = shift_from_left(cO1), rO1r = shift_from_right(cO1);
      const RawVec<OffT> rI1 = shift_from_left(cI1), rD1 = shift_from_right(cD1);
      RawVec<OffT> rO2l{}, rO2r{}, rI2{}, rD2{};
      if (P2) {
        rO2l = shift_from_left(cO2);
        rO2r = shift_from_right(cO2);
        rI2 = shift_from_left(cI2);
        rD2 = shift_from_right(cD2);
      }
      int32_t ins1[VEC], del1[VEC], ins2[VEC], del2[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        ins1[j] = max((int32_t)rO1l.w[j], (int32_t)rI1.w[j]) + 1;
        del1[j] = max((int32_t)rO1r.w[j], (int32_t)rD1.w[j]);
        int32_t ins = ins1[j], del = del1[j];
        if (P2) {
          ins2[j] = max((int32_t)rO2l.w[j], (int32_t)rI2.w[j]) + 1;
          del2[j] = max((int32_t)rO2r.w[j], (int32_t)rD2.w[j]);
          ins = max(ins, ins2[j]);
          del = max(del, del2[j]);
        } else {
          ins2[j] = del2[j] = OFF_NULL;
        }
        const int32_t mm = max(del, max((int32_t)cMx.w[j] + 1, ins));
        // in-bounds <=> 0 <= value <= hmax.  mm is the max of the cell's five values, so "some
        // non-NULL value of this cell is out of bounds" <=> mm > hmax (any negative is a NULL(+n)).
        const int hmax = clamp_from_m1(hbase + j, tlen);
        lane_oob |= lane_on && mm > hmax;
        m[j] = (mm > hmax || mm < 0 || !lane_on) ? OFF_NULL : mm;
      }
      if (lane_on) {
        buf_store_vec<OffT>(rs, voff, tI1, ins1, tlen);
        buf_store_vec<OffT>(rs, voff, tD1, del1, tlen);
        if (P2) {
          buf_store_vec<OffT>(rs, voff, tI2, ins2, tlen);
          buf_store_vec<OffT>(rs, voff, tD2, del2, tlen);
        }
      }
    }
    PROF_DRAIN();
    PROF_ADD_L(STAT_T_CR_ALU, tc1);
    const unsigned long long tc2 = PROF_NOW();
    extend_cells<OffT>(lds, cx, dir, k0, m, ext_iters);
    PROF_DRAIN();
    PROF_ADD_L(STAT_T_CR_EXTEND, tc2);
    const unsigned long long tc3 = PROF_NOW();
    int it_maxak = 0;
#pragma unroll
    for (int j = 0; j < VEC; ++j)
      if (m[j] >= 0) it_maxak = max(it_maxak, wenc_of<OffT>() ? 2 * m[j] + abs(k0 + j) : 2 * m[j] - (k0 + j));  // h + v
    lane_maxak = max(lane_maxak, it_maxak);
    if (lane_on) buf_store_vec<OffT>(rs, voff, tM, m, vcap);
    PROF_DRAIN();
    PROF_ADD_L(STAT_T_CR_STORE, tc3);
  }
  if (threadIdx.x == 0) { const unsigned nw = (unsigned)((colHi - ((colLo & ~(VEC - 1)) - VEC) - VEC) / WSTRIDE + 1); if (BASE) sh.win_base += nw; else sh.win_single += nw; }
  const unsigned long long tc4 = PROF_NOW();
  const int wmax = wave_max_i32(lane_maxak);  // one reduction per row: the row's max antidiagonal
  const bool woob = __any(lane_oob);
  if (lane == 0) {
    atomicMax(&acc.maxak, wmax);
    if (woob) acc.oob = 1;
  }
  PROF_ADD_L(STAT_T_CR_REDUCE, tc4);
  return hi - lo + 1;
}

// ---------------------------------------------------------------------------------------------
// Multi-step windows (round 2): while the two searches are far apart and no row has been trimmed, a
// window advances T <= TMAX consecutive scores in one pass.  The I/D rows, which only ever feed the
// next e1 / e2 steps of their own diagonal neighbourhood, stay in registers for the whole pass; only
// the M rows (read again x, o1+e1 and o2+e2 steps later) and the pass's last e1 / e2 I/D rows go to
// HBM.  With T <= min(x, o1+e1, o2+e2) every M source of the pass was written by an earlier pass, so
// all of its row loads (2(e1+e2) I/D vectors + 3T M vectors) are issued up front in one burst, one
// memory round trip per T steps instead of one per step.  Per T steps a window moves
// 2(e1+e2) + 3T loads + T + 2(e1+e2) stores instead of 12T (default penalties, T = 5: 32 instead of
// 60 lane vectors).  Values a step cannot know -- the +-1 neighbours beyond the window -- creep in by
// one column per step from both edges, so HALO = ceil(TMAX/4) lanes on either side are recomputed by
// the neighbouring windows and never stored: a window produces (64 - 2 HALO) * 4 columns.
// What the overlap search of phase 2 needs (the I/D rows of the last `scope` scores) is not kept by
// such passes: find_breakpoint leaves this mode a safe margin before the searches can meet and a
// sub-problem that still runs out of I/D history is searched again step by step (BP_RESTART).
// A pass in which some value leaves the matrix (the trimmed-hull case) is discarded and redone step by
// step: nothing it wrote aliases a row a step-by-step redo reads (ring >= scope + TMAX + 1).
// ---------------------------------------------------------------------------------------------
#ifndef AWV_TMAX
#define AWV_TMAX 5
#endif
constexpr int TMAX = AWV_TMAX;  // steps per sweep (one burst of row loads)
#ifndef AWV_CHAIN_MAX
#define AWV_CHAIN_MAX 3
#endif
// Chained sweeps: with T = TMAX = x and o1 + e1 = 2 x (the default scores: 5 and 10) the M rows a sweep needs
// from 5 and 10 scores back are exactly the previous one / two sweeps' own results, so up to CHAIN_MAX sweeps
// of one window run back to back with those rows (and the I/D queues) in registers: per 15 scores a window
// then loads 6 + 15 + 10 + 5 + 15 (gap-open-2 sources, always from memory) ... see DESIGN.md section 4.
constexpr int CHAIN_MAX = AWV_CHAIN_MAX;
#ifndef AWV_CHAIN_MAX32
#define AWV_CHAIN_MAX32 2
#endif
constexpr int CHAIN_MAX32 = AWV_CHAIN_MAX32;  // 32-bit rows: a sweep's M rows are 20 registers, so only the previous sweep's are kept
#ifndef AWV_TMAX32
#define AWV_TMAX32 5
#endif
constexpr int TMAX32 = AWV_TMAX32;  // steps per sweep with 32-bit rows (their M sources are loaded one step ahead, not all up front)
constexpr int MSTEPS = 16;  // most steps one pass can cover (a lane table entry per step and per source)
static_assert(TMAX >= 2 && TMAX <= 8 && TMAX * CHAIN_MAX <= MSTEPS - 1, "pass length");

struct MultiPlan {
  int Tn;                    // steps in this pass = TMAX-step sweeps * nh (the last sweep may be shorter only when nh == 1)
  int nh;                    // sweeps chained in this pass
  int halo;                  // unproductive lanes on either side of a window = ceil(Tn / 4)
  int lo_min, hi_max;        // union of the step hulls
  int vlo, vhi;              // VGPR: lane r = source r's stored hull as lane-aligned columns (empty: BIG / -BIG)
  int vslo, vshi;            // VGPR: lane t = predicted hull of step t (score s0 + 1 + t) in diagonals; lo > hi: empty
  int int_lo, int_hi;        // window origins whose 64 lane vectors lie inside every source's stored hull
  int end_comp, end_col;     // base case: component and column of the end cell (k = tlen - plen); end_col < 0: no termination test
  int lds_chain;             // 32-bit rows: 1 = the previous sweep's x-lag rows are kept in LDS (Lds::seq, unused by this sub-problem) for the next sweep
};

// source r of a pass that starts after score s0: which component at which score.  Lanes 0 .. NS0-1 are the
// I/D rows the pass begins with; lane NS0 + 16 w + step is M source w (0: s - x, 1: s - o1 - e1, 2: s - o2 - e2)
// of step `step`.
template <bool P2, int E1, int E2>
__device__ __forceinline__ void multi_source(const DevPenalties& pn, int s0, int r, int& comp, int& score) {
  constexpr int NS0 = 2 * E1 + (P2 ? 2 * E2 : 0);
  if (r < E1) { comp = C_I1; score = s0 - E1 + 1 + r; }
  else if (r < 2 * E1) { comp = C_D1; score = s0 - E1 + 1 + (r - E1); }
  else if (P2 && r < 2 * E1 + E2) { comp = C_I2; score = s0 - E2 + 1 + (r - 2 * E1); }
  else if (P2 && r < NS0) { comp = C_D2; score = s0 - E2 + 1 + (r - 2 * E1 - E2); }
  else {
    const int w = (r - NS0) >> 4, step = (r - NS0) & 15;
    comp = C_M;
    score = s0 + 1 + step - (w == 0 ? pn.x : w == 1 ? pn.o1 + pn.e1 : pn.o2 + pn.e2);
  }
}

template <bool P2, typename OffT, int E1, int E2, bool BASE>
__device__ __forceinline__ void plan_multi(const KParams& kp, const Lds<OffT>& lds, const SubCtx& cx, int dir, int s0, int Tn, int nh, MultiPlan& mp) {
  constexpr int NS0 = 2 * E1 + (P2 ? 2 * E2 : 0);
  constexpr int NT = P2 ? 3 : 2;
  static_assert(NS0 + 16 * NT <= 64, "one lane per source");
  const DevPenalties& pn = kp.pen;
  const int kmin = dir ? cx.kmin[1] : cx.kmin[0];
  const int lane = threadIdx.x & 63;
  mp.Tn = Tn;
  mp.nh = nh;
  mp.halo = (Tn + 3) >> 2;
  mp.lo_min = K_BIG;
  mp.hi_max = -K_BIG;
  mp.end_comp = C_M;
  mp.end_col = -1;
  mp.lds_chain = 0;
  mp.vslo = 1;
  mp.vshi = 0;
#if AWV_PLAN_PIPELINED
  // The predicted metadata of the pass's steps, one after the other (step t's I/D rows feed step t + 1 / t + 2), without an LDS
  // round trip per step: the I/D rows of the last e1 / e2 scores travel in scalar registers (queues like compute_rows_multi's),
  // and the three M rows a step reads -- x, o1 + e1 and o2 + e2 scores back, all written at least two iterations ago -- are
  // fetched one iteration ahead by lanes 0..2.  (plan_step did one dependent LDS round trip per score: 15 in a row per pass and
  // direction, a quarter of a narrow sub-problem's pass.)  All arithmetic on the scalar unit (s_min / s_max / s_add).
  {
    typedef typename MetaTraits<OffT>::Stored MetaStored;
    const int rmask = kp.ring - 1;
    MetaStored* const meta = lds.ring_meta + (size_t)dir * NCOMP * kp.ring;
    auto ld_mlags = [&](int score) -> RowMeta {  // lane 0 / 1 / 2: the M row x / o1 + e1 / o2 + e2 scores before `score`
      const int lag = lane == 0 ? pn.x : lane == 1 ? pn.o1 + pn.e1 : pn.o2 + pn.e2;
      const int sl = score - lag;
      RowMeta m = ROW_EMPTY;
      if (lane < (P2 ? 3 : 2) && sl >= 0) m = meta_load(&meta[C_M * kp.ring + (sl & rmask)]);
      return m;
    };
    // the I/D rows the pass begins with (scores s0 - e + 1 .. s0): lanes 0 .. NS0 - 1 fetch them in one LDS round trip -- the lane
    // order of multi_source is the queues' -- together with the first step's M rows (lanes 0..2 of a second access)
    RowMeta q0 = ROW_EMPTY;
    {
      int qc, qs;
      multi_source<P2, E1, E2>(pn, s0, lane, qc, qs);
      if (lane < NS0 && qs >= 0) q0 = meta_load(&meta[qc * kp.ring + (qs & rmask)]);
    }
    RowMeta cur = ld_mlags(s0 + 1);
    auto hand = [&](int l) { return RowMeta{__builtin_amdgcn_readlane(q0.lo, l), __builtin_amdgcn_readlane(q0.hi, l)}; };
    RowMeta qI1[E1], qD1[E1], qI2[E2], qD2[E2];
#pragma unroll
    for (int j = 0; j < E1; ++j) { qI1[j] = hand(j); qD1[j] = hand(E1 + j); }
#pragma unroll
    for (int j = 0; j < E2; ++j) {
      qI2[j] = P2 ? hand(2 * E1 + j) : ROW_EMPTY;
      qD2[j] = P2 ? hand(2 * E1 + E2 + j) : ROW_EMPTY;
    }
    for (int t = 0; t < Tn; ++t) {
      const int score = s0 + 1 + t;
      const RowMeta nxt = ld_mlags(score + 1);  // (in flight while this step's arithmetic runs)
      const int Mx_lo = __builtin_amdgcn_readlane(cur.lo, 0), Mx_hi = __builtin_amdgcn_readlane(cur.hi, 0);
      const int O1_lo = __builtin_amdgcn_readlane(cur.lo, 1), O1_hi = __builtin_amdgcn_readlane(cur.hi, 1);
      int plo[NCOMP], phi[NCOMP];
      plo[C_I1] = s_add(s_min(O1_lo, qI1[0].lo), 1);
      phi[C_I1] = s_add(s_max(O1_hi, qI1[0].hi), 1);
      plo[C_D1] = s_add(s_min(O1_lo, qD1[0].lo), -1);
      phi[C_D1] = s_add(s_max(O1_hi, qD1[0].hi), -1);
      plo[C_I2] = plo[C_D2] = K_BIG;
      phi[C_I2] = phi[C_D2] = -K_BIG;
      int mlo = s_min(Mx_lo, s_min(plo[C_I1], plo[C_D1])), mhi = s_max(Mx_hi, s_max(phi[C_I1], phi[C_D1]));
      if (P2) {
        const int O2_lo = __builtin_amdgcn_readlane(cur.lo, 2), O2_hi = __builtin_amdgcn_readlane(cur.hi, 2);
        plo[C_I2] = s_add(s_min(O2_lo, qI2[0].lo), 1);
        phi[C_I2] = s_add(s_max(O2_hi, qI2[0].hi), 1);
        plo[C_D2] = s_add(s_min(O2_lo, qD2[0].lo), -1);
        phi[C_D2] = s_add(s_max(O2_hi, qD2[0].hi), -1);
        mlo = s_min(mlo, s_min(plo[C_I2], plo[C_D2]));
        mhi = s_max(mhi, s_max(phi[C_I2], phi[C_D2]));
      }
      plo[C_M] = mlo;
      phi[C_M] = mhi;
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) {
        meta_store_scalar(&meta[c * kp.ring + (score & rmask)], plo[c], phi[c]);
        if (BASE && lane == 0) lds.meta_log[score * NCOMP + c] = plo[c] > phi[c] ? ROW_EMPTY : RowMeta{plo[c], phi[c]};  // history for the backtrace
      }
      // this score's I/D rows enter the queues (an empty one as {K_BIG, -K_BIG}: it drops out of the next minima / maxima)
      auto norm = [](int lo, int hi) { return lo > hi ? ROW_EMPTY : RowMeta{lo, hi}; };
#pragma unroll
      for (int j = 0; j + 1 < E1; ++j) { qI1[j] = qI1[j + 1]; qD1[j] = qD1[j + 1]; }
      qI1[E1 - 1] = norm(plo[C_I1], phi[C_I1]);
      qD1[E1 - 1] = norm(plo[C_D1], phi[C_D1]);
#pragma unroll
      for (int j = 0; j + 1 < E2; ++j) { qI2[j] = qI2[j + 1]; qD2[j] = qD2[j + 1]; }
      qI2[E2 - 1] = norm(plo[C_I2], phi[C_I2]);
      qD2[E2 - 1] = norm(plo[C_D2], phi[C_D2]);
      if (lane == t) { mp.vslo = mlo; mp.vshi = mhi; }
      if (mlo <= mhi) {
        mp.lo_min = min(mp.lo_min, mlo);
        mp.hi_max = max(mp.hi_max, mhi);
      }
      cur = nxt;
    }
  }
#else
  for (int t = 0; t < Tn; ++t) {  // (uniform) the predicted metadata of step t is in LDS before step t + 1 is planned
    StepPlan pl;
    plan_step<P2, BASE, OffT, true>(kp, lds, dir, s0 + 1 + t, pl);
    if (lane == t) { mp.vslo = pl.lo; mp.vshi = pl.hi; }
    if (pl.lo <= pl.hi) {
      mp.lo_min = min(mp.lo_min, pl.lo);
      mp.hi_max = max(mp.hi_max, pl.hi);
    }
  }
#endif
  // every source row's stored extent = the hull of the step that wrote it (its M row's metadata), fetched by
  // lane r in one LDS round trip.  Score 0 is special -- the search's origin: one cell in the begin
  // component's row, nothing stored for the others -- so there the component's own range counts; rows
  // of negative scores do not exist.  (Sources that a chained sweep takes from registers are never looked up.)
  int comp, score;
  multi_source<P2, E1, E2>(pn, s0, lane, comp, score);
  const bool used = lane < NS0 || (lane < NS0 + 16 * NT && ((lane - NS0) & 15) < Tn);
  RowMeta h = ROW_EMPTY;
  if (used && score >= 0)
    h = meta_load(&lds.ring_meta[(dir * NCOMP + (score == 0 ? comp : C_M)) * kp.ring + (score & (kp.ring - 1))]);
  const bool empty = h.lo > h.hi;
  mp.vlo = empty ? K_BIG : ((h.lo - kmin) & ~3);
  mp.vhi = empty ? -K_BIG : ((h.hi - kmin) | 3);
  // (a chained pass leaves out what it never loads: M source 0 beyond the first sweep, source 1 beyond the second)
  bool counted = used;
  if (nh > 1 && lane >= NS0) {
    const int w = (lane - NS0) >> 4, step = (lane - NS0) & 15;
    if ((w == 0 && step >= TMAX) || (w == 1 && step >= 2 * TMAX)) counted = false;
  }
  mp.int_lo = wave_max_i32(counted ? mp.vlo : INT_MIN);
  mp.int_hi = -wave_max_i32(counted ? -mp.vhi : INT_MIN) - 255;
}

// One pass: Tn steps of one direction over all windows of the rows.  Returns the number of cells.
// BASE: the base case's plain WFA (rows indexed by score in the history arena; every step also stores its I/D
// rows -- the backtrace reads them -- and tells whether the end cell has been reached).
// CHAIN: the pass may consist of several sweeps per window (mp.nh > 1); needs TMAX == x and 2 TMAX == o1 + e1.
// DEEPP: a pass of the breakpoint search that keeps everything the overlap search will read (deep_phase): every step's I/D
// rows go to memory as well, and the rows' max antidiagonals are reported per step (maxak_out[t]) -- exactly what the
// step-by-step loop leaves behind, at a pass's price.
template <bool P2, typename OffT, int E1, int E2, bool BASE, bool CHAIN, bool DEEPP = false>
__device__ __forceinline__ int compute_rows_multi(const KParams& kp, Shared& sh, const Lds<OffT>& lds, const SubCtx& cx, rsrc_t rs,
                                                  int dir, int s0, const MultiPlan& mp, Acc& acc, int* maxak_out, unsigned& ext_iters) {
  constexpr bool DEEP = BASE || DEEPP;  // every step's I/D rows go to memory
  static_assert(!(DEEPP && (BASE || CHAIN)), "deep passes of the search are single sweeps over the ring");
  constexpr bool W16 = sizeof(OffT) == 2;  // 16-bit rows: packed arithmetic; 32-bit rows: the same recurrences on 32-bit registers
  static_assert(E1 >= 1 && E1 <= 2 && E2 >= 1 && E2 <= 2, "register-resident I/D depth");
  static_assert(!(BASE && CHAIN), "the base case runs single sweeps");
  static_assert(W16 || !CHAIN || (TMAX32 == TMAX && CHAIN_MAX32 >= 2 && CHAIN_MAX32 <= 3), "chained 32-bit sweeps are TMAX long");
  constexpr int CH = W16 ? CHAIN_MAX : CHAIN_MAX32;  // sweeps a pass may chain
  constexpr int NWAVES = WG / 64;
  constexpr int VEC = 4, ESZ = (int)sizeof(OffT);
  constexpr int TM = W16 ? TMAX : TMAX32;  // steps per sweep
  constexpr int NW = W16 ? 2 : 4;          // 32-bit words per lane vector
  constexpr int NS0 = 2 * E1 + (P2 ? 2 * E2 : 0);
  constexpr int NT = P2 ? 3 : 2;
  typedef RawVec<OffT> V;
  const DevPenalties& pn = kp.pen;
  const int lane = threadIdx.x & 63;
  const int kmin = dir ? cx.kmin[1] : cx.kmin[0];
  const int plen = cx.plen, tlen = cx.tlen;
  const int Tn = mp.Tn, nh = CHAIN ? mp.nh : 1;
  if (mp.lo_min > mp.hi_max) return 0;  // all steps empty
  if (mp.lo_min - kmin < 256 + 1 || mp.hi_max - kmin + 256 + VEC + 2 > cx.wcols) {
    sh.error = ST_CAPACITY;
    return 0;
  }
  const int colLoMin = mp.lo_min - kmin, colHiMax = mp.hi_max - kmin;
  const unsigned nullw = W16 ? ((unsigned)(unsigned short)NULL16) * 0x00010001u : (unsigned)OFF_NULL;
  const int halo = mp.halo;                 // unproductive lanes on either side: their columns go invalid one per step
  const int stride = (64 - 2 * halo) * VEC;  // new columns per window
  const bool productive = lane >= halo && lane < 64 - halo;
  bool lane_oob = false;
  int lane_maxak = 0;       // max antidiagonal over all of the pass's productive cells
  int lane_maxak_t[DEEPP ? TM : 1];  // DEEPP: per step
#pragma unroll
  for (int t = 0; t < (DEEPP ? TM : 1); ++t) lane_maxak_t[t] = 0;
  unsigned reach_mask = 0;  // BASE: bit t = the end cell has been reached at step t (uniform)
  typedef short s2 __attribute__((ext_vector_type(2)));
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  auto as2 = [](unsigned w) { return __builtin_bit_cast(s2, w); };
  auto asu = [](s2 v) { return __builtin_bit_cast(unsigned, v); };
  const s2 one = {1, 1};
  constexpr bool WE = wenc_of<OffT>();  // AWV_WIDE16: stored w = h - max(k, 0)
  const int vcap = WE ? min(plen, tlen) : tlen;  // largest stored value inside the matrix
  const short tl1 = (short)(vcap + 1);
  const s2 tlen1 = {tl1, tl1};
  const us2 null_u = {(unsigned short)NULL16, (unsigned short)NULL16};
  auto canon = [&](unsigned w) {  // v < 0 ? NULL16 : min(v, tlen + 1), per half
    const s2 c = __builtin_elementwise_min(as2(w), tlen1);
    return asu(__builtin_bit_cast(s2, __builtin_elementwise_min(__builtin_bit_cast(us2, c), null_u)));
  };
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  auto st = [&](int soff, int voff, const V& v) {
    if constexpr (W16) {
      u32x2 w2;
      w2[0] = v.w[0];
      w2[1] = v.w[1];
      __builtin_amdgcn_raw_buffer_store_b64(w2, rs, voff, soff, 0);
    } else {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      u32x4 w4;
#pragma unroll
      for (int j = 0; j < 4; ++j) w4[j] = v.w[j];
      __builtin_amdgcn_raw_buffer_store_b128(w4, rs, voff, soff, 0);
    }
  };
  // stored form of an I/D vector: canonical halves with 16-bit rows, the values themselves with 32-bit rows
  auto stored = [&](const V& v) {
    V o;
#pragma unroll
    for (int r = 0; r < NW; ++r) o.w[r] = W16 ? canon(v.w[r]) : v.w[r];
    return o;
  };
  int nwin = 0;
  // (declared out here: with XPREF these registers carry the next window's first-sweep sources across the window loop)
  constexpr bool ALIAS = CHAIN && W16 && (AWV_CHAIN_ALIAS != 0);
  constexpr bool PREF = ALIAS && P2 && (AWV_TAP_PREFETCH != 0);
  constexpr bool XPREF = PREF && (AWV_WINDOW_PREFETCH != 0);
  V Mp1[CHAIN ? TM : 1], Mp2[CHAIN && (ALIAS || CH >= 3) ? TM : 1];
  V tapO[PREF ? TM : 1], tapN[PREF ? TM : 1];
  bool pre_loaded = false;  // (uniform) XPREF: the previous window has issued this window's first-sweep loads
  constexpr bool LCH = CHAIN && !W16 && (AWV_LDS_CHAIN != 0);
  const bool lds_chain = LCH && mp.lds_chain != 0;  // (uniform)
  typedef unsigned int u32x4l __attribute__((ext_vector_type(4)));
  // this wave's slots: [step of the sweep][lane], 16 B each (the region is 16-byte aligned: Lds::seq starts at a 16-byte multiple)
  u32x4l* const chain_slot = reinterpret_cast<u32x4l*>(const_cast<uint32_t*>(lds.seq)) + (size_t)(threadIdx.x >> 6) * (TM * 64) + lane;
  for (int cb = (colLoMin & ~(VEC - 1)) - halo * VEC + (NWAVES > 1 ? (int)(threadIdx.x >> 6) * stride : 0); cb + halo * VEC <= colHiMax;
       cb += stride * NWAVES) {
    ++nwin;
    const bool has_next = XPREF && cb + stride * NWAVES + halo * VEC <= colHiMax;  // (uniform) this wave has another window in this pass
    const int c0 = cb + lane * VEC;
    const int k0 = c0 + kmin;
    const int voff = c0 * ESZ;
    const bool interior = cb >= mp.int_lo && cb <= mp.int_hi;  // wave-uniform
    // lanes that can matter: everything that lies inside some source's stored hull is within `halo` lanes of the final hull
    const bool load_on = c0 + (halo + 2) * VEC > colLoMin && c0 - (halo + 1) * VEC <= colHiMax;
    auto lmask = [&](int r, V& v) {  // edge windows: lane vectors outside the hull of a source's step hold nothing (or stale data)
      const int alo = __builtin_amdgcn_readlane(mp.vlo, r), ahi = __builtin_amdgcn_readlane(mp.vhi, r);
      const bool keep = c0 >= alo && c0 <= ahi;
#pragma unroll
      for (int r = 0; r < NW; ++r) v.w[r] = keep ? v.w[r] : nullw;
    };
    const unsigned long long tm0 = PROF_NOW();
    AWV_PRIO(3);
    // ---- the I/D rows the pass begins with
    V qI1[E1], qD1[E1], qI2[E2], qD2[E2];
#pragma unroll
    for (int j = 0; j < E1; ++j) { qI1[j] = V{}; qD1[j] = V{}; }
#pragma unroll
    for (int j = 0; j < E2; ++j) { qI2[j] = V{}; qD2[j] = V{}; }
    if (load_on) {
#pragma unroll
      for (int j = 0; j < E1; ++j) {
        qI1[j] = buf_load_raw<OffT, 2>(rs, voff, row_off<BASE, OffT>(kp, dir, C_I1, s0 - E1 + 1 + j));
        qD1[j] = buf_load_raw<OffT, 2>(rs, voff, row_off<BASE, OffT>(kp, dir, C_D1, s0 - E1 + 1 + j));
      }
      if (P2) {
#pragma unroll
        for (int j = 0; j < E2; ++j) {
          qI2[j] = buf_load_raw<OffT, 2>(rs, voff, row_off<BASE, OffT>(kp, dir, C_I2, s0 - E2 + 1 + j));
          qD2[j] = buf_load_raw<OffT, 2>(rs, voff, row_off<BASE, OffT>(kp, dir, C_D2, s0 - E2 + 1 + j));
        }
      }
    }
    // the last two sweeps' own M rows (canonical stored form), what the next sweeps read 5 / 10 scores back.
    // 16-bit rows (ALIAS): the same registers ARE the first sweep's sources -- its x- and (o1 + e1)-lag M rows are loaded straight
    // into Mp1 / Mp2, every step reads its sources there and leaves Mp2[t] = Mp1[t], Mp1[t] = its own row behind: the next sweep's
    // 5- and 10-back rows without a select per step, a second copy of the sweep's results or a separate set of load registers
    // (30 registers less than keeping taps, Mp1 / Mp2 and the new rows apart; the (o1 + e1)-lag rows of the second sweep are the
    // first sweep's x-lag rows -- no load for them either).
    if (!(XPREF && pre_loaded)) {
#pragma unroll
      for (int t = 0; t < (CHAIN ? TM : 1); ++t) Mp1[t] = V{};
#pragma unroll
      for (int t = 0; t < (CHAIN && (ALIAS || CH >= 3) ? TM : 1); ++t) Mp2[t] = V{};
    }
    // ALIAS + AWV_TAP_PREFETCH: the one source a chained sweep still loads -- the (o2 + e2)-lag M rows, 25 scores back, written
    // by earlier passes whatever the sweep -- is loaded a whole sweep ahead: sweep h computes from tapO while tapN (sweep h + 1's)
    // is in flight, so only a window's first sweep waits for memory.
#pragma unroll
    for (int t = 0; t < (PREF ? TM : 1); ++t) {
      if (!(XPREF && pre_loaded)) tapO[t] = V{};
      tapN[t] = V{};
    }
    if constexpr (PREF) {
      if (load_on && !(XPREF && pre_loaded)) {
#pragma unroll
        for (int t = 0; t < TM; ++t)
          if (t < min(TM, Tn)) tapO[t] = buf_load_raw<OffT>(rs, voff, row_off<BASE, OffT>(kp, dir, C_M, s0 + 1 + t - pn.o2 - pn.e2));
      }
    }
    int win_maxak = INT_MIN / 2;  // max(2 c_j - j) over this window's steps (far-apart passes; folded into lane_maxak below)
    // per cell, the largest / smallest offset inside the matrix on its diagonal (the same for every step of the pass)
    int hmaxv[VEC], hminv[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      hmaxv[j] = WE ? wenc_max(k0 + j, plen, tlen, vcap) : clamp_from_m1(plen + k0 + j, tlen);
      hminv[j] = WE ? 0 : max(k0 + j, 0);
    }
    // AWV_WIDE16: an insertion into diagonal k adds 1 only for k <= 0, a deletion only for k >= 0
    s2 incI[2], incD[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int ka = k0 + 2 * r;
      incI[r] = WE ? s2{(short)(ka <= 0), (short)(ka + 1 <= 0)} : one;
      incD[r] = s2{(short)(ka >= 0), (short)(ka + 1 >= 0)};
    }
#pragma nounroll
    for (int h = 0; h < nh; ++h) {
      const int sb = s0 + h * TM;   // this sweep covers scores sb + 1 .. sb + TM
      const int tb = h * TM;        // its first step index within the pass
      const int tn = min(TM, Tn - tb);
      AWV_PRIO(3);
      const bool own0 = CHAIN && h >= 1, own1 = CHAIN && (ALIAS || lds_chain ? h >= 1 : (CH >= 3 && h >= 2));  // (uniform) M sources 0 / 1 come from registers (or, 32-bit rows, from LDS)
      // ---- 16-bit rows: all row loads of the sweep, back to back (one memory round trip per sweep).  32-bit rows (a lane
      // vector is four registers; TM x NT of them do not fit): the M sources of a step are loaded one step ahead -- `cur`
      // feeds step t while `nxt` (step t + 1) is in flight; these kernels are bound by HBM bytes, not by the round trips
      constexpr int TAPS = W16 ? TM : 1;
      V tap[TAPS][NT];
#pragma unroll
      for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int w = 0; w < NT; ++w) tap[t][w] = V{};
      auto load_taps = [&](int t, V (&dst)[NT]) {  // M sources of step t (of this sweep), masked to their stored hulls
#pragma unroll
        for (int w = 0; w < NT; ++w) dst[w] = V{};
        if (t < tn) {
          if (load_on) {
            if (!own0) dst[0] = buf_load_raw<OffT>(rs, voff, row_off<BASE, OffT>(kp, dir, C_M, sb + 1 + t - pn.x));
            if (!own1) dst[1] = buf_load_raw<OffT>(rs, voff, row_off<BASE, OffT>(kp, dir, C_M, sb + 1 + t - pn.o1 - pn.e1));
            if (P2) dst[NT - 1] = buf_load_raw<OffT>(rs, voff, row_off<BASE, OffT>(kp, dir, C_M, sb + 1 + t - pn.o2 - pn.e2));
          }
          if (!interior) {
            if (!own0) lmask(NS0 + tb + t, dst[0]);
            if (!own1) lmask(NS0 + 16 + tb + t, dst[1]);
            if (P2) lmask(NS0 + 32 + tb + t, dst[NT - 1]);
          }
        }
      };
      if constexpr (W16) {
        if (load_on) {
#pragma unroll
          for (int t = 0; t < TM; ++t) {
            if (t < tn) {
              if constexpr (ALIAS) {
                if (!own0 && !(XPREF && pre_loaded)) {  // (own0 = own1 here: the first sweep only; with XPREF the previous window may have issued them)
                  Mp1[t] = buf_load_raw<OffT>(rs, voff, row_off<BASE, OffT>(kp, dir, C_M, sb + 1 + t - pn.x));
                  Mp2[t] = buf_load_raw<OffT>(rs, voff, row_off<BASE, OffT>(kp, dir, C_M, sb + 1 + t - pn.o1 - pn.e1));
                }
              } else {
                if (!own0) tap[t][0] = buf_load_raw<OffT>(rs, voff, row_off<BASE, OffT>(kp, dir, C_M, sb + 1 + t - pn.x));
                if (!own1) tap[t][1] = buf_load_raw<OffT>(rs, voff, row_off<BASE, OffT>(kp, dir, C_M, sb + 1 + t - pn.o1 - pn.e1));
              }
              if constexpr (PREF) {  // the NEXT sweep's rows (a pass of several sweeps is made of whole sweeps)
                if (h + 1 < nh) tapN[t] = buf_load_raw<OffT>(rs, voff, row_off<BASE, OffT>(kp, dir, C_M, sb + TM + 1 + t - pn.o2 - pn.e2));
              } else if (P2) tap[t][NT - 1] = buf_load_raw<OffT>(rs, voff, row_off<BASE, OffT>(kp, dir, C_M, sb + 1 + t - pn.o2 - pn.e2));
            }
          }
        }
      }
      AWV_PRIO(0);
      if (!interior) {
        if (h == 0) {
#pragma unroll
          for (int j = 0; j < E1; ++j) { lmask(j, qI1[j]); lmask(E1 + j, qD1[j]); }
          if (P2) {
#pragma unroll
            for (int j = 0; j < E2; ++j) { lmask(2 * E1 + j, qI2[j]); lmask(2 * E1 + E2 + j, qD2[j]); }
          }
        }
        if constexpr (W16) {
#pragma unroll
          for (int t = 0; t < TM; ++t) {
            if (t < tn) {
              if constexpr (ALIAS) {
                if (!own0) lmask(NS0 + tb + t, Mp1[t]);
                if (!own1) lmask(NS0 + 16 + tb + t, Mp2[t]);
              } else {
                if (!own0) lmask(NS0 + tb + t, tap[t][0]);
                if (!own1) lmask(NS0 + 16 + tb + t, tap[t][1]);
              }
              if constexpr (PREF) lmask(NS0 + 32 + tb + t, tapO[t]);
              else if (P2) lmask(NS0 + 32 + tb + t, tap[t][NT - 1]);
            }
          }
        }
      }
      V cur[NT], nxt[NT];
#pragma unroll
      for (int w = 0; w < NT; ++w) { cur[w] = V{}; nxt[w] = V{}; }
      if constexpr (!W16) load_taps(0, cur);
      PROF_DRAIN();
      PROF_ADD_L(STAT_T_CR_LOAD, tm0);
      // ---- the steps
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        if (t < tn) {
          const unsigned long long tm1 = PROF_NOW();
          const int lo_t = __builtin_amdgcn_readlane(mp.vslo, tb + t), hi_t = __builtin_amdgcn_readlane(mp.vshi, tb + t);
          // every lane inside the step's hull computes (the halo lanes' values feed their neighbours and,
          // chained, the next sweeps); only the productive ones store, count and flag
          const bool in_hull = c0 + VEC > lo_t - kmin && c0 <= hi_t - kmin;
          const bool lane_on = productive && in_hull;
          if constexpr (!W16) {
            if (t + 1 < TM) load_taps(t + 1, nxt);
          }
          V cMx = W16 ? tap[W16 ? t : 0][0] : cur[0], cO1 = W16 ? tap[W16 ? t : 0][1] : cur[1];
          const V cO2 = PREF ? tapO[PREF ? t : 0] : W16 ? tap[W16 ? t : 0][NT - 1] : cur[NT - 1];
          if constexpr (ALIAS) {
            cMx = Mp1[t];
            cO1 = Mp2[t];
          } else if (CHAIN) {
            if (own0) cMx = Mp1[t];
            if constexpr (LCH) {
              if (lds_chain) {  // this step's (o1 + e1)-lag source out of the slot, its x-lag source -- the next sweep's (o1 + e1)-lag source -- into it (LDS operations of a wave complete in order)
                if (own1) {
                  const u32x4l w4 = chain_slot[t * 64];
#pragma unroll
                  for (int j = 0; j < NW; ++j) cO1.w[j] = w4[j & 3];
                }
                if (h + 1 < nh) {
                  u32x4l w4;
#pragma unroll
                  for (int j = 0; j < 4; ++j) w4[j] = cMx.w[j & (NW - 1)];
                  chain_slot[t * 64] = w4;
                }
              }
            }
            if (CH >= 3 && own1 && !lds_chain) cO1 = Mp2[CH >= 3 ? t : 0];
          }
          V nI1{}, nD1{}, nI2{}, nD2{};
          int32_t m[VEC];
#if AWV_LEAN_EXT
          int32_t cand[VEC];
          bool okc[VEC];
#endif
          if constexpr (W16) {
            // max first, neighbour shift after: max(O[k-1], I[k-1]) = (max(O, I))[k-1] -- one shift per gap kind instead of two
            auto pkmax = [&](const V& a, const V& b) {
              V o;
              o.w[0] = asu(__builtin_elementwise_max(as2(a.w[0]), as2(b.w[0])));
              o.w[1] = asu(__builtin_elementwise_max(as2(a.w[1]), as2(b.w[1])));
              return o;
            };
            const V sI1 = shift_from_left(pkmax(cO1, qI1[0])), sD1 = shift_from_right(pkmax(cO1, qD1[0]));
            V sI2{}, sD2{};
            if (P2) {
              sI2 = shift_from_left(pkmax(cO2, qI2[0]));
              sD2 = shift_from_right(pkmax(cO2, qD2[0]));
            }
  #pragma unroll
            for (int r = 0; r < 2; ++r) {
              const s2 ins1 = as2(sI1.w[r]) + incI[r];
              const s2 del1 = WE ? as2(sD1.w[r]) + incD[r] : as2(sD1.w[r]);
              s2 ins = ins1, del = del1;
              nI1.w[r] = asu(ins1);
              nD1.w[r] = asu(del1);
              if (P2) {
                const s2 ins2 = as2(sI2.w[r]) + incI[r];
                const s2 del2 = WE ? as2(sD2.w[r]) + incD[r] : as2(sD2.w[r]);
                ins = __builtin_elementwise_max(ins, ins2);
                del = __builtin_elementwise_max(del, del2);
                nI2.w[r] = asu(ins2);
                nD2.w[r] = asu(del2);
              }
              const s2 mm2 = __builtin_elementwise_max(del, __builtin_elementwise_max(as2(cMx.w[r]) + one, ins));
  #pragma unroll
              for (int e = 0; e < 2; ++e) {
                const int mm = (int)mm2[e];
                const int hmax = hmaxv[2 * r + e];
                lane_oob |= lane_on && mm > hmax;
                // (h < k, i.e. a negative pattern position, never occurs in a real wavefront; the halo lanes' stale values
                // -- which are extended like any other now that chained sweeps read them back -- may hold anything)
#if AWV_LEAN_EXT
                cand[2 * r + e] = mm;
                okc[2 * r + e] = in_hull && mm <= hmax && mm >= hminv[2 * r + e];
#else
                m[2 * r + e] = (mm > hmax || mm < hminv[2 * r + e] || !in_hull) ? OFF_NULL : mm;
#endif
              }
            }
          } else {
            const V rO1l = shift_from_left(cO1), rO1r = shift_from_right(cO1);
            const V rI1 = shift_from_left(qI1[0]), rD1 = shift_from_right(qD1[0]);
            V rO2l{}, rO2r{}, rI2{}, rD2{};
            if (P2) {
              rO2l = shift_from_left(cO2);
              rO2r = shift_from_right(cO2);
              rI2 = shift_from_left(qI2[0]);
              rD2 = shift_from_right(qD2[0]);
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              const int32_t ins1 = max((int32_t)rO1l.w[j], (int32_t)rI1.w[j]) + 1;
              const int32_t del1 = max((int32_t)rO1r.w[j], (int32_t)rD1.w[j]);
              int32_t ins = ins1, del = del1;
              nI1.w[j] = (unsigned)ins1;
              nD1.w[j] = (unsigned)del1;
              if (P2) {
                const int32_t ins2 = max((int32_t)rO2l.w[j], (int32_t)rI2.w[j]) + 1;
                const int32_t del2 = max((int32_t)rO2r.w[j], (int32_t)rD2.w[j]);
                ins = max(ins, ins2);
                del = max(del, del2);
                nI2.w[j] = (unsigned)ins2;
                nD2.w[j] = (unsigned)del2;
              }
              const int32_t mm = max(del, max((int32_t)cMx.w[j] + 1, ins));
              const int hmax = hmaxv[j];
              lane_oob |= lane_on && mm > hmax;
#if AWV_LEAN_EXT
              cand[j] = mm;
              okc[j] = in_hull && mm <= hmax && mm >= hminv[j];
#else
              m[j] = (mm > hmax || mm < hminv[j] || !in_hull) ? OFF_NULL : mm;
#endif
            }
          }
          if (DEEP && lane_on) {  // this score's I/D rows (canonical form), whole lane vectors over the step's hull
            st(row_off<BASE, OffT>(kp, dir, C_I1, sb + 1 + t), voff, stored(nI1));
            st(row_off<BASE, OffT>(kp, dir, C_D1, sb + 1 + t), voff, stored(nD1));
            if (P2) {
              st(row_off<BASE, OffT>(kp, dir, C_I2, sb + 1 + t), voff, stored(nI2));
              st(row_off<BASE, OffT>(kp, dir, C_D2, sb + 1 + t), voff, stored(nD2));
            }
          }
          if (BASE && mp.end_comp != C_M) {  // end cell in an indel component: has its offset reached the text end?
            const V& ev = mp.end_comp == C_I1 ? nI1 : mp.end_comp == C_D1 ? nD1 : mp.end_comp == C_I2 ? nI2 : nD2;
            const int je = mp.end_col - c0;  // element of this lane's vector, if it holds the end column
            const bool mine = lane_on && je >= 0 && je < VEC;
            int val;
            if constexpr (W16) {
              const unsigned w = (je & 2) ? ev.w[1] : ev.w[0];
              val = (je & 1) ? ((int)w >> 16) : (((int)w << 16) >> 16);
            } else {
              val = (int)(je == 0 ? ev.w[0] : je == 1 ? ev.w[1] : je == 2 ? ev.w[2] : ev.w[3]);
            }
            if (__any(mine && val >= (WE ? vcap : tlen))) reach_mask |= 1u << (tb + t);  // (the end cell's diagonal is tlen - plen: h = tlen <=> w = min(plen, tlen))
          }
          PROF_DRAIN();
          PROF_ADD_L(STAT_T_CR_ALU, tm1);
          const unsigned long long tm2 = PROF_NOW();
#if AWV_LEAN_EXT
          // (batched probe reads where the window loop has the registers for them: the base case, and the chained far-apart passes
          // when they do not spend those registers on loading a sweep ahead -- which pays more: config 2, same box, 1645 ms batched /
          // 1615 ms a sweep ahead / 1632 ms both (two spills); the deep passes' 15 source vectors and the 32-bit rows' four-word lane
          // vectors leave no room: the compiler's one-read-at-a-time order there)
          extend_cells_lean<OffT, (W16 && !WE && WG == 64 && ((ALIAS && !PREF) || BASE)) ? AWV_PROBE_BATCH : 0>(lds, cx, dir, k0, cand, okc, hmaxv, m, ext_iters);
#else
          extend_cells_n<OffT, 4, true>(lds, cx, dir, k0, m, ext_iters);
#endif
          PROF_DRAIN();
          PROF_ADD_L(STAT_T_CR_EXTEND, tm2);
          const unsigned long long tm3 = PROF_NOW();
          {  // canonical stored form of the M cells: what goes to memory and what the next sweeps read back
            V mv;
            int c[VEC];
            if constexpr (W16) {
              mv.w[0] = pack_canon16(m[0], m[1], vcap + 1, c[0], c[1]);
              mv.w[1] = pack_canon16(m[2], m[3], vcap + 1, c[2], c[3]);
            } else {
#pragma unroll
              for (int j = 0; j < VEC; ++j) {
                mv.w[j] = (unsigned)m[j];
                c[j] = max(m[j], -(1 << 28));  // (a clamp from below for the antidiagonal maximum: 2 c - k stays far below 0 for a NULL, without overflow)
              }
            }
            if (!BASE) {
              // the pass's max antidiagonal, reduced once after the last window.  From the clamped values: a NULL is
              // -16384 there, so 2 c - k stays far below any real antidiagonal without a test per cell; cells outside
              // the step's hull are NULL, and the halo lanes' sums are dropped at the end.
              if constexpr (WE) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                  int& mk = DEEPP ? lane_maxak_t[DEEPP ? t : 0] : lane_maxak;
                  mk = max(mk, 2 * m[j] + abs(k0 + j));  // h + v = 2 w + |k| (a NULL is OFF_NULL in m: far below 0; |k| may exceed 16384, so not from the clamped value)
                }
              } else {
                // h + v = 2 c - (k0 + j): the lane's k0 is the same for every step of the window, so the steps only gather
                // max(2 c_j - j) (one shift-add per cell, two three-way maxima) and k0 comes off once
                const int g = max(max(2 * c[0], 2 * c[1] - 1), max(2 * c[2] - 2, 2 * c[3] - 3));
                if constexpr (DEEPP) lane_maxak_t[t] = max(lane_maxak_t[t], g - k0);
                else win_maxak = max(win_maxak, g);
              }
            }
            if (lane_on) st(row_off<BASE, OffT>(kp, dir, C_M, sb + 1 + t), voff, mv);
            if constexpr (CHAIN && W16) {  // in place: this step has read its entries already (without ALIAS: only what the next sweeps take from registers)
              if (XPREF && has_next && h == nh - 1) {
                // the window's last sweep: nobody reads this step's chain registers again -- they take the NEXT window's first-sweep
                // sources of the same step (its loads are then a whole sweep old when that window starts)
                const int voff_n = voff + stride * NWAVES * ESZ, c0_n = c0 + stride * NWAVES;
                const bool load_on_n = c0_n + (halo + 2) * VEC > colLoMin && c0_n - (halo + 1) * VEC <= colHiMax;
                Mp1[t] = V{};
                Mp2[XPREF ? t : 0] = V{};
                tapO[XPREF ? t : 0] = V{};
                if (load_on_n) {
                  Mp1[t] = buf_load_raw<OffT>(rs, voff_n, row_off<BASE, OffT>(kp, dir, C_M, s0 + 1 + t - pn.x));
                  Mp2[XPREF ? t : 0] = buf_load_raw<OffT>(rs, voff_n, row_off<BASE, OffT>(kp, dir, C_M, s0 + 1 + t - pn.o1 - pn.e1));
                  tapO[XPREF ? t : 0] = buf_load_raw<OffT>(rs, voff_n, row_off<BASE, OffT>(kp, dir, C_M, s0 + 1 + t - pn.o2 - pn.e2));
                }
              } else {
                if constexpr (ALIAS || CH >= 3) Mp2[ALIAS || CH >= 3 ? t : 0] = Mp1[t];
                Mp1[t] = mv;
              }
            }
            if constexpr (CHAIN && !W16) {  // in place: this step has read its entries already (the registers of a second copy are not there)
              if (CH >= 3) Mp2[CH >= 3 ? t : 0] = Mp1[t];
              Mp1[t] = mv;
            }
          }
          if (BASE && mp.end_comp == C_M) {
            const int je = mp.end_col - c0;
            const bool mine = lane_on && je >= 0 && je < VEC;
            const int val = je == 0 ? m[0] : je == 1 ? m[1] : je == 2 ? m[2] : m[3];
            if (__any(mine && val >= (WE ? vcap : tlen))) reach_mask |= 1u << (tb + t);
          }
          PROF_DRAIN();
          PROF_ADD_L(STAT_T_CR_STORE, tm3);
          // the I/D rows of this score enter the register queues (oldest first)
#pragma unroll
          for (int j = 0; j + 1 < E1; ++j) { qI1[j] = qI1[j + 1]; qD1[j] = qD1[j + 1]; }
          qI1[E1 - 1] = nI1;
          qD1[E1 - 1] = nD1;
          if (P2) {
#pragma unroll
            for (int j = 0; j + 1 < E2; ++j) { qI2[j] = qI2[j + 1]; qD2[j] = qD2[j + 1]; }
            qI2[E2 - 1] = nI2;
            qD2[E2 - 1] = nD2;
          }
          if constexpr (!W16) {
#pragma unroll
            for (int w = 0; w < NT; ++w) cur[w] = nxt[w];
          }
        }
      }
      if constexpr (PREF) {
        if (!(XPREF && has_next && h == nh - 1)) {
#pragma unroll
          for (int t = 0; t < TM; ++t) tapO[t] = tapN[t];
        }
      }
    }
    pre_loaded = has_next;
    if constexpr (!BASE && !DEEPP && !WE) lane_maxak = max(lane_maxak, win_maxak - k0);
    // ---- the pass's last e1 / e2 I/D rows (canonical form, whole lane vectors over their step's hull)
    if (!DEEP) {
      // queue entry j was produced by step Tn - E + j (Tn >= E); its lanes are those of that step's hull
#pragma unroll
      for (int j = 0; j < E1; ++j) {
        const int tj = Tn - E1 + j;
        const int lo_j = __builtin_amdgcn_readlane(mp.vslo, tj), hi_j = __builtin_amdgcn_readlane(mp.vshi, tj);
        const bool on = productive && c0 + VEC > lo_j - kmin && c0 <= hi_j - kmin;
        if (on) {
          st(row_off<BASE, OffT>(kp, dir, C_I1, s0 + 1 + tj), voff, stored(qI1[j]));
          st(row_off<BASE, OffT>(kp, dir, C_D1, s0 + 1 + tj), voff, stored(qD1[j]));
        }
      }
      if (P2) {
#pragma unroll
        for (int j = 0; j < E2; ++j) {
          const int tj = Tn - E2 + j;
          const int lo_j = __builtin_amdgcn_readlane(mp.vslo, tj), hi_j = __builtin_amdgcn_readlane(mp.vshi, tj);
          const bool on = productive && c0 + VEC > lo_j - kmin && c0 <= hi_j - kmin;
          if (on) {
            st(row_off<BASE, OffT>(kp, dir, C_I2, s0 + 1 + tj), voff, stored(qI2[j]));
            st(row_off<BASE, OffT>(kp, dir, C_D2, s0 + 1 + tj), voff, stored(qD2[j]));
          }
        }
      }
    }
  }
  if (threadIdx.x == 0) (BASE ? sh.win_base_multi : sh.win_multi) += (unsigned)(nwin * nh);
  int cells = 0;
  const bool woob = __any(lane_oob);
  for (int t = 0; t < Tn; ++t) {
    const int lo_t = __builtin_amdgcn_readlane(mp.vslo, t), hi_t = __builtin_amdgcn_readlane(mp.vshi, t);
    if (lo_t <= hi_t) cells += hi_t - lo_t + 1;
  }
  if (lane == 0 && woob) acc.oob = 1;
  if (!BASE) {
    if constexpr (DEEPP) {
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const int wmax = wave_max_i32(productive ? lane_maxak_t[t] : 0);
        if (lane == 0 && t < Tn) atomicMax(&maxak_out[t], wmax);
      }
    } else {
      const int wmax = wave_max_i32(productive ? lane_maxak : 0);
      if (lane == 0) atomicMax(&maxak_out[0], wmax);
    }
  }
  if (BASE && lane == 0 && reach_mask) atomicOr(&acc.reach, (int)reach_mask);
  return cells;
}

// The far-apart phase as a function of its own (never inlined): inlined into the search loop, the ~110
// live vector registers of a pass met everything else the kernel keeps alive and the compiler spilled
// into the hot loop (even the buffer descriptor, which then has to be re-made uniform lane by lane).
// One call per breakpoint search runs ALL of its multi-step passes -- both directions in lockstep, one
// barrier per pass -- together with WFA2's phase-1 bookkeeping for the scores they cover, and hands the
// search back to the step-by-step loop when (MP_MARGIN) the furthest points come within the safety margin
// of meeting, (MP_DISCARD) a pass saw a value leave the matrix (that pass is not counted: its rows are
// redone step by step) or (MP_MET) the furthest points met inside a pass: the I/D rows the overlap
// search needs were then never written and the caller runs the search again step by step.
// Inputs beyond the few scalars come from LDS (Shared::pctx), results go back through Shared::pres.
typedef __attribute__((address_space(3))) Shared* lds_shared_ptr;
typedef __attribute__((address_space(3))) unsigned char* lds_bytes_ptr;
constexpr int MP_MARGIN = 0, MP_DISCARD = 1, MP_MET = 2, MP_ERROR = 3, MP_DEEP_MET = 4;
template <bool P2, typename OffT, int E1, int E2>
__device__ __attribute__((noinline)) void deep_phase(unsigned sh_addr, unsigned dyn_addr, int s0_v, int fmax_v, int rmax_v, int Tn_v, int pass_v,
                                                     int far_npass_v, unsigned far_cells_lo, unsigned far_cells_hi);
// safety margin = BASE + MUL4/4 * (scope + T) * (recent advance per step), in antidiagonal units.  Tuned on config 2
// (profiles/r02/margin_ab.json): 256 + 2.0 x -> 1926 ms, 26 restarts; 256 + 1.5 x -> 1903 ms, 942 restarts of 979 k searches;
// 128 + 1.5 x -> 1911 ms, 13 k restarts; 64 + 1.25 x -> 2154 ms, 124 k restarts.
// attempts of a breakpoint search: 3 = the margins below, four times the margins, step by step; 2 = round 2's rule (step by step at once)
#ifndef AWV_RESTART_ATTEMPTS
#define AWV_RESTART_ATTEMPTS 3
#endif
#ifndef AWV_MARGIN_BASE
#define AWV_MARGIN_BASE 256
#endif
#ifndef AWV_MARGIN_MUL4
#define AWV_MARGIN_MUL4 6
#endif
constexpr int MARGIN_BASE = AWV_MARGIN_BASE, MARGIN_MUL4 = AWV_MARGIN_MUL4;
template <bool P2, typename OffT, int E1, int E2, bool CHAIN>
__device__ __attribute__((noinline)) void multi_phase(unsigned sh_addr, unsigned dyn_addr, int s0_v, int fmax_v, int rmax_v, int Tn_v, int pass_v, int deep_v) {
  Shared& sh = *(Shared*)__builtin_assume_aligned((Shared*)(lds_shared_ptr)(uintptr_t)uni((int)sh_addr), 8);
  unsigned char* dyn_smem = (unsigned char*)__builtin_assume_aligned((unsigned char*)(lds_bytes_ptr)(uintptr_t)uni((int)dyn_addr), 16);
  const int Tn = uni(Tn_v);
  int sc = uni(s0_v);  // both directions stand at the same score when the phase begins and after every pass
  int fmax = uni(fmax_v), rmax = uni(rmax_v), pass = uni(pass_v);
  const PassCtx& pc = sh.pctx;
  KParams kp{};
  kp.ring = uni(pc.ring);
  kp.wcap = uni(pc.wcap);
  kp.pen.x = uni(pc.x);
  kp.pen.o1 = uni(pc.o1);
  kp.pen.e1 = uni(pc.e1);
  kp.pen.o2 = uni(pc.o2);
  kp.pen.e2 = uni(pc.e2);
  kp.pen.two_piece = P2 ? 1 : 0;
  kp.pen.scope = max(kp.pen.x, max(kp.pen.o1 + kp.pen.e1, P2 ? kp.pen.o2 + kp.pen.e2 : 0)) + 1;
  kp.lds_meta_bytes = uni(pc.lds_meta_bytes);
  auto uni64 = [](unsigned long long v) { return ((unsigned long long)(unsigned)uni((int)(v >> 32)) << 32) | (unsigned)uni((int)v); };
  SubCtx cx;
  cx.plen = uni(pc.plen);
  cx.tlen = uni(pc.tlen);
  cx.kmin[0] = uni(pc.kmin[0]);
  cx.kmin[1] = uni(pc.kmin[1]);
  cx.wcols = uni(pc.wcols);
  cx.seq_mode = uni(pc.seq_mode);
  cx.p_w0 = uni(pc.p_w0);
  cx.t_w0 = uni(pc.t_w0);
  cx.p_bit = uni(pc.p_bit);
  cx.t_bit = uni(pc.t_bit);
  cx.P[0] = (gseq_t)(uintptr_t)uni64(pc.P[0]);
  cx.P[1] = (gseq_t)(uintptr_t)uni64(pc.P[1]);
  cx.T[0] = (gseq_t)(uintptr_t)uni64(pc.T[0]);
  cx.T[1] = (gseq_t)(uintptr_t)uni64(pc.T[1]);
  cx.Pw = nullptr;
  cx.Tw = nullptr;
  cx.pb_abs = cx.tb_abs = 0;
  Lds<OffT> lds;
  typedef typename MetaTraits<OffT>::Stored MetaStored;
  lds.ring_meta = reinterpret_cast<MetaStored*>(dyn_smem);
  lds.bi_A = reinterpret_cast<int*>(lds.ring_meta + 2 * NCOMP * kp.ring);
  lds.bi_oob = lds.bi_A + 2 * kp.ring;
  lds.firstk = lds.bi_oob + 2 * kp.ring;
  lds.seq = reinterpret_cast<uint32_t*>(dyn_smem + kp.lds_meta_bytes);
  lds.meta_log = nullptr;
  const rsrc_t rs = make_rsrc((void*)(uintptr_t)uni64(pc.ring_mem), (size_t)uni64(pc.ring_bytes));
  const int rmask = kp.ring - 1;
  const int max_antidiagonal = cx.plen + cx.tlen - 1;
  const int arun_start = fmax + rmax;
  int arun0 = fmax, arun1 = rmax;  // max antidiagonal over every row computed so far, per direction
  int nsteps = 0, npass = 0, why = MP_MARGIN;
  unsigned long long cells = 0;
  unsigned ext_iters = 0;
  // 32-bit rows: a third chained sweep when the staging region of the packed sequences can hold the chain rows (AWV_LDS_CHAIN)
  const bool lds_chain = CHAIN && sizeof(OffT) == 4 && (AWV_LDS_CHAIN != 0) && cx.seq_mode != 1 &&
                         uni(pc.lds_seq_bytes) >= (WG / 64) * TMAX32 * 64 * 16;
  const int margin_shift = (uni(deep_v) >> 8) & 3;  // a restarted search: the margin times four (find_breakpoint)
  const bool long_reads = (uni(deep_v) & 0x400) != 0;  // a 16-bit search of a launch with 32-bit rows: those reads' margin factor
  const int chain_cap = CHAIN ? min(max(uni(pc.chain_max), 1), sizeof(OffT) == 2 ? CHAIN_MAX : (lds_chain ? 3 : CHAIN_MAX32)) : 1;
  for (;;) {
    // Start keeping every I/D row well before the furthest points can meet: the margin is several times what
    // the two searches advance while `scope` more rows (and one more pass) are computed.  The longest chain
    // whose own length still fits in front of that margin is taken.
    const int grow = nsteps > 0 ? (arun0 + arun1 - arun_start) / nsteps : 0;
    int nh = 0;
    for (int c = chain_cap; c >= 1; --c) {
      const int T = c > 1 ? TMAX * c : Tn;
      // (32-bit rows = long sequences with long exact runs: the searches advance in larger bursts, 2.0 x instead of 1.5 x -- config 4:
      // 9.91 s and 4.9 k restarted searches against 10.31 s and 26 k)
      const int mul4 = (sizeof(OffT) == 2 && !long_reads) ? MARGIN_MUL4 : MARGIN_MUL4 + 2;
      if (arun0 + arun1 < max_antidiagonal - ((MARGIN_BASE + mul4 * (kp.pen.scope + T) * max(grow, 8) / 4) << margin_shift)) { nh = c; break; }
    }
    if (nh == 0) { why = MP_MARGIN; break; }
    const int T = nh > 1 ? TMAX * nh : Tn;
    const int aslot = pass % 3;
    int nc0 = 0, nc1 = 0;
    // (a real loop over the direction: one copy of the pass code; per-direction values are picked by
    // selects, never by indexing a register array with the run-time `dir`)
#pragma nounroll
    for (int dir = 0; dir < 2; ++dir) {
      MultiPlan mp;
      const unsigned long long tpl = PROF_NOW();
      plan_multi<P2, OffT, E1, E2, false>(kp, lds, cx, dir, sc, T, nh, mp);
      mp.lds_chain = lds_chain ? 1 : 0;
      PROF_ADD_L(STAT_T_CR_REDUCE, tpl);  // (diagnostic build: the passes' planning is booked under "reduce")
      Acc& acc = dir ? sh.acc[aslot][1] : sh.acc[aslot][0];
      const int nc = compute_rows_multi<P2, OffT, E1, E2, false, CHAIN>(kp, sh, lds, cx, rs, dir, sc, mp, acc, dir ? sh.chain_maxak[1] : sh.chain_maxak[0], ext_iters);
      if (dir) nc1 = nc; else nc0 = nc;
    }
    __syncthreads();
    if (uni(sh.error)) { why = MP_ERROR; break; }
    if (uni(sh.acc[aslot][0].oob) != 0 || uni(sh.acc[aslot][1].oob) != 0) { why = MP_DISCARD; break; }  // the pass assumed untrimmed rows
    cells += (unsigned long long)(nc0 + nc1);
    // The pass's rows become official.  WFA2's phase-1 order is forward, test, reverse, test per score (A.6), but
    // the running maxima only grow: some test inside the pass holds exactly when the one after its last score
    // does -- so one maximum per direction over the whole pass decides it, and no per-score reduction is
    // needed.  (No per-score entries -- bi_A / bi_oob -- are written for a far-apart pass's scores: their only reader, the
    // overlap search of phase 2, looks at the last `scope` scores of either side, all of which lie at or above `deep_since`
    // (deep_ok), i.e. past every far-apart pass; a ring slot is rewritten by the deep pass or step that next lands on it.)
    const int P0 = uni(sh.chain_maxak[0][0]), P1 = uni(sh.chain_maxak[1][0]);
    arun0 = max(arun0, P0);
    arun1 = max(arun1, P1);
    fmax = max(fmax, P0);
    rmax = max(rmax, P1);
    const bool met = fmax + rmax >= max_antidiagonal;
    __syncthreads();  // everyone has read the pass's maxima
    if (threadIdx.x < 16) sh.chain_maxak[threadIdx.x >> 3][threadIdx.x & 7] = 0;
    if (threadIdx.x == 0) { acc_reset(sh.acc[(pass + 2) % 3][0]); acc_reset(sh.acc[(pass + 2) % 3][1]); }
    __syncthreads();  // ... and they are clear before the next pass adds to them
    ++pass;
    ++npass;
    nsteps += T;
    sc += T;
    if (met) { why = MP_MET; break; }
  }
  __syncthreads();  // every thread is past its last look at the accumulators
  if (threadIdx.x == 0) {
    PhaseResult& pr = sh.pres;
    pr.why = why;
    pr.sc = sc;
    pr.sf = pr.sr = sc;
    pr.last_fwd = 0;
    pr.deep_from = sc;
    pr.fmax = fmax;
    pr.rmax = rmax;
    pr.npass = npass;
    pr.cells = cells;
    pr.deep_cells = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) { acc_reset(sh.acc[i][0]); acc_reset(sh.acc[i][1]); }  // whatever the caller's pass counter says next, its slot is clean
  }
  if (threadIdx.x < 16) sh.chain_maxak[threadIdx.x >> 3][threadIdx.x & 7] = 0;
  atomicAdd(&sh.ext_multi, (unsigned long long)ext_iters);
  __syncthreads();
  // The margin is reached: the rest of phase 1 runs in passes that store every I/D row (deep_phase), called from HERE -- a call
  // site of its own in find_breakpoint_fn cost that function's step-by-step loop register spills (config 5's forced-gap pairs,
  // which live in that loop: +24 %); its result record replaces the one above and carries the far-apart passes' counts along.
  if ((uni(deep_v) & 1) != 0 && why == MP_MARGIN)
    deep_phase<P2, OffT, E1, E2>(sh_addr, dyn_addr, sc, fmax, rmax, Tn, pass, npass, (unsigned)cells, (unsigned)(cells >> 32));
}

// The rest of phase 1 in passes (round 3).  From the safety margin on, every I/D row has to be in memory -- the overlap search
// of phase 2 reads those of the last `scope` scores -- and round 2 went step by step there: 80 % of all step-by-step window-steps
// of config 2 lay in that zone (profiles/r03/zones.json).  deep_phase runs it in single sweeps of T scores that keep the I/D
// queues in registers like any pass AND store every step's I/D rows (compute_rows_multi<DEEPP>), with the rows' max
// antidiagonals per score, and replays WFA2's phase-1 bookkeeping (forward, test, reverse, test -- A.6) over the pass's scores
// afterwards, exactly as the step-by-step loop would have run it.  The pass in which the furthest points can meet ends the
// phase (MP_DEEP_MET): pres.sf / pres.sr are the official scores at that test, pres.last_fwd which side advanced last, and
// pres.sc the score both directions are COMPUTED through -- the rows beyond the official scores are the ones phase 2 asks for
// next, already there (nothing reads a row above the official scores, so computing them early changes no result; the ring
// holds scope + T + 1 rows).  A pass in which a value leaves the matrix is discarded as in multi_phase (MP_DISCARD).
template <bool P2, typename OffT, int E1, int E2>
__device__ __attribute__((noinline)) void deep_phase(unsigned sh_addr, unsigned dyn_addr, int s0_v, int fmax_v, int rmax_v, int Tn_v, int pass_v,
                                                     int far_npass_v, unsigned far_cells_lo, unsigned far_cells_hi) {
  Shared& sh = *(Shared*)__builtin_assume_aligned((Shared*)(lds_shared_ptr)(uintptr_t)uni((int)sh_addr), 8);
  unsigned char* dyn_smem = (unsigned char*)__builtin_assume_aligned((unsigned char*)(lds_bytes_ptr)(uintptr_t)uni((int)dyn_addr), 16);
  const int Tn = uni(Tn_v);
  int sc = uni(s0_v);  // both directions stand at the same score when the phase begins and after every pass
  int fmax = uni(fmax_v), rmax = uni(rmax_v), pass = uni(pass_v);
  const PassCtx& pc = sh.pctx;
  KParams kp{};
  kp.ring = uni(pc.ring);
  kp.wcap = uni(pc.wcap);
  kp.pen.x = uni(pc.x);
  kp.pen.o1 = uni(pc.o1);
  kp.pen.e1 = uni(pc.e1);
  kp.pen.o2 = uni(pc.o2);
  kp.pen.e2 = uni(pc.e2);
  kp.pen.two_piece = P2 ? 1 : 0;
  kp.pen.scope = max(kp.pen.x, max(kp.pen.o1 + kp.pen.e1, P2 ? kp.pen.o2 + kp.pen.e2 : 0)) + 1;
  kp.lds_meta_bytes = uni(pc.lds_meta_bytes);
  auto uni64 = [](unsigned long long v) { return ((unsigned long long)(unsigned)uni((int)(v >> 32)) << 32) | (unsigned)uni((int)v); };
  SubCtx cx;
  cx.plen = uni(pc.plen);
  cx.tlen = uni(pc.tlen);
  cx.kmin[0] = uni(pc.kmin[0]);
  cx.kmin[1] = uni(pc.kmin[1]);
  cx.wcols = uni(pc.wcols);
  cx.seq_mode = uni(pc.seq_mode);
  cx.p_w0 = uni(pc.p_w0);
  cx.t_w0 = uni(pc.t_w0);
  cx.p_bit = uni(pc.p_bit);
  cx.t_bit = uni(pc.t_bit);
  cx.P[0] = (gseq_t)(uintptr_t)uni64(pc.P[0]);
  cx.P[1] = (gseq_t)(uintptr_t)uni64(pc.P[1]);
  cx.T[0] = (gseq_t)(uintptr_t)uni64(pc.T[0]);
  cx.T[1] = (gseq_t)(uintptr_t)uni64(pc.T[1]);
  cx.Pw = nullptr;
  cx.Tw = nullptr;
  cx.pb_abs = cx.tb_abs = 0;
  Lds<OffT> lds;
  typedef typename MetaTraits<OffT>::Stored MetaStored;
  lds.ring_meta = reinterpret_cast<MetaStored*>(dyn_smem);
  lds.bi_A = reinterpret_cast<int*>(lds.ring_meta + 2 * NCOMP * kp.ring);
  lds.bi_oob = lds.bi_A + 2 * kp.ring;
  lds.firstk = lds.bi_oob + 2 * kp.ring;
  lds.seq = reinterpret_cast<uint32_t*>(dyn_smem + kp.lds_meta_bytes);
  lds.meta_log = nullptr;
  const rsrc_t rs = make_rsrc((void*)(uintptr_t)uni64(pc.ring_mem), (size_t)uni64(pc.ring_bytes));
  const int rmask = kp.ring - 1;
  const int max_antidiagonal = cx.plen + cx.tlen - 1;
  int sf = sc, sr = sc, last_fwd = 0;  // official scores and the side that advanced last (WFA2's phase-1 loop, A.6)
  const int deep_from = sc, far_npass = uni(far_npass_v);
  const unsigned long long far_cells = ((unsigned long long)(unsigned)uni((int)far_cells_hi) << 32) | (unsigned)uni((int)far_cells_lo);
  int npass = 0, why = MP_DISCARD;
  unsigned long long cells = 0;
  unsigned ext_iters = 0;
  for (;;) {
    const int aslot = pass % 3;
    int nc0 = 0, nc1 = 0;
#pragma nounroll
    for (int dir = 0; dir < 2; ++dir) {  // (one copy of the pass code, as in multi_phase)
      MultiPlan mp;
      const unsigned long long tpl = PROF_NOW();
      plan_multi<P2, OffT, E1, E2, false>(kp, lds, cx, dir, sc, Tn, 1, mp);
      PROF_ADD_L(STAT_T_CR_REDUCE, tpl);
      Acc& acc = dir ? sh.acc[aslot][1] : sh.acc[aslot][0];
      const int nc = compute_rows_multi<P2, OffT, E1, E2, false, false, true>(kp, sh, lds, cx, rs, dir, sc, mp, acc, dir ? sh.chain_maxak[1] : sh.chain_maxak[0], ext_iters);
      if (dir) nc1 = nc; else nc0 = nc;
    }
    __syncthreads();
    if (uni(sh.error)) { why = MP_ERROR; break; }
    if (uni(sh.acc[aslot][0].oob) != 0 || uni(sh.acc[aslot][1].oob) != 0) { why = MP_DISCARD; break; }  // the pass assumed untrimmed rows: redone step by step
    cells += (unsigned long long)(nc0 + nc1);
    // The pass's rows become available: exact per-score maxima (what finalize_row records after a step), no out-of-bounds value.
    // Then WFA2's phase-1 order over the pass's scores: forward, test, reverse, test -- the first test that holds ends phase 1
    // right there (every thread computes the same from the same LDS values).
    bool met = false;
    for (int t = 0; t < Tn; ++t) {
      const int slot = (sc + 1 + t) & rmask;
      const int A0 = uni(sh.chain_maxak[0][t]), A1 = uni(sh.chain_maxak[1][t]);
      lds.bi_A[slot] = A0;
      lds.bi_A[kp.ring + slot] = A1;
      lds.bi_oob[slot] = 0;
      lds.bi_oob[kp.ring + slot] = 0;
      if (!met) {
        ++sf;
        fmax = max(fmax, A0);
        last_fwd = 1;
        if (fmax + rmax >= max_antidiagonal) met = true;
        else {
          ++sr;
          rmax = max(rmax, A1);
          last_fwd = 0;
          if (fmax + rmax >= max_antidiagonal) met = true;
        }
      }
    }
    __syncthreads();  // everyone has read the pass's maxima
    if (threadIdx.x < 16) sh.chain_maxak[threadIdx.x >> 3][threadIdx.x & 7] = 0;
    if (threadIdx.x == 0) { acc_reset(sh.acc[(pass + 2) % 3][0]); acc_reset(sh.acc[(pass + 2) % 3][1]); }
    __syncthreads();  // ... and they are clear before the next pass adds to them
    ++pass;
    ++npass;
    sc += Tn;
    if (met) { why = MP_DEEP_MET; break; }
  }
  __syncthreads();  // every thread is past its last look at the accumulators
  if (threadIdx.x == 0) {
    PhaseResult& pr = sh.pres;
    pr.why = why;
    pr.sc = sc;
    pr.sf = sf;
    pr.sr = sr;
    pr.last_fwd = last_fwd;
    pr.deep_from = deep_from;
    pr.fmax = fmax;
    pr.rmax = rmax;
    pr.npass = far_npass + npass;
    pr.cells = far_cells + cells;
    pr.deep_cells = cells;
#pragma unroll
    for (int i = 0; i < 3; ++i) { acc_reset(sh.acc[i][0]); acc_reset(sh.acc[i][1]); }
  }
  if (threadIdx.x < 16) sh.chain_maxak[threadIdx.x >> 3][threadIdx.x & 7] = 0;
  atomicAdd(&sh.ext_multi, (unsigned long long)ext_iters);
  __syncthreads();
}

// The base case's counterpart of multi_phase: plain WFA over the history arena in multi-step passes
// (one memory round trip per T scores instead of one per score -- the base case's rows are a window
// or two wide, so it is bound by exactly that latency), until the end cell is reached (MP_MET: pres.sc
// is the final score), a pass sees a value leave the matrix (MP_DISCARD: continue step by step after
// pres.sc) or the history's score capacity comes near (MP_MARGIN: likewise).
template <bool P2, typename OffT, int E1, int E2>
__device__ __attribute__((noinline)) void base_phase(unsigned sh_addr, unsigned dyn_addr, int s0_v, int Tn_v, int pass_v) {
  Shared& sh = *(Shared*)__builtin_assume_aligned((Shared*)(lds_shared_ptr)(uintptr_t)uni((int)sh_addr), 8);
  unsigned char* dyn_smem = (unsigned char*)__builtin_assume_aligned((unsigned char*)(lds_bytes_ptr)(uintptr_t)uni((int)dyn_addr), 16);
  const int Tn = uni(Tn_v);
  int sc = uni(s0_v), pass = uni(pass_v);
  const PassCtx& pc = sh.pctx;
  KParams kp{};
  kp.ring = uni(pc.ring);
  kp.wb_cap = uni(pc.wb_cap);
  kp.sb_cap = uni(pc.sb_cap);
  kp.pen.x = uni(pc.x);
  kp.pen.o1 = uni(pc.o1);
  kp.pen.e1 = uni(pc.e1);
  kp.pen.o2 = uni(pc.o2);
  kp.pen.e2 = uni(pc.e2);
  kp.pen.two_piece = P2 ? 1 : 0;
  kp.pen.scope = max(kp.pen.x, max(kp.pen.o1 + kp.pen.e1, P2 ? kp.pen.o2 + kp.pen.e2 : 0)) + 1;
  kp.lds_meta_bytes = uni(pc.lds_meta_bytes);
  auto uni64 = [](unsigned long long v) { return ((unsigned long long)(unsigned)uni((int)(v >> 32)) << 32) | (unsigned)uni((int)v); };
  SubCtx cx;
  cx.plen = uni(pc.plen);
  cx.tlen = uni(pc.tlen);
  cx.kmin[0] = uni(pc.kmin[0]);
  cx.kmin[1] = cx.kmin[0];
  cx.wcols = uni(pc.wcols);
  cx.seq_mode = uni(pc.seq_mode);
  cx.p_w0 = uni(pc.p_w0);
  cx.t_w0 = uni(pc.t_w0);
  cx.p_bit = uni(pc.p_bit);
  cx.t_bit = uni(pc.t_bit);
  cx.P[0] = cx.P[1] = (gseq_t)(uintptr_t)uni64(pc.P[0]);
  cx.T[0] = cx.T[1] = (gseq_t)(uintptr_t)uni64(pc.T[0]);
  cx.Pw = nullptr;
  cx.Tw = nullptr;
  cx.pb_abs = cx.tb_abs = 0;
  Lds<OffT> lds;
  typedef typename MetaTraits<OffT>::Stored MetaStored;
  lds.ring_meta = reinterpret_cast<MetaStored*>(dyn_smem);
  lds.bi_A = reinterpret_cast<int*>(lds.ring_meta + 2 * NCOMP * kp.ring);
  lds.bi_oob = lds.bi_A + 2 * kp.ring;
  lds.firstk = lds.bi_oob + 2 * kp.ring;
  lds.seq = reinterpret_cast<uint32_t*>(dyn_smem + kp.lds_meta_bytes);
  lds.meta_log = (RowMeta*)(__attribute__((address_space(1))) RowMeta*)(uintptr_t)uni64(pc.meta_log);
  const rsrc_t rs = make_rsrc((void*)(uintptr_t)uni64(pc.ring_mem), (size_t)uni64(pc.ring_bytes));
  const int end_comp = uni(pc.end_comp);
  const int end_col = (cx.tlen - cx.plen) - cx.kmin[0];
  int npass = 0, why = MP_MARGIN;
  unsigned long long cells = 0;
  unsigned ext_iters = 0;
  for (;;) {
    if (sc + Tn > kp.sb_cap) { why = MP_MARGIN; break; }  // the last scores before the capacity bound: step by step (they report CAPACITY themselves)
    const int aslot = pass % 3;
    MultiPlan mp;
    plan_multi<P2, OffT, E1, E2, true>(kp, lds, cx, 0, sc, Tn, 1, mp);
    mp.end_comp = end_comp;
    mp.end_col = end_col;
    const int nc = compute_rows_multi<P2, OffT, E1, E2, true, false>(kp, sh, lds, cx, rs, 0, sc, mp, sh.acc[aslot][0], nullptr, ext_iters);
    __syncthreads();
    if (uni(sh.error)) { why = MP_ERROR; break; }
    if (uni(sh.acc[aslot][0].oob) != 0) { why = MP_DISCARD; break; }  // the pass assumed untrimmed rows
    const int reach = uni(sh.acc[aslot][0].reach);
    if (threadIdx.x == 0) acc_reset(sh.acc[(pass + 2) % 3][0]);
    ++pass;
    ++npass;
    if (reach) {  // the end cell was reached at the first such step: that score is the penalty
      const int t_end = __builtin_ctz((unsigned)reach);
      for (int t = 0; t <= t_end; ++t) {
        const int lo_t = __builtin_amdgcn_readlane(mp.vslo, t), hi_t = __builtin_amdgcn_readlane(mp.vshi, t);
        if (lo_t <= hi_t) cells += (unsigned long long)(hi_t - lo_t + 1);
      }
      sc += t_end + 1;
      why = MP_MET;
      break;
    }
    cells += (unsigned long long)nc;
    sc += Tn;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    PhaseResult& pr = sh.pres;
    pr.why = why;
    pr.sc = sc;
    pr.fmax = pr.rmax = 0;
    pr.npass = npass;
    pr.cells = cells;
#pragma unroll
    for (int i = 0; i < 3; ++i) { acc_reset(sh.acc[i][0]); acc_reset(sh.acc[i][1]); }
  }
  atomicAdd(&sh.ext_multi, (unsigned long long)ext_iters);
  __syncthreads();
}

// Rare path (some value went out of bounds): find the trimmed hulls (A.3 "trim ends") by
// rescanning the rows just written.  Must run after the barrier that follows compute_row.
template <bool P2, bool BASE, typename OffT>
__device__ __forceinline__ void trim_pass(const KParams& kp, const SubCtx& cx, void* mem, int dir, int score,
                                          int lo, int hi, Acc& acc) {
  // The trimmed range of a component = [first in-bounds cell, last in-bounds cell] of the row just written.
  // Out-of-bounds cells sit at the row's ends (where the wavefront runs off the matrix), so the row is
  // searched from both ends inwards, 64 columns at a time, and each end stops at its first hit: a few
  // chunks per step instead of the whole row -- which for the rows of a forced gap (tens of thousands of
  // columns, trimmed at every score) used to cost more than computing them.  Every wave searches its own
  // chunks (chunk i belongs to wave i mod nwaves); the minimum / maximum over the waves is the row's.
  constexpr int NW = WG / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kmin = cx.kmin[dir];
  const int plen = cx.plen, tlen = cx.tlen;
  const int colLo = lo - kmin, colHi = hi - kmin;
  const int first = colLo & ~63;
  const int n = (colHi - first) / 64 + 1;  // chunks of the row
  int wlo[NCOMP], whi[NCOMP], ilo[NCOMP];
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) { wlo[c] = INT_MAX; whi[c] = INT_MIN; ilo[c] = -1; }
  auto chunk_mask = [&](int c, int i) -> uint64_t {
    const int cb = first + 64 * i;
    const int col = cb + lane;
    const int k = col + kmin;
    bool inb = false;
    if (col >= colLo && col <= colHi) {
      const int32_t v = off_load1<OffT>(row_ptr<BASE, OffT>(kp, mem, dir, c, score) + col, k);
      inb = (uint32_t)v <= (uint32_t)tlen && (uint32_t)(v - k) <= (uint32_t)plen;
    }
    return __ballot(inb);
  };
  const unsigned all = P2 ? 0x1Fu : ((1u << C_M) | (1u << C_I1) | (1u << C_D1));
  unsigned need = all;
  for (int i = wave; i < n && need; i += NW) {  // from the low end
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) {
      if (!((need >> c) & 1u)) continue;
      const uint64_t mask = chunk_mask(c, i);
      if (mask) {
        wlo[c] = first + 64 * i + (int)__builtin_ctzll(mask);
        ilo[c] = i;
        need &= ~(1u << c);
      }
    }
  }
  unsigned needh = all & ~need;  // (a component without a hit among this wave's chunks has none to find from the other end either)
  const int ilast = n - 1 - (((n - 1 - wave) % NW + NW) % NW);  // this wave's last chunk (< wave: it has none)
  for (int i = ilast; i >= 0 && needh; i -= NW) {  // from the high end; ends at the chunk of the low-end hit at the latest
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) {
      if (!((needh >> c) & 1u)) continue;
      const uint64_t mask = chunk_mask(c, i);
      if (mask) {
        whi[c] = first + 64 * i + 63 - (int)__builtin_clzll(mask);
        needh &= ~(1u << c);
      }
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) {
      if (wlo[c] != INT_MAX) {
        atomicMin(&acc.hull_lo[c], wlo[c]);
        atomicMax(&acc.hull_hi[c], whi[c]);
      }
    }
  }
}

// trim_pass as a function of its own (never inlined into the step-by-step loop, whose register allocation is
// full: inlined, the two-ended search cost config 2 ten per cent through spills in that loop).  Geometry
// comes from Shared::pctx (written by find_breakpoint / base_align for the running sub-problem).
template <bool P2, bool BASE, typename OffT>
__device__ __attribute__((noinline)) void trim_pass_fn(unsigned sh_addr, int dir_v, int score_v, int lo_v, int hi_v, int aslot_v) {
  Shared& sh = *(Shared*)__builtin_assume_aligned((Shared*)(__attribute__((address_space(3))) Shared*)(uintptr_t)uni((int)sh_addr), 8);
  const int dir = uni(dir_v), score = uni(score_v), lo = uni(lo_v), hi = uni(hi_v), aslot = uni(aslot_v);
  const PassCtx& pc = sh.pctx;
  KParams kp{};
  kp.ring = uni(pc.ring);
  kp.wcap = uni(pc.wcap);
  kp.wb_cap = uni(pc.wb_cap);
  SubCtx cx;
  cx.plen = uni(pc.plen);
  cx.tlen = uni(pc.tlen);
  cx.kmin[0] = uni(pc.kmin[0]);
  cx.kmin[1] = uni(pc.kmin[1]);
  void* mem = (void*)(__attribute__((address_space(1))) char*)(uintptr_t)(((unsigned long long)(unsigned)uni((int)(pc.ring_mem >> 32)) << 32) | (unsigned)uni((int)pc.ring_mem));
  Acc& acc = dir ? sh.acc[aslot][1] : sh.acc[aslot][0];
  if (dir) trim_pass<P2, BASE, OffT>(kp, cx, mem, 1, score, lo, hi, acc);
  else trim_pass<P2, BASE, OffT>(kp, cx, mem, 0, score, lo, hi, acc);
}

// after the barrier: every thread writes the same values (benign same-value stores).  The predicted
// metadata is already in place; only a trimmed row (rare) is patched.
template <bool BASE, typename OffT>
__device__ __forceinline__ void finalize_row(const KParams& kp, const Lds<OffT>& lds, const SubCtx& cx, int dir, int score,
                                             const Acc& acc, bool trimmed) {
  if (trimmed) {
    const int kmin = cx.kmin[dir];
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) {
      RowMeta m = ROW_EMPTY;
      const int l = uni(acc.hull_lo[c]), h = uni(acc.hull_hi[c]);
      if (l != INT_MAX) { m.lo = l + kmin; m.hi = h + kmin; }
      meta_store(&lds.ring_meta[(dir * NCOMP + c) * kp.ring + (score & (kp.ring - 1))], m);
      if (BASE && (threadIdx.x & 63) == 0) lds.meta_log[score * NCOMP + c] = m;
    }
  }
  if (!BASE) {
    lds.bi_A[(dir) * kp.ring + (score & (kp.ring - 1))] = uni(acc.maxak);
    lds.bi_oob[(dir) * kp.ring + (score & (kp.ring - 1))] = uni(acc.oob);
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
__device__ __forceinline__ void emit_run(Emit& em, uint8_t op, int len) {
  uint8_t* p = em.cig + em.n;
  for (int i = cold_tid(); i < len; i += WG) p[i] = op;
  em.n += len;
  em.cnt[0] += op == 'M' ? len : 0;
  em.cnt[1] += op == 'X' ? len : 0;
  em.cnt[2] += op == 'I' ? len : 0;
  em.cnt[3] += op == 'D' ? len : 0;
}

// ---------------------------------------------------------------------------------------------
// Base case: plain WFA with full history + backtrace (A.5), wavefront_bialign_base
// ---------------------------------------------------------------------------------------------
template <typename OffT>
__device__ __forceinline__ int bt_fetch(const KParams& kp, const RowMeta* base_meta, const OffT* hist, int kmin,
                                        int max_score, int comp, int score, int k, int add, int type) {
  // the row's metadata and the value are fetched together -- one memory round trip per backtrace step instead of two in series
  // (a score or diagonal outside what exists reads a valid dummy position and is rejected afterwards)
  const bool sok = score >= 0 && score <= max_score;
  const int sr = sok ? score : 0;
  const RowMeta m = base_meta[sr * NCOMP + comp];
  int32_t v = (int32_t)hist[((size_t)sr * NCOMP + comp) * (size_t)kp.wb_cap + (size_t)min(max(k - kmin, 0), kp.wb_cap - 1)];
  if (!sok || k < m.lo || k > m.hi || v < 0) return -1;
  if (wenc_of<OffT>()) v += max(k, 0);  // stored form -> text offset
  return ((v + add) << 4) | type;
}

enum { BT_I1_OPEN = 1, BT_I1_EXT = 2, BT_I2_OPEN = 3, BT_I2_EXT = 4, BT_D1_OPEN = 5, BT_D1_EXT = 6, BT_D2_OPEN = 7, BT_D2_EXT = 8, BT_M = 9 };

template <bool P2, typename OffT>
__device__ __forceinline__ int base_align(const KParams& kp, Shared& sh, const Lds<OffT>& lds, SubCtx cx, void* hist_mem, rsrc_t hist_rs, uint32_t* events,
                          int cb, int ce, Emit& em, int& penalty_out, unsigned long long* lstats) {
  const DevPenalties& pn = kp.pen;
  const int tid = cold_tid(), lane = tid & 63;
  const int plen = cx.plen, tlen = cx.tlen;
  OffT* hist = (OffT*)hist_mem;
  const RowMeta* base_meta = lds.meta_log;  // HBM log, written by plan_step / finalize_row
  const int kspan_lo = min(plen, kp.sb_cap), kspan_hi = min(tlen, kp.sb_cap);
  cx.kmin[0] = -kspan_lo - 4 - COL_PAD;
  cx.wcols = kspan_lo + kspan_hi + 9 + 2 * COL_PAD;
  if (cx.wcols > kp.wb_cap) return ST_CAPACITY;
  const int kmin = cx.kmin[0];
  // score 0
  for (int c = tid; c < NCOMP; c += WG) {
    const RowMeta m0 = (c == cb) ? RowMeta{0, 0} : ROW_EMPTY;
    meta_store(&lds.ring_meta[c * kp.ring + 0], m0);
    lds.meta_log[c] = m0;
  }
  if (tid == 0) {
    unsigned it = 0;
    int v0 = 0;
    if (cb == C_M) v0 = cx.seq_mode == 2 ? extend_lcp_gw<0>(cx, 0, 0, it) : extend_lcp(cx.P[0], cx.T[0], 0, 0, plen, tlen, it);
    for (int j = 0; j < 4; ++j) hist[(size_t)cb * kp.wb_cap + (((0 - kmin) & ~3) + j)] = (OffT)(sizeof(OffT) == 2 ? NULL16 : OFF_NULL);
    hist[(size_t)cb * kp.wb_cap + (0 - kmin)] = (OffT)v0;  // the score-0 row: one cell in a whole lane vector
    acc_reset(sh.acc[0][0]);
    acc_reset(sh.acc[1][0]);
    acc_reset(sh.acc[2][0]);
  }
  constexpr bool MULTI_BUILD = !DIRSPLIT;
  const unsigned sh_addr = (unsigned)(uintptr_t)&sh, dyn_addr = (unsigned)(uintptr_t)lds.ring_meta;  // LDS addresses
  if (tid == 0) {  // what base_phase / trim_pass_fn read back (uniform; the barrier below publishes it)
    PassCtx& pc = sh.pctx;
    pc.ring_mem = (unsigned long long)(uintptr_t)hist_mem;
    pc.ring_bytes = (unsigned long long)kp.hist_slot_stride;
    pc.meta_log = (unsigned long long)(uintptr_t)lds.meta_log;
    pc.P[0] = pc.P[1] = (unsigned long long)(uintptr_t)cx.P[0];
    pc.T[0] = pc.T[1] = (unsigned long long)(uintptr_t)cx.T[0];
    pc.ring = kp.ring;
    pc.wcap = kp.wcap;
    pc.wb_cap = kp.wb_cap;
    pc.sb_cap = kp.sb_cap;
    pc.end_comp = ce;
    pc.x = pn.x; pc.o1 = pn.o1; pc.e1 = pn.e1; pc.o2 = pn.o2; pc.e2 = pn.e2;
    pc.lds_meta_bytes = kp.lds_meta_bytes;
    pc.plen = plen; pc.tlen = tlen;
    pc.kmin[0] = pc.kmin[1] = cx.kmin[0];
    pc.wcols = cx.wcols;
    pc.seq_mode = cx.seq_mode; pc.p_w0 = cx.p_w0; pc.t_w0 = cx.t_w0; pc.p_bit = cx.p_bit; pc.t_bit = cx.t_bit;
    sh.ext_multi = 0;
  }
  __syncthreads();
  const int k_end = tlen - plen;
  int score = 0;
  unsigned ext_iters = 0;
  unsigned long long cells = 0;
  int pass = 0;
  bool dirty = false;  // some row of this sub-problem was trimmed: later steps mask element by element
  bool multi_open = MULTI_BUILD && kp.multi_T > 0;  // multi-step passes (base_phase) until one is discarded or the capacity bound is near
  const unsigned long long tb0 = PROF_NOW();
  for (;;) {
    // termination (wavefront_termination_end2end): end component reaches (plen, tlen)
    const RowMeta me = get_meta<OffT>(kp, lds, 0, ce, score);
    if (k_end >= me.lo && k_end <= me.hi) {
      const int32_t v = uni(off_load1<OffT>(hist + ((size_t)score * NCOMP + ce) * (size_t)kp.wb_cap + (k_end - kmin), k_end));
      if (v >= tlen) break;
    }
    if (MULTI_BUILD && multi_open && !dirty) {
      if constexpr (MULTI_BUILD) {
        if (P2 || pn.e1 == 1) base_phase<P2, OffT, P2 ? 2 : 1, 1>(sh_addr, dyn_addr, score, kp.multi_T, pass);
        else base_phase<P2, OffT, 2, 1>(sh_addr, dyn_addr, score, kp.multi_T, pass);
      }
      const int why = uni(sh.pres.why);
      if (why == MP_ERROR) return uni(sh.error);
      score = uni(sh.pres.sc);
      pass += uni(sh.pres.npass);
      cells += ((unsigned long long)(unsigned)uni((int)(sh.pres.cells >> 32)) << 32) | (unsigned)uni((int)sh.pres.cells);
      multi_open = false;  // (discarded pass or capacity margin: the rest goes step by step)
      __syncthreads();     // (sh.pres may be rewritten only after everyone has read it)
      if (why == MP_MET) break;  // `score` is the first score at which the end cell was reached
      continue;
    }
    ++score;
    if (score > kp.sb_cap) return ST_CAPACITY;
    Acc& acc = sh.acc[pass % 3][0];
    StepPlan pl;
    plan_step<P2, true, OffT>(kp, lds, 0, score, pl);
    cells += compute_row<P2, true, OffT>(kp, sh, lds, cx, hist_rs, 0, score, pl, dirty, acc, ext_iters);
    __syncthreads();
    if (uni(sh.error)) return uni(sh.error);
    const bool trim = uni(acc.oob) != 0;
    dirty = dirty || trim;
    if (trim) {
      trim_pass_fn<P2, true, OffT>(sh_addr, 0, score, pl.lo, pl.hi, pass % 3);
      __syncthreads();
    }
    finalize_row<true, OffT>(kp, lds, cx, 0, score, acc, trim);
    if (tid == 0) acc_reset(sh.acc[(pass + 2) % 3][0]);
    ++pass;
  }
  penalty_out = score;
  PROF_ADD(STAT_T_BASE_STEPS, tb0);
  const unsigned long long tb1 = PROF_NOW();
  if (tid == 0) {
    lstats[STAT_CELLS] += cells;
    lstats[STAT_BASE] += 1;
  }
  atomicAdd(&lstats[STAT_EXTEND], (unsigned long long)ext_iters);
  if (MULTI_BUILD && tid == 0) lstats[STAT_EXTEND] += sh.ext_multi;
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
      // max over lanes 0..15 on the DPP network (row_shr 1, 2, 4, 8 leave it in lane 15; lanes 9.. hold -1)
      cand = max(cand, __builtin_amdgcn_update_dpp(-1, cand, 0x111, 0xf, 0xf, false));
      cand = max(cand, __builtin_amdgcn_update_dpp(-1, cand, 0x112, 0xf, 0xf, false));
      cand = max(cand, __builtin_amdgcn_update_dpp(-1, cand, 0x114, 0xf, 0xf, false));
      cand = max(cand, __builtin_amdgcn_update_dpp(-1, cand, 0x118, 0xf, 0xf, false));
      const int max_all = __builtin_amdgcn_readlane(cand, 15);
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
  PROF_ADD(STAT_T_BACKTRACE, tb1);
  const unsigned long long tb2 = PROF_NOW();
  if (uni(sh.error)) return uni(sh.error);
  // ---- emission: events were produced end -> start
  const int nev = uni(sh.nev);
  for (int e = nev - 1; e >= 0; --e) {
    const uint32_t ev = (uint32_t)uni((int)events[e]);
    const uint32_t opi = ev & 3u;
    emit_run(em, opi == 0 ? 'M' : opi == 1 ? 'X' : opi == 2 ? 'I' : 'D', (int)(ev >> 2));
  }
  __syncthreads();  // events / sh.nev are reused by the next base case
  PROF_ADD(STAT_T_EMIT, tb2);
  return ST_OK;
}

// ---------------------------------------------------------------------------------------------
// BiWFA breakpoint search (A.6)
// ---------------------------------------------------------------------------------------------
constexpr int BP_OK = 0, BP_END_REACHED = 100, BP_RESTART = 101;

template <bool P2, typename OffT>
__device__ __forceinline__ void bialign_overlap(const KParams& kp, Shared& sh, const Lds<OffT>& lds, const SubCtx& cx, void* ring_mem, rsrc_t ring_rs, int d0, int s0,
                                int s1, bool fwd, Breakpoint& bp, unsigned long long* lstats) {
  constexpr int VEC = OffTraits<OffT>::VEC;
  const DevPenalties& pn = kp.pen;
  const int d1 = 1 - d0;
  const int tid = cold_tid(), lane = tid & 63, wave = tid >> 6;
  const int rmask = kp.ring - 1;
  const int plen = cx.plen, tlen = cx.tlen, L = plen + tlen, D = tlen - plen;
  const int slot0 = s0 & rmask;
  const int A0 = uni(lds.bi_A[(d0) * kp.ring + (slot0)]), oob0 = uni(lds.bi_oob[(d0) * kp.ring + (slot0)]);
  const int kmin0 = cx.kmin[d0], kmin1 = cx.kmin[d1];
  const int Cm = D - kmin0 - kmin1;  // col1 = Cm - col0, Cm == 63 (mod 64)
  // exact pre-filter: an overlap needs antidiag0 + antidiag1 >= plen + tlen on some diagonal;
  // every in-bounds cell of any component at score s is <= the (extended) M cell, so the rows'
  // max M antidiagonals bound it unless an out-of-bounds value was seen (oob).
  // Evaluated for all candidate scores at once: lane i looks at score s1 - i (64 per block).
  const int gmax = P2 ? max(pn.o1, pn.o2) : pn.o1;
  auto group_mask = [&](int ib) -> uint64_t {
    const int i = ib + lane, si = s1 - i;
    bool ok = false;
    if (i < pn.scope && si >= 0) {
      const int slot1 = si & rmask;
      ok = oob0 || lds.bi_oob[d1 * kp.ring + slot1] || (A0 + lds.bi_A[d1 * kp.ring + slot1] >= L);
    }
    return __ballot(ok);
  };
  bool any = false;
  for (int ib = 0; ib < pn.scope && !any; ib += 64) {
    uint64_t gm = group_mask(ib);
    while (gm) {
      const int i = ib + (int)__builtin_ctzll(gm);
      gm &= gm - 1;
      if (s0 + (s1 - i) - gmax < bp.score) { any = true; break; }
    }
  }
  if (!any) return;
  for (int i = tid; i < pn.scope * NCOMP; i += WG) lds.firstk[i] = INT_MAX;
  __syncthreads();
  // stage 1: parallel scan of every candidate wavefront pair that passes the row filter (superset of
  // what the sequential search visits: the best score only decreases within a call), 256 columns
  // (one chunk) at a time: chunk c of side 0 faces the mirrored chunk of side 1.
  constexpr int ESZ = (int)sizeof(OffT);
  // All components of one score pair are scanned together: per candidate chunk the rows of every
  // wanted component are loaded back to back (one memory round trip), then compared in turn.
  // Per component the first hit in ascending k wins, as in the sequential search.
  auto scan_all = [&](int i, int si, unsigned want) {
    int ca[NCOMP], cbn[NCOMP], so0[NCOMP], so1[NCOMP];
    unsigned live = 0;
    int chlo = INT_MAX, chhi = -1;
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) {
      ca[c] = 1; cbn[c] = 0; so0[c] = so1[c] = 0;
      if (!P2 && (c == C_I2 || c == C_D2)) continue;
      if (!((want >> c) & 1u)) continue;
      const RowMeta r0 = uni(meta_load(&lds.ring_meta[(d0 * NCOMP + c) * kp.ring + slot0]));
      const RowMeta r1 = uni(meta_load(&lds.ring_meta[(d1 * NCOMP + c) * kp.ring + (si & rmask)]));
      if (row_empty(r0) || row_empty(r1)) continue;
      const int a = max(r0.lo, D - r1.hi), b = min(r0.hi, D - r1.lo);
      if (a > b) continue;
      ca[c] = a - kmin0;
      cbn[c] = b - kmin0;
      so0[c] = row_off<false, OffT>(kp, d0, c, s0);
      so1[c] = row_off<false, OffT>(kp, d1, c, si);
      live |= 1u << c;
      chlo = min(chlo, ca[c] >> 8);
      chhi = max(chhi, cbn[c] >> 8);
    }
    if (!live) return;
    // M-row gate: on every diagonal an in-bounds I/D offset is <= the M offset of the same score, so a
    // chunk whose two M rows nowhere reach h0 + h1 >= tlen holds no overlap of any component.  Two
    // loads decide that before the other components' rows are touched.  (Not with out-of-bounds
    // values around: their M cells are stored as NULL while the I/D cells keep the value.)
    const bool m_gate = !oob0 && !uni(lds.bi_oob[d1 * kp.ring + (si & rmask)]);
    const int so0M = row_off<false, OffT>(kp, d0, C_M, s0), so1M = row_off<false, OffT>(kp, d1, C_M, si);
    // One wave per pair: the gate loads of up to four chunks are in flight together (a candidate's
    // range is a few chunks and nearly all of them fail the gate), so a candidate costs one memory
    // round trip instead of one per chunk.
    const bool pregated = m_gate && (WG == 64 || DIRSPLIT) && chhi - chlo < 64;
    uint64_t gate_pass = ~0ull;  // bit (ch - chlo): the chunk passed the gate
    if (pregated) {
      gate_pass = 0;
      for (int cg = chlo; cg <= chhi; cg += 4) {
        RawVec<OffT> g0[4], g1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          g0[u] = RawVec<OffT>{};
          g1[u] = RawVec<OffT>{};
          if (cg + u <= chhi) {
            const int c0 = ((cg + u) << 8) + lane * VEC;
            g0[u] = buf_load_raw<OffT>(ring_rs, c0 * ESZ, so0M);
            g1[u] = buf_load_raw<OffT>(ring_rs, (Cm - c0 - (VEC - 1)) * ESZ, so1M);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (cg + u > chhi) continue;
          int32_t a0[VEC], a1[VEC];
          unpack_raw<OffT>(g0[u], a0);
          unpack_raw<OffT>(g1[u], a1);
          {
            const int c0 = ((cg + u) << 8) + lane * VEC;
            decode4<OffT>(a0, c0 + kmin0);
            decode4<OffT>(a1, Cm - c0 - (VEC - 1) + kmin1);
          }
          bool reach = false;
#pragma unroll
          for (int j = 0; j < VEC; ++j) reach = reach || (a0[j] >= 0 && a1[VEC - 1 - j] >= 0 && a0[j] + a1[VEC - 1 - j] >= tlen);
          if (__ballot(reach) != 0) gate_pass |= 1ull << (cg + u - chlo);
        }
      }
      if (gate_pass == 0) return;
    }
    int nth = 0;  // waves take the candidate chunks round-robin, each in ascending order
    for (int ch = chlo; ch <= chhi && live; ++ch) {
      if (pregated && !((gate_pass >> (ch - chlo)) & 1ull)) continue;
      unsigned here = 0;
#pragma unroll
      for (int c = 0; c < NCOMP; ++c)
        if (((live >> c) & 1u) && ch >= (ca[c] >> 8) && ch <= (cbn[c] >> 8)) here |= 1u << c;
      if (!here) continue;
      if ((nth++ % (WG / 64)) != wave) continue;
      if (tid == 0) lstats[STAT_OVERLAP] += (unsigned long long)__builtin_popcount(here);
      const int cbase = ch << 8;
      const int c0 = cbase + lane * VEC;
      if (m_gate && !pregated) {
        int32_t g0[VEC], g1[VEC];
        unpack_raw<OffT>(buf_load_raw<OffT>(ring_rs, c0 * ESZ, so0M), g0);
        unpack_raw<OffT>(buf_load_raw<OffT>(ring_rs, (Cm - c0 - (VEC - 1)) * ESZ, so1M), g1);
        decode4<OffT>(g0, c0 + kmin0);
        decode4<OffT>(g1, Cm - c0 - (VEC - 1) + kmin1);
        bool reach = false;
#pragma unroll
        for (int j = 0; j < VEC; ++j) reach = reach || (g0[j] >= 0 && g1[VEC - 1 - j] >= 0 && g0[j] + g1[VEC - 1 - j] >= tlen);
        if (__ballot(reach) == 0) continue;
      }
      RawVec<OffT> q0[NCOMP], q1[NCOMP];
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) {
        if (!P2 && (c == C_I2 || c == C_D2)) continue;
        q0[c] = buf_load_raw<OffT>(ring_rs, c0 * ESZ, so0[c]);
        q1[c] = buf_load_raw<OffT>(ring_rs, (Cm - c0 - (VEC - 1)) * ESZ, so1[c]);  // mirrored: v1[VEC-1-j] pairs with v0[j]
      }
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) {
        if (!P2 && (c == C_I2 || c == C_D2)) continue;
        if (!((here >> c) & 1u)) continue;
        int32_t v0[VEC], v1[VEC];
        unpack_raw<OffT>(q0[c], v0);
        unpack_raw<OffT>(q1[c], v1);
        decode4<OffT>(v0, c0 + kmin0);
        decode4<OffT>(v1, Cm - c0 - (VEC - 1) + kmin1);
        int first = VEC;
#pragma unroll
        for (int j = VEC - 1; j >= 0; --j) {
          const int col0 = c0 + j;
          const int k0 = col0 + kmin0, k1 = D - k0;
          const int32_t h0 = v0[j] < 0 ? OFF_NULL : v0[j], h1 = v1[VEC - 1 - j] < 0 ? OFF_NULL : v1[VEC - 1 - j];
          bool cond = col0 >= ca[c] && col0 <= cbn[c] && (h0 + h1 >= tlen);
          if (c != C_M) {  // indel2indel skips out-of-bounds forward coordinates
            const int kf = fwd ? k0 : k1, hf = fwd ? h0 : h1;
            cond = cond && !((hf - kf) > plen || hf > tlen);
          }
          if (cond) first = j;
        }
        const uint64_t mask = __ballot(first < VEC);
        if (mask) {
          const int src = (int)__builtin_ctzll(mask);
          const int fj = __builtin_amdgcn_readlane(first, src);
          if (lane == 0) atomicMin(&lds.firstk[i * NCOMP + c], cbase + src * VEC + fj + kmin0);
          live &= ~(1u << c);
        }
      }
    }
  };
  for (int ib = 0; ib < pn.scope; ib += 64) {
    uint64_t gm = group_mask(ib);
    while (gm) {
      const int i = ib + (int)__builtin_ctzll(gm);
      gm &= gm - 1;
      const int si = s1 - i;
      const bool w2 = P2 && s0 + si - pn.o2 < bp.score, w1 = s0 + si - pn.o1 < bp.score, w0 = s0 + si < bp.score;
      if (!(w2 || w1 || w0)) continue;
      scan_all(i, si, (w2 ? (1u << C_D2) | (1u << C_I2) : 0u) | (w1 ? (1u << C_D1) | (1u << C_I1) : 0u) | (w0 ? 1u << C_M : 0u));
    }
  }
  __syncthreads();
  // stage 2: replay in WFA2's order (per i: D2, I2, D1, I1, M; first k ascending)
  auto apply = [&](int c, int i, int si, int gap_open) {
    const int k0 = uni(lds.firstk[i * NCOMP + c]);
    if (k0 == INT_MAX) return;
    if (s0 + si - gap_open >= bp.score) return;
    const int k1 = D - k0;
    const int32_t h0 = uni(off_load1<OffT>(row_ptr<false, OffT>(kp, ring_mem, d0, c, s0) + (k0 - kmin0), k0));
    const int32_t h1 = uni(off_load1<OffT>(row_ptr<false, OffT>(kp, ring_mem, d1, c, si) + (k1 - kmin1), k1));
    if (fwd) { bp.sf = s0; bp.sr = si; bp.kf = k0; bp.kr = k1; bp.off_f = h0; bp.off_r = h1; }
    else { bp.sf = si; bp.sr = s0; bp.kf = k1; bp.kr = k0; bp.off_f = h1; bp.off_r = h0; }
    bp.score = s0 + si - gap_open;
    bp.comp = c;
  };
  for (int ib = 0; ib < pn.scope; ib += 64) {
    uint64_t gm = group_mask(ib);
    while (gm) {
      const int i = ib + (int)__builtin_ctzll(gm);
      gm &= gm - 1;
      const int si = s1 - i;
      if (P2 && s0 + si - pn.o2 < bp.score) { apply(C_D2, i, si, pn.o2); apply(C_I2, i, si, pn.o2); }
      if (s0 + si - pn.o1 < bp.score) { apply(C_D1, i, si, pn.o1); apply(C_I1, i, si, pn.o1); }
      if (s0 + si >= bp.score) continue;
      apply(C_M, i, si, 0);
    }
  }
  __syncthreads();  // firstk is rewritten by the next call
}

template <bool P2, typename OffT>
__device__ __forceinline__ int find_breakpoint(const KParams& kp, Shared& sh, const Lds<OffT>& lds, SubCtx cx, void* ring_mem, rsrc_t ring_rs, int cb, int ce,
                               int score_remaining, int known_score, int attempt, Breakpoint& bp, unsigned long long* lstats) {
  const DevPenalties& pn = kp.pen;
  const int tid = cold_tid();
  const int plen = cx.plen, tlen = cx.tlen;
  const int rmask = kp.ring - 1;
  // column spaces: forward kmin, reverse kmin mirrored on chunk boundaries (C == 63 mod 64)
  {
    long long bound = (long long)score_remaining + 2LL * pn.scope + 16;
    // the diagonal range is also clipped to what a row can hold; a wavefront that outgrows it is
    // detected in compute_row (ST_CAPACITY) and the pair is re-run by the host with wider rows
    const long long half = (kp.wcap - 2 * COL_PAD - 9 - 256) / 2;
    bound = min(bound, half);
    const int blo = (int)min((long long)plen, bound), bhi = (int)min((long long)tlen, bound);
    const int need = -blo - 4 - COL_PAD;
    cx.kmin[0] = need;
    const int D = tlen - plen;
    const int c0 = D - need - need;
    const int adj = ((255 - c0) % 256 + 256) % 256;  // mirror on whole 256-column chunks
    cx.kmin[1] = need - adj;
    cx.wcols = blo + bhi + 9 + adj + 2 * COL_PAD;
    if (cx.wcols > kp.wcap) return ST_CAPACITY;
  }
  const unsigned sh_addr = (unsigned)(uintptr_t)&sh, dyn_addr = (unsigned)(uintptr_t)lds.ring_meta;  // LDS addresses (low half of the flat ones)
  if (tid == 0) {  // what multi_phase / trim_pass_fn read back (uniform; the barrier below publishes it)
    PassCtx& pc = sh.pctx;
    pc.ring_mem = (unsigned long long)(uintptr_t)ring_mem;
    pc.ring_bytes = (unsigned long long)kp.ring_slot_stride;
    pc.P[0] = (unsigned long long)(uintptr_t)cx.P[0];
    pc.P[1] = (unsigned long long)(uintptr_t)cx.P[1];
    pc.T[0] = (unsigned long long)(uintptr_t)cx.T[0];
    pc.T[1] = (unsigned long long)(uintptr_t)cx.T[1];
    pc.ring = kp.ring;
    pc.wcap = kp.wcap;
    pc.x = pn.x; pc.o1 = pn.o1; pc.e1 = pn.e1; pc.o2 = pn.o2; pc.e2 = pn.e2;
    pc.lds_meta_bytes = kp.lds_meta_bytes;
    pc.chain_max = kp.chain_max;
    pc.plen = plen; pc.tlen = tlen;
    pc.kmin[0] = cx.kmin[0]; pc.kmin[1] = cx.kmin[1];
    pc.wcols = cx.wcols;
    pc.seq_mode = cx.seq_mode; pc.p_w0 = cx.p_w0; pc.t_w0 = cx.t_w0; pc.p_bit = cx.p_bit; pc.t_bit = cx.t_bit;
    sh.ext_multi = 0;
  }
  // score-0 wavefronts (wavefront_unialign_init by begin component)
  for (int i = tid; i < 2 * NCOMP; i += WG) {
    const int dir = i / NCOMP, c = i % NCOMP;
    const int begin = dir == 0 ? cb : ce;
    meta_store(&lds.ring_meta[(dir * NCOMP + c) * kp.ring + 0], (c == begin) ? RowMeta{0, 0} : ROW_EMPTY);
  }
  if (tid == 0 || tid == (WG > 64 ? 64 : 1)) {
    const int dir = tid ? 1 : 0;
    const int begin = dir == 0 ? cb : ce;
    unsigned it = 0;
    int v0 = 0;
    if (begin == C_M)
      v0 = cx.seq_mode == 2 ? (dir ? extend_lcp_gw<1>(cx, 0, 0, it) : extend_lcp_gw<0>(cx, 0, 0, it))
                            : extend_lcp(dir ? cx.P[1] : cx.P[0], dir ? cx.T[1] : cx.T[0], 0, 0, plen, tlen, it);
    {  // the score-0 row: one cell, stored as a whole lane vector like every other row
      const int col = 0 - (dir ? cx.kmin[1] : cx.kmin[0]);
      OffT* row = row_ptr<false, OffT>(kp, ring_mem, dir, begin, 0);
      for (int j = 0; j < 4; ++j) row[(col & ~3) + j] = (OffT)(sizeof(OffT) == 2 ? NULL16 : OFF_NULL);
      row[col] = (OffT)v0;
    }
    sh.ext0[dir] = v0;
    lds.bi_A[(dir) * kp.ring + (0)] = (begin == C_M) ? 2 * v0 : 0;
    lds.bi_oob[(dir) * kp.ring + (0)] = 0;
    acc_reset(sh.acc[0][dir]);
    acc_reset(sh.acc[1][dir]);
    acc_reset(sh.acc[2][dir]);
  }
  __syncthreads();
  if (cb == C_M && ce == C_M && plen == tlen && (uni(sh.ext0[0]) >= tlen || uni(sh.ext0[1]) >= tlen)) return BP_END_REACHED;
  const int max_antidiagonal = plen + tlen - 1;
  int sc[2] = {0, 0};    // official scores (forward, reverse)
  int comp[2] = {0, 0};  // computed scores (may run one ahead: speculative row)
  int fmax = uni(lds.bi_A[(0) * kp.ring + (0)]), rmax = uni(lds.bi_A[(1) * kp.ring + (0)]);
  bp.score = INT_MAX;
  unsigned ext_iters = 0;
  unsigned long long cells = 0, multi_cells = 0, deep_cells = 0;
  int pass = 0;
  const long long max_steps = ((long long)pn.o1 + pn.o2 + 2LL * (pn.e1 + pn.e2) + pn.x) * ((long long)plen + tlen + 4) + 1024;
  long long steps = 0;
  int rc = BP_OK;
  const int gap_opening = P2 ? max(pn.o1, pn.o2) : pn.o1;
  bool last_fwd = false;
  // A sub-problem handed down by a parent's breakpoint has a known optimal score: its share of the
  // parent's optimal alignment (the forward / reverse score at the breakpoint) minus the gap open of
  // the breakpoint's component -- in the child that boundary gap is pre-paid at its begin or end, while
  // both of the parent's searches had paid for entering it.  A cheaper way through the child would make
  // the parent's alignment cheaper too.  WFA2 keeps searching until no better overlap is possible but
  // only ever replaces the breakpoint by a strictly better one, so once that score is reached the rest
  // of the search cannot change the result and is skipped.
  const bool known_optimum = known_score != INT_MAX;
  bool dirty[2] = {false, false};  // per direction: some row was trimmed, later steps mask element by element
  // Multi-step passes (compute_rows_multi) while the searches are far apart; step by step -- every I/D
  // row kept, as the overlap search needs them -- from a safe margin before they can meet.
  constexpr bool MULTI_BUILD = !DIRSPLIT;
  // attempt 0: the margins of multi_phase; 1 (the furthest points met inside a far-apart pass): once more with four times the
  // margin; 2 (again): step by step throughout
  const bool force_single = (attempt & 0xff) >= AWV_RESTART_ATTEMPTS - 1;
  // (bits 8, 9 = log2 of the margin's multiplier, bit 10 = the margin factor of long reads although the rows are 16-bit)
  const int deep_arg = kp.deep_passes | ((attempt & 0xff) == 1 && !force_single ? 0x200 : 0) | ((attempt & 0x100) ? 0x400 : 0);
  const int multi_T = (MULTI_BUILD && !force_single && plen + tlen > AWV_MIN_PASS_LEN) ? kp.multi_T : 0;  // 0: step by step throughout
  bool deep_on = multi_T == 0;     // every step stores its I/D rows
  int deep_since[2] = {deep_on ? 0 : INT_MAX, deep_on ? 0 : INT_MAX};  // first score from which all I/D rows are in HBM
  // the overlap search of (d0, s0) against d1 reads the I/D rows of s0 and of scores s1 - scope + 1 .. s1
  auto deep_ok = [&](int d0, int s0, int d1, int s1) {
    const int lo_need = max(s1 - (pn.scope - 1), 1);
    return (s0 == 0 || deep_since[d0] <= s0) && (s1 == 0 || deep_since[d1] <= lo_need);
  };
  int phase = 1;
  // One loop for both phases (A.6).  Each iteration first makes sure the next forward and the next
  // reverse wavefront exist (one fused pass, one barrier), then runs WFA2's bookkeeping for it.
  for (;;) {
    if (phase == 1 && fmax + rmax >= max_antidiagonal) phase = 2;
    {
      Acc* a = sh.acc[pass % 3];
      int plo[2] = {1, 1}, phi[2] = {0, 0};
      bool need[2];
      const unsigned long long tp0 = PROF_NOW();
      if (MULTI_BUILD && !deep_on && (dirty[0] || dirty[1])) {
        deep_on = true;
        deep_since[0] = comp[0] + 1;
        deep_since[1] = comp[1] + 1;
      }
      if (MULTI_BUILD && !deep_on && phase == 1 && comp[0] == sc[0] && comp[1] == sc[1] && sc[0] == sc[1] &&
          (AWV_EARLY_PASSES || sc[0] + 1 - (pn.scope - 1) >= 1)) {
        // ---- the far-apart phase: all multi-step passes of this search in one call (multi_phase)
        if constexpr (MULTI_BUILD) {
          // (chained sweeps only where the scores allow them: 2-piece with x = TMAX and o1 + e1 = 2 TMAX, the default set)
          if (P2) multi_phase<P2, OffT, 2, 1, P2>(sh_addr, dyn_addr, sc[0], fmax, rmax, multi_T, pass, deep_arg);
          else if (pn.e1 == 1) multi_phase<P2, OffT, 1, 1, false>(sh_addr, dyn_addr, sc[0], fmax, rmax, multi_T, pass, deep_arg);
          else multi_phase<P2, OffT, 2, 1, false>(sh_addr, dyn_addr, sc[0], fmax, rmax, multi_T, pass, deep_arg);
        }
        PROF_ADD(STAT_T_BI_COMPUTE, tp0);
        const int why = uni(sh.pres.why);
        if (why == MP_ERROR) { rc = uni(sh.error); break; }
        if (why == MP_MET) { rc = BP_RESTART; break; }
        // the far-apart passes and (from the margin on) the passes that store every I/D row, in one call: both directions are
        // COMPUTED through pres.sc; the official scores may stand below it when phase 1 ended inside the last pass (the rows
        // beyond them are the ones phase 2 asks for next)
        steps += (long long)(uni(sh.pres.sc) - sc[0]);
        comp[0] = comp[1] = uni(sh.pres.sc);
        sc[0] = uni(sh.pres.sf);
        sc[1] = uni(sh.pres.sr);
        last_fwd = uni(sh.pres.last_fwd) != 0;
        fmax = uni(sh.pres.fmax);
        rmax = uni(sh.pres.rmax);
        pass += uni(sh.pres.npass);
        {
          const unsigned long long c = ((unsigned long long)(unsigned)uni((int)(sh.pres.cells >> 32)) << 32) | (unsigned)uni((int)sh.pres.cells);
          const unsigned long long dc = ((unsigned long long)(unsigned)uni((int)(sh.pres.deep_cells >> 32)) << 32) | (unsigned)uni((int)sh.pres.deep_cells);
          cells += c;
          multi_cells += c;
          deep_cells += dc;
        }
        deep_on = true;  // from here on every row of every component goes to HBM
        deep_since[0] = deep_since[1] = uni(sh.pres.deep_from) + 1;
        __syncthreads();  // (sh.pres may be rewritten by the next search only after everyone has read it)
        continue;
      } else {
#pragma unroll
      for (int dir = 0; dir < 2; ++dir) {  // unrolled: every per-direction array keeps constant indices
        need[dir] = comp[dir] == sc[dir];
        if (need[dir] && (!DIRSPLIT || uni((int)(threadIdx.x >> 6)) == dir)) {
          StepPlan pl;
          plan_step<P2, false, OffT>(kp, lds, dir, sc[dir] + 1, pl);
          cells += compute_row<P2, false, OffT>(kp, sh, lds, cx, ring_rs, dir, sc[dir] + 1, pl, dirty[dir], a[dir], ext_iters);
          plo[dir] = pl.lo;
          phi[dir] = pl.hi;
        }
      }
#ifdef AWV_DIAG
      // diagnostic build: step-by-step window-steps by zone, reported through awv_stats.prof[9..13]
      // ([9] sub-problems without passes, [10] before the passes can start, [11] margin zone of phase 1, [12] phase 2, [13] trimmed rows)
      if (tid == 0) {
        const int zone = (dirty[0] || dirty[1]) ? 4 : multi_T == 0 ? 0 : phase == 2 ? 3 : (sc[0] + 1 - (pn.scope - 1) >= 1) ? 2 : 1;
        unsigned nw = 0;
        if (need[0] && plo[0] <= phi[0]) nw += (unsigned)((phi[0] - plo[0] + 4) / 248 + 1);
        if (need[1] && plo[1] <= phi[1]) nw += (unsigned)((phi[1] - plo[1] + 4) / 248 + 1);
        lstats[STAT_T_CR_LOAD + zone] += nw;
      }
#endif
      if (need[0] || need[1]) {
        PROF_ADD(STAT_T_BI_COMPUTE, tp0);
        PROF_INC(STAT_N_PASSES);
        const unsigned long long tp1 = PROF_NOW();
        __syncthreads();
        PROF_ADD(STAT_T_BI_BARRIER, tp1);
        const unsigned long long tp2 = PROF_NOW();
        if (uni(sh.error)) { rc = uni(sh.error); break; }
        const bool trim0 = need[0] && uni(a[0].oob) != 0, trim1 = need[1] && uni(a[1].oob) != 0;
        dirty[0] = dirty[0] || trim0;
        dirty[1] = dirty[1] || trim1;
        if (trim0 || trim1) {
          if (DIRSPLIT) {  // the other wave planned that row: its hull is the M row's predicted metadata
            if (trim0) { const RowMeta mm = get_meta(kp, lds, 0, C_M, sc[0] + 1); plo[0] = mm.lo; phi[0] = mm.hi; }
            if (trim1) { const RowMeta mm = get_meta(kp, lds, 1, C_M, sc[1] + 1); plo[1] = mm.lo; phi[1] = mm.hi; }
          }
          if (trim0) trim_pass_fn<P2, false, OffT>(sh_addr, 0, sc[0] + 1, plo[0], phi[0], pass % 3);
          if (trim1) trim_pass_fn<P2, false, OffT>(sh_addr, 1, sc[1] + 1, plo[1], phi[1], pass % 3);
          __syncthreads();
        }
        if (need[0]) { finalize_row<false, OffT>(kp, lds, cx, 0, sc[0] + 1, a[0], trim0); comp[0] = sc[0] + 1; }
        if (need[1]) { finalize_row<false, OffT>(kp, lds, cx, 1, sc[1] + 1, a[1], trim1); comp[1] = sc[1] + 1; }
        if (tid == 0) { acc_reset(sh.acc[(pass + 2) % 3][0]); acc_reset(sh.acc[(pass + 2) % 3][1]); }
        ++pass;
        PROF_ADD(STAT_T_BI_FINALIZE, tp2);
      }
      }
    }
    if (phase == 1) {
      // phase 1: until the furthest points can collide
      ++sc[0];
      fmax = max(fmax, uni(lds.bi_A[0 * kp.ring + (sc[0] & rmask)]));
      last_fwd = true;
      if (fmax + rmax >= max_antidiagonal) { phase = 2; continue; }
      ++sc[1];
      rmax = max(rmax, uni(lds.bi_A[1 * kp.ring + (sc[1] & rmask)]));
      last_fwd = false;
    } else {
      // phase 2: until no better overlap is possible
      if (last_fwd) {
        const int min_sr = (sc[1] > pn.scope - 1) ? sc[1] - (pn.scope - 1) : 0;
        if (sc[0] + min_sr - gap_opening >= bp.score) break;
        if (!deep_ok(0, sc[0], 1, sc[1])) { rc = BP_RESTART; break; }
        const unsigned long long to0 = PROF_NOW();
        bialign_overlap<P2, OffT>(kp, sh, lds, cx, ring_mem, ring_rs, 0, sc[0], sc[1], true, bp, lstats);
        PROF_ADD(STAT_T_OVERLAP, to0);
        if (known_optimum && bp.score == known_score) break;
        ++sc[1];
      }
      const int min_sf = (sc[0] > pn.scope - 1) ? sc[0] - (pn.scope - 1) : 0;
      if (min_sf + sc[1] - gap_opening >= bp.score) break;
      if (!deep_ok(1, sc[1], 0, sc[0])) { rc = BP_RESTART; break; }
      const unsigned long long to1 = PROF_NOW();
      bialign_overlap<P2, OffT>(kp, sh, lds, cx, ring_mem, ring_rs, 1, sc[1], sc[0], false, bp, lstats);
      PROF_ADD(STAT_T_OVERLAP, to1);
      if (known_optimum && bp.score == known_score) break;
      ++sc[0];
      last_fwd = true;
    }
    if (++steps > max_steps) { rc = ST_MAX_STEPS; break; }
  }
  if (DIRSPLIT) {
    if ((tid & 63) == 0) atomicAdd(&lstats[STAT_CELLS], cells);
    if (tid == 0) lstats[STAT_BREAKPOINTS] += 1;
  } else if (tid == 0) {
    lstats[STAT_CELLS] += cells;
    lstats[STAT_MULTI_CELLS] += multi_cells;
    lstats[STAT_DEEP_CELLS] += deep_cells;
    lstats[STAT_BREAKPOINTS] += 1;
  }
  atomicAdd(&lstats[STAT_EXTEND], (unsigned long long)ext_iters);
  if (!DIRSPLIT && tid == 0) lstats[STAT_EXTEND] += sh.ext_multi;
  __syncthreads();  // LDS metadata is rewritten by the next sub-problem
  if (rc == BP_OK && bp.score == INT_MAX) rc = ST_INTERNAL;
  return rc;
}

// The breakpoint search as a function of its own (never inlined): the DFS loop, the base case and the CIGAR
// emission keep their state out of its register file, and edits on either side stop perturbing the
// other's allocation (the step-by-step loop inside is full: an inlined change elsewhere in the kernel
// once cost config 2 ten per cent through spills there).  Inputs come from Shared::pctx, filled by the
// kernel for the running sub-problem; the breakpoint goes back through Shared::bp_out.
template <bool P2, typename OffT>
__device__ __attribute__((noinline)) int find_breakpoint_fn(unsigned sh_addr, unsigned dyn_addr, unsigned lstats_addr, int cb_v, int ce_v,
                                                            int score_remaining_v, int known_v, int attempt_v) {
  Shared& sh = *(Shared*)__builtin_assume_aligned((Shared*)(__attribute__((address_space(3))) Shared*)(uintptr_t)uni((int)sh_addr), 8);
  unsigned char* dyn_smem = (unsigned char*)__builtin_assume_aligned((unsigned char*)(__attribute__((address_space(3))) unsigned char*)(uintptr_t)uni((int)dyn_addr), 16);
  unsigned long long* lstats = (unsigned long long*)__builtin_assume_aligned((unsigned long long*)(__attribute__((address_space(3))) unsigned long long*)(uintptr_t)uni((int)lstats_addr), 8);
  const int cb = uni(cb_v), ce = uni(ce_v), score_remaining = uni(score_remaining_v), known = uni(known_v);
  const int attempt = uni(attempt_v);  // bits 0..7: the attempt, bit 8: a 16-bit search inside a launch with 32-bit rows
  const PassCtx& pc = sh.pctx;
  auto uni64 = [](unsigned long long v) { return ((unsigned long long)(unsigned)uni((int)(v >> 32)) << 32) | (unsigned)uni((int)v); };
  KParams kp{};
  kp.ring = uni(pc.ring);
  kp.wcap = uni(pc.wcap);
  kp.pen.x = uni(pc.x);
  kp.pen.o1 = uni(pc.o1);
  kp.pen.e1 = uni(pc.e1);
  kp.pen.o2 = uni(pc.o2);
  kp.pen.e2 = uni(pc.e2);
  kp.pen.two_piece = P2 ? 1 : 0;
  kp.pen.scope = max(kp.pen.x, max(kp.pen.o1 + kp.pen.e1, P2 ? kp.pen.o2 + kp.pen.e2 : 0)) + 1;
  kp.lds_meta_bytes = uni(pc.lds_meta_bytes);
  kp.lds_seq_bytes = uni(pc.lds_seq_bytes);
  kp.chain_max = uni(pc.chain_max);
  kp.multi_T = uni(pc.multi_T);
  kp.deep_passes = uni(pc.deep_passes);
  kp.ring_slot_stride = (size_t)uni64(pc.ring_bytes);
  SubCtx cx;
  cx.plen = uni(pc.plen);
  cx.tlen = uni(pc.tlen);
  cx.kmin[0] = cx.kmin[1] = 0;
  cx.wcols = 0;
  cx.seq_mode = uni(pc.seq_mode);
  cx.p_w0 = uni(pc.p_w0);
  cx.t_w0 = uni(pc.t_w0);
  cx.p_bit = uni(pc.p_bit);
  cx.t_bit = uni(pc.t_bit);
  cx.P[0] = (gseq_t)(uintptr_t)uni64(pc.P[0]);
  cx.P[1] = (gseq_t)(uintptr_t)uni64(pc.P[1]);
  cx.T[0] = (gseq_t)(uintptr_t)uni64(pc.T[0]);
  cx.T[1] = (gseq_t)(uintptr_t)uni64(pc.T[1]);
  cx.Pw = (gwords_t)(uintptr_t)uni64(pc.Pw);
  cx.Tw = (gwords_t)(uintptr_t)uni64(pc.Tw);
  cx.pb_abs = uni(pc.pb_abs);
  cx.tb_abs = uni(pc.tb_abs);
  Lds<OffT> lds;
  typedef typename MetaTraits<OffT>::Stored MetaStored;
  lds.ring_meta = reinterpret_cast<MetaStored*>(dyn_smem);
  lds.bi_A = reinterpret_cast<int*>(lds.ring_meta + 2 * NCOMP * kp.ring);
  lds.bi_oob = lds.bi_A + 2 * kp.ring;
  lds.firstk = lds.bi_oob + 2 * kp.ring;
  lds.seq = reinterpret_cast<uint32_t*>(dyn_smem + kp.lds_meta_bytes);
  lds.meta_log = nullptr;
  void* ring_mem = (void*)(__attribute__((address_space(1))) char*)(uintptr_t)uni64(pc.ring_mem);  // (a global pointer, not a generic one: the row probes stay global_load)
  const rsrc_t ring_rs = make_rsrc(ring_mem, kp.ring_slot_stride);
  __syncthreads();  // everyone has read pctx before the search rewrites parts of it
  Breakpoint bp;
  const int rc = find_breakpoint<P2, OffT>(kp, sh, lds, cx, ring_mem, ring_rs, cb, ce, score_remaining, known, attempt, bp, lstats);
  if (threadIdx.x == 0) sh.bp_out = bp;
  __syncthreads();
  return rc;
}

// ---------------------------------------------------------------------------------------------
// The kernel: persistent workgroups, one pair at a time, DFS over the BiWFA recursion
// ---------------------------------------------------------------------------------------------
template <bool P2, typename OffT>
__global__ __launch_bounds__(WG, WAVES_PER_SIMD) void biwfa_align_kernel(KParams kp) {
  __shared__ Shared sh;
  __shared__ unsigned long long lstats[STAT_N];
  static_assert(sizeof(Shared) + sizeof(unsigned long long) * STAT_N <= STATIC_LDS_RESERVE, "static LDS reserve");
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  Lds<OffT> lds;
  typedef typename MetaTraits<OffT>::Stored MetaStored;
  lds.ring_meta = reinterpret_cast<MetaStored*>(dyn_smem);
  lds.bi_A = reinterpret_cast<int*>(lds.ring_meta + 2 * NCOMP * kp.ring);
  lds.bi_oob = lds.bi_A + 2 * kp.ring;
  lds.firstk = lds.bi_oob + 2 * kp.ring;
  lds.seq = reinterpret_cast<uint32_t*>(dyn_smem + kp.lds_meta_bytes);
  const int tid = cold_tid();
  const DevPenalties& pn = kp.pen;
  void* ring_mem = (char*)kp.ring_mem + (size_t)blockIdx.x * kp.ring_slot_stride;
  void* hist = (char*)kp.hist_mem + (size_t)blockIdx.x * kp.hist_slot_stride;
  uint32_t* events = kp.ev_mem + (size_t)blockIdx.x * kp.ev_slot_stride;
  Task* const stack = reinterpret_cast<Task*>(events + kp.wcap);  // the DFS stack lives behind the events (HBM; touched once per sub-problem)
  const rsrc_t hist_rs = make_rsrc(hist, kp.hist_slot_stride);
  lds.meta_log = reinterpret_cast<RowMeta*>((char*)hist + kp.hist_meta_offset);
  if (tid < STAT_N) lstats[tid] = 0;
#ifdef AWV_PROF
  if (tid < 5) sh.prof[tid] = 0;
#endif
  const unsigned long long clk_c0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
  if (tid < 16) sh.chain_maxak[tid >> 3][tid & 7] = 0;
  __syncthreads();
  for (;;) {
    if (tid == 0) {
      sh.cur_pair = (long long)atomicAdd(kp.work_counter, 1ULL);
      sh.error = 0;
      sh.win_single = sh.win_multi = sh.win_base = sh.win_base_multi = 0;
    }
    __syncthreads();
    const long long pair = ((long long)uni((int)(sh.cur_pair >> 32)) << 32) | (unsigned)uni((int)sh.cur_pair);
    if (pair >= kp.npairs) break;
    const int qi = kp.pair_q[pair], ti = kp.pair_t[pair];
    const int qv = kp.pair_rc[pair] ? 2 : 0;
    const uint64_t qoff = kp.seq_off[qi], toff = kp.seq_off[ti];
    const int plenT = kp.seq_len[qi], tlenT = kp.seq_len[ti];
    const uint8_t* Pf = kp.seq[qv] + qoff;
    const uint8_t* Pr = kp.seq[qv + 1] + qoff;
    const uint8_t* Tf = kp.seq[0] + toff;
    const uint8_t* Tr = kp.seq[1] + toff;
    // 2-bit packed views: usable when both sequences are pure upper-case ACGT (bytes compare verbatim)
    const int qok = (kp.seq2_ok[qi] >> (qv ? 1 : 0)) & 1, tok = kp.seq2_ok[ti] & 1;
    gwords_t Pw = nullptr, Tw = nullptr;
    if (qok && tok && kp.lds_seq_bytes > 0) {
      Pw = (gwords_t)(uintptr_t)(kp.seq2[qv ? 1 : 0] + kp.seq2_off[qi]);
      Tw = (gwords_t)(uintptr_t)(kp.seq2[0] + kp.seq2_off[ti]);
    }
    const unsigned long long tt0 = PROF_NOW();
    Emit em;
    em.cig = kp.cigar + kp.cigar_off[pair];
    em.n = 0;
    em.cnt[0] = em.cnt[1] = em.cnt[2] = em.cnt[3] = 0;
    int status = ST_OK;
    int penalty = -1;
    int sp = 0;
    if (tid == 0) {
      const bool min_length = max(plenT, tlenT) <= FALLBACK_MIN_LENGTH;
      stack[0] = Task{0, plenT, 0, tlenT, C_M, C_M, min_length ? 0 : INT_MAX, INT_MAX};
    }
    sp = 1;
    __syncthreads();
    bool top = true;
    while (sp > 0 && status == ST_OK) {
      Task t = stack[sp - 1];
      t.pb = uni(t.pb); t.pe = uni(t.pe); t.tb = uni(t.tb); t.te = uni(t.te);
      t.cb = uni(t.cb); t.ce = uni(t.ce); t.score_remaining = uni(t.score_remaining); t.known = uni(t.known);
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
      cx.P[0] = to_global(Pf + t.pb);
      cx.T[0] = to_global(Tf + t.tb);
      cx.P[1] = to_global(Pr + (plenT - t.pe));
      cx.T[1] = to_global(Tr + (tlenT - t.te));
      cx.kmin[0] = cx.kmin[1] = 0;
      cx.wcols = 0;
      cx.Pw = Pw;
      cx.Tw = Tw;
      cx.pb_abs = t.pb;
      cx.tb_abs = t.tb;
      stage_sequences<OffT>(kp, lds, cx);
      bool do_base = t.score_remaining <= FALLBACK_MIN_SCORE;
      Breakpoint bp;
      if (!do_base) {
        int rc = BP_OK;
        for (int attempt = 0; attempt < AWV_RESTART_ATTEMPTS; ++attempt) {  // (one call site: the search stays inlined)
          // later attempts: the searches met before the I/D history was being kept -- once more with a wider margin, then step by step
          if (tid == 0) {  // the search's inputs (find_breakpoint_fn reads them back after a barrier)
            PassCtx& pc = sh.pctx;
            pc.ring_mem = (unsigned long long)(uintptr_t)ring_mem;
            pc.ring_bytes = (unsigned long long)kp.ring_slot_stride;
            pc.P[0] = (unsigned long long)(uintptr_t)cx.P[0];
            pc.P[1] = (unsigned long long)(uintptr_t)cx.P[1];
            pc.T[0] = (unsigned long long)(uintptr_t)cx.T[0];
            pc.T[1] = (unsigned long long)(uintptr_t)cx.T[1];
            pc.Pw = (unsigned long long)(uintptr_t)cx.Pw;
            pc.Tw = (unsigned long long)(uintptr_t)cx.Tw;
            pc.ring = kp.ring;
            pc.wcap = kp.wcap;
            pc.x = pn.x; pc.o1 = pn.o1; pc.e1 = pn.e1; pc.o2 = pn.o2; pc.e2 = pn.e2;
            pc.lds_meta_bytes = kp.lds_meta_bytes;
            pc.lds_seq_bytes = kp.lds_seq_bytes;
            pc.chain_max = kp.chain_max;
            pc.multi_T = kp.multi_T;
            pc.deep_passes = kp.deep_passes;
            pc.plen = plen; pc.tlen = tlen;
            pc.seq_mode = cx.seq_mode; pc.p_w0 = cx.p_w0; pc.t_w0 = cx.t_w0; pc.p_bit = cx.p_bit; pc.t_bit = cx.t_bit;
            pc.pb_abs = cx.pb_abs; pc.tb_abs = cx.tb_abs;
          }
          __syncthreads();
          // A launch with 32-bit rows searches the sub-problems that have become short enough with 16-bit rows (AWV_SUB16): the
          // same ring arena at half the bytes per row, packed arithmetic, the 16-bit metadata layout inside the same LDS region --
          // a search leaves nothing behind but its breakpoint.  (The base case stays with the launch's row width.)
          if constexpr (AWV_SUB16 && sizeof(OffT) == 4 && !WENC) {
            if (kp.sub16 != 0 && max(plen, tlen) < SUB16_MAX_LEN)
              rc = uni(find_breakpoint_fn<P2, int16_t>((unsigned)(uintptr_t)&sh, (unsigned)(uintptr_t)lds.ring_meta, (unsigned)(uintptr_t)lstats,
                                                       t.cb, t.ce, t.score_remaining, t.known, attempt | 0x100));  // (0x100: the long reads' margin)
            else
              rc = uni(find_breakpoint_fn<P2, OffT>((unsigned)(uintptr_t)&sh, (unsigned)(uintptr_t)lds.ring_meta, (unsigned)(uintptr_t)lstats,
                                                    t.cb, t.ce, t.score_remaining, t.known, attempt));
          } else {
            rc = uni(find_breakpoint_fn<P2, OffT>((unsigned)(uintptr_t)&sh, (unsigned)(uintptr_t)lds.ring_meta, (unsigned)(uintptr_t)lstats,
                                                  t.cb, t.ce, t.score_remaining, t.known, attempt));
          }
          if (rc == BP_OK) {
            bp.score = uni(sh.bp_out.score); bp.sf = uni(sh.bp_out.sf); bp.sr = uni(sh.bp_out.sr); bp.kf = uni(sh.bp_out.kf);
            bp.kr = uni(sh.bp_out.kr); bp.off_f = uni(sh.bp_out.off_f); bp.off_r = uni(sh.bp_out.off_r); bp.comp = uni(sh.bp_out.comp);
          }
          if (rc != BP_RESTART) break;
          if (tid == 0) lstats[STAT_RESTARTS] += 1;
        }
        if (rc == BP_END_REACHED) do_base = true;  // wavefront_bialign_exception -> plain WFA
        else if (rc != BP_OK) { status = rc; break; }
      }
      if (do_base) {
        int pen_b = 0;
        const int rc = base_align<P2, OffT>(kp, sh, lds, cx, hist, hist_rs, events, t.cb, t.ce, em, pen_b, lstats);
        if (rc != ST_OK) { status = rc; break; }
        if (top) penalty = pen_b;
        top = false;
        continue;
      }
      const int bh = bp.off_f, bv = bp.off_f - bp.kf;
      if (bh < 0 || bh > tlen || bv < 0 || bv > plen || sp + 2 > STACK_CAP) { status = ST_INTERNAL; break; }
      if (tid == 0) {
        const int open_c = bp.comp == C_M ? 0 : ((bp.comp == C_I1 || bp.comp == C_D1) ? kp.pen.o1 : kp.pen.o2);
        stack[sp] = Task{t.pb + bv, t.pe, t.tb + bh, t.te, bp.comp, t.ce, bp.sr, bp.sr - open_c};      // right half
        stack[sp + 1] = Task{t.pb, t.pb + bv, t.tb, t.tb + bh, t.cb, bp.comp, bp.sf, bp.sf - open_c};  // left half first
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
      PROF_ADD(STAT_T_TOTAL, tt0);
      lstats[STAT_WIN_SINGLE] += sh.win_single; lstats[STAT_WIN_MULTI] += sh.win_multi; lstats[STAT_WIN_BASE] += sh.win_base; lstats[STAT_WIN_BASE_MULTI] += sh.win_base_multi;
      sh.win_single = sh.win_multi = sh.win_base = sh.win_base_multi = 0;
      if (status == ST_OK) {
        lstats[STAT_ALIGNED_BP] += (unsigned long long)plenT;
        lstats[STAT_PAIRS] += 1;
      }
    }
    __syncthreads();
  }
  __syncthreads();
  if (tid == 0) {
    lstats[STAT_CLK_CYCLES] = __builtin_amdgcn_s_memtime() - clk_c0;
    lstats[STAT_CLK_TICKS] = __builtin_amdgcn_s_memrealtime() - clk_r0;
  }
#ifdef AWV_PROF
  if (tid < 5) lstats[STAT_T_CR_LOAD + tid] = sh.prof[tid];
#endif
  __syncthreads();
  if (tid < STAT_N && lstats[tid]) atomicAdd(&kp.stats[tid], lstats[tid]);
}

}  // namespace AWV_NS
