/*
 * biwfa_oracle.c -- CPU restatement of WFA2-lib's end-to-end BiWFA (gap-affine and 2-piece
 * gap-affine), the arithmetic behind the reference's `wf.align(query, target)`
 * (/root/reference/src/alignment.rs:226-236, src/wfa.rs:221-231).
 *
 * TEST INFRASTRUCTURE ONLY (see biwfa_oracle.h).  PARITY UNPINNED against WFA2-lib itself:
 * lib_wfa2 @ 2f9d9a48 / WFA2-lib are not in the container (Cargo.toml:27, Cargo.lock:598-601);
 * this file follows the published WFA / BiWFA algorithms with WFA2-lib's conventions as
 * recorded in SURVEY.md Appendix A (A.1 .. A.7).  Each function names the WFA2-lib routine it
 * restates [RECALLED] and the Appendix-A paragraph it implements.
 *
 * Structure (function-for-function after WFA2-lib so a later comparison localises drift):
 *   wf_compute_affine / wf_compute_affine2p   <- wavefront_compute_affine{,2p}      (A.3)
 *   wf_extend_end2end                         <- wavefront_extend_end2end{,_max}    (A.4)
 *   wf_backtrace_affine                       <- wavefront_backtrace_affine         (A.5)
 *   bialign_overlap / _breakpoint_*           <- wavefront_bialign_overlap etc.     (A.6)
 *   bialign_find_breakpoint                   <- wavefront_bialign_find_breakpoint  (A.6)
 *   bialign_alignment / bialign_base          <- wavefront_bialign_alignment/_base  (A.6)
 */
#include "biwfa_oracle.h"

#include <limits.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int32_t wf_offset_t;

/* A.1: offsets store h; v = offset - k; NULL = INT32_MIN/2 */
#define WF_OFFSET_NULL (INT32_MIN / 2)
#define WF_H(k, off) (off)
#define WF_V(k, off) ((off) - (k))
#define WF_ANTIDIAGONAL(k, off) (2 * (off) - (k))
#define WF_K_INVERSE(k, plen, tlen) ((tlen) - (plen) - (k))

#define WF_BIALIGN_FALLBACK_MIN_SCORE 250  /* A.6 */
#define WF_BIALIGN_FALLBACK_MIN_LENGTH 100 /* A.6 (short sequences go straight to plain WFA) */

#define WF_STATUS_OK 0
#define WF_STATUS_END_REACHED 1
#define WF_STATUS_ERROR (-1)

#define MAXI(a, b) ((a) > (b) ? (a) : (b))
#define MINI(a, b) ((a) < (b) ? (a) : (b))

enum { COMP_M = 0, COMP_I1 = 1, COMP_I2 = 2, COMP_D1 = 3, COMP_D2 = 4, NCOMP = 5 };

#define SEQ_PAD 16
#define PATTERN_EOS '!'
#define TEXT_EOS '?'

/* ------------------------------------------------------------------------------------------
 * One wavefront (one component at one score).  `present` == WFA2's pointer being non-NULL;
 * `null` == WFA2's wavefront->null flag (set by trimming to empty).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  bool present;
  bool null;
  int lo, hi;
  int init_min, init_max; /* cells in [init_min,init_max] hold defined values */
  int max_ak;             /* M rows: max antidiagonal after extension (fast-overlap filter only) */
  bool saw_oob;           /* some non-NULL value of this score was out of bounds (filter only) */
  int alo, ahi;           /* allocated index range */
  wf_offset_t* mem;       /* storage; cell k lives at mem[k - alo] */
  size_t cap;             /* elements allocated (modular rows own their buffer) */
} wavefront_t;

#define WF_AT(w, k) ((w)->mem[(k) - (w)->alo])

/* bump arena for the high-memory (full history) aligner */
typedef struct arena_chunk {
  struct arena_chunk* next;
  size_t cap, used;
  wf_offset_t data[];
} arena_chunk_t;

typedef struct {
  arena_chunk_t* head;    /* chunk list (kept across alignments) */
  arena_chunk_t* current;
} arena_t;

static void arena_reset(arena_t* ar) {
  for (arena_chunk_t* c = ar->head; c; c = c->next) c->used = 0;
  ar->current = ar->head;
}
static wf_offset_t* arena_alloc(arena_t* ar, size_t n) {
  while (ar->current) {
    if (ar->current->cap - ar->current->used >= n) {
      wf_offset_t* p = ar->current->data + ar->current->used;
      ar->current->used += n;
      return p;
    }
    if (!ar->current->next) break;
    ar->current = ar->current->next;
  }
  size_t cap = n > ((size_t)1 << 20) ? n : ((size_t)1 << 20);
  arena_chunk_t* c = (arena_chunk_t*)malloc(sizeof(arena_chunk_t) + cap * sizeof(wf_offset_t));
  if (!c) return NULL;
  c->next = NULL;
  c->cap = cap;
  c->used = n;
  if (ar->current) ar->current->next = c; else ar->head = c;
  ar->current = c;
  return c->data;
}
static void arena_free(arena_t* ar) {
  arena_chunk_t* c = ar->head;
  while (c) { arena_chunk_t* n = c->next; free(c); c = n; }
  ar->head = ar->current = NULL;
}

/* ------------------------------------------------------------------------------------------
 * One unidirectional aligner state (WFA2's wavefront_aligner_t, the part used here).
 * modular == score-only ring of max_score_scope rows (forward / reverse BiWFA aligners);
 * !modular == full history (the "subsidiary" aligner used by base cases, A.6).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  /* penalties */
  int x, o1, e1, o2, e2;
  bool two_piece;
  int max_score_scope; /* A.3 ring depth: max(x, o1+e1, o2+e2) + 1 */
  /* sequences (views into padded buffers) */
  const char* pattern;
  const char* text;
  int plen, tlen;
  /* components */
  bool modular;
  int nrows;                /* modular: max_score_scope; else capacity in scores */
  wavefront_t* wf[NCOMP];   /* wf[c][score_index] */
  wavefront_t wavefront_null;
  int historic_min_lo, historic_max_hi;
  arena_t arena;
  /* alignment state */
  int component_begin, component_end;
  int end_score, end_k;
  wf_offset_t end_offset;
  int status;
  awo_stats_t* stats;
  bool fast_overlap; /* exact pre-filter for the overlap search (CPU-baseline mode; same results) */
} wf_aligner_t;

struct awo_aligner {
  awo_penalties_t pen;
  wf_aligner_t fwd, rev, sub;
  /* padded sequence buffers (A.4): forward and reversed copies */
  char *pbuf, *tbuf, *prbuf, *trbuf;
  size_t pcap, tcap;
  int plen_total, tlen_total;
  /* current bounds + saved chars under the temporary sentinels */
  int pb, pe, tb, te;
  bool bounds_set;
  char saved[4];
  /* cigar under construction */
  uint8_t* cigar;
  int cigar_cap, cigar_n;
  uint8_t* bt_buf; /* backtrace scratch (filled end->start) */
  int bt_cap;
  awo_stats_t* stats;
  int error;
};

static int score_index(const wf_aligner_t* a, int score) {
  return a->modular ? score % a->max_score_scope : score;
}

static void aligner_init(wf_aligner_t* a, const awo_penalties_t* pen, bool modular) {
  memset(a, 0, sizeof(*a));
  a->x = pen->mismatch;
  a->o1 = pen->gap_open1;
  a->e1 = pen->gap_ext1;
  a->two_piece = pen->two_piece != 0;
  a->o2 = a->two_piece ? pen->gap_open2 : pen->gap_open1;
  a->e2 = a->two_piece ? pen->gap_ext2 : pen->gap_ext1;
  int scope = MAXI(a->x, a->o1 + a->e1);
  if (a->two_piece) scope = MAXI(scope, a->o2 + a->e2);
  a->max_score_scope = scope + 1;
  a->modular = modular;
  a->nrows = modular ? a->max_score_scope : 0;
  for (int c = 0; c < NCOMP; ++c) {
    a->wf[c] = modular ? (wavefront_t*)calloc((size_t)a->nrows, sizeof(wavefront_t)) : NULL;
  }
}

static void aligner_destroy(wf_aligner_t* a) {
  for (int c = 0; c < NCOMP; ++c) {
    if (a->wf[c]) {
      if (a->modular)
        for (int i = 0; i < a->nrows; ++i) free(a->wf[c][i].mem);
      free(a->wf[c]);
    }
  }
  free(a->wavefront_null.mem);
  arena_free(&a->arena);
}

/* grow the per-score arrays of the full-history aligner */
static bool aligner_reserve_scores(wf_aligner_t* a, int score) {
  if (a->modular || score < a->nrows) return true;
  int n = a->nrows ? a->nrows : 512;
  while (n <= score) n *= 2;
  for (int c = 0; c < NCOMP; ++c) {
    wavefront_t* p = (wavefront_t*)realloc(a->wf[c], (size_t)n * sizeof(wavefront_t));
    if (!p) return false;
    memset(p + a->nrows, 0, (size_t)(n - a->nrows) * sizeof(wavefront_t));
    a->wf[c] = p;
  }
  a->nrows = n;
  return true;
}

/* (re)allocate the storage of an output wavefront for index range [alo,ahi]
 * (WFA2: wavefront_slab_allocate with the historic lo/hi) */
static bool wavefront_allocate(wf_aligner_t* a, wavefront_t* w, int alo, int ahi) {
  size_t n = (size_t)(ahi - alo + 1);
  if (a->modular) {
    if (w->cap < n) {
      size_t cap = w->cap ? w->cap : 256;
      while (cap < n) cap *= 2;
      wf_offset_t* p = (wf_offset_t*)realloc(w->mem, cap * sizeof(wf_offset_t));
      if (!p) return false;
      w->mem = p;
      w->cap = cap;
    }
  } else {
    w->mem = arena_alloc(&a->arena, n);
    if (!w->mem) return false;
    w->cap = n;
  }
  w->alo = alo;
  w->ahi = ahi;
  return true;
}

/* the shared all-NULL wavefront (WFA2: wf_components.wavefront_null; lo=1, hi=-1) */
static bool wavefront_null_cover(wf_aligner_t* a, int lo, int hi) {
  wavefront_t* w = &a->wavefront_null;
  if (w->mem && w->alo <= lo && hi <= w->ahi) return true;
  int alo = w->mem ? MINI(w->alo, lo) : lo;
  int ahi = w->mem ? MAXI(w->ahi, hi) : hi;
  alo -= 64;
  ahi += 64;
  size_t n = (size_t)(ahi - alo + 1);
  wf_offset_t* p = (wf_offset_t*)realloc(w->mem, n * sizeof(wf_offset_t));
  if (!p) return false;
  for (size_t i = 0; i < n; ++i) p[i] = WF_OFFSET_NULL;
  w->mem = p;
  w->cap = n;
  w->alo = alo;
  w->ahi = ahi;
  w->present = false;
  w->null = true;
  w->lo = 1;
  w->hi = -1;
  w->init_min = alo;
  w->init_max = ahi;
  return true;
}

/* WFA2: wavefront_compute_init_ends -- make cells of an input readable (as NULL) over
 * [lo,hi] beyond its own (trimmed) range */
static bool wavefront_init_ends(wf_aligner_t* a, wavefront_t* w, int lo, int hi) {
  if (w == &a->wavefront_null) return wavefront_null_cover(a, lo, hi);
  if (lo < w->alo || hi > w->ahi) return false; /* historic allocation must cover it */
  if (w->init_min > w->init_max) { /* nothing initialised (trimmed to empty) */
    for (int k = lo; k <= hi; ++k) WF_AT(w, k) = WF_OFFSET_NULL;
    w->init_min = lo;
    w->init_max = hi;
    return true;
  }
  if (w->init_max < hi) {
    for (int k = w->init_max + 1; k <= hi; ++k) WF_AT(w, k) = WF_OFFSET_NULL;
    w->init_max = hi;
  }
  if (w->init_min > lo) {
    for (int k = lo; k < w->init_min; ++k) WF_AT(w, k) = WF_OFFSET_NULL;
    w->init_min = lo;
  }
  return true;
}

/* WFA2: wavefront_compute_get_*wavefront -- negative score or absent row => null wavefront */
static wavefront_t* fetch_wavefront(wf_aligner_t* a, int comp, int score) {
  if (score < 0) return &a->wavefront_null;
  if (!a->modular && score >= a->nrows) return &a->wavefront_null;
  wavefront_t* w = &a->wf[comp][score_index(a, score)];
  return w->present ? w : &a->wavefront_null;
}

/* WFA2: wavefront_compute_trim_ends (A.3 "trimmed inward past out-of-range/NULL ends") */
static void wavefront_trim_ends(const wf_aligner_t* a, wavefront_t* w) {
  const uint32_t plen = (uint32_t)a->plen, tlen = (uint32_t)a->tlen;
  int k;
  const int lo = w->lo;
  for (k = w->hi; k >= lo; --k) {
    const wf_offset_t off = WF_AT(w, k);
    const uint32_t h = (uint32_t)WF_H(k, off), v = (uint32_t)WF_V(k, off);
    if (h <= tlen && v <= plen) break;
  }
  w->hi = k;
  w->init_max = k;
  const int hi = w->hi;
  for (k = w->lo; k <= hi; ++k) {
    const wf_offset_t off = WF_AT(w, k);
    const uint32_t h = (uint32_t)WF_H(k, off), v = (uint32_t)WF_V(k, off);
    if (h <= tlen && v <= plen) break;
  }
  w->lo = k;
  w->init_min = k;
  w->null = (w->lo > w->hi);
}

/* mark the rows of a null step absent (WFA2: wavefront_compute_allocate_output_null) */
static void allocate_output_null(wf_aligner_t* a, int score) {
  const int si = score_index(a, score);
  for (int c = 0; c < NCOMP; ++c) a->wf[c][si].present = false;
}

static wavefront_t* allocate_output(wf_aligner_t* a, int comp, int score, int lo, int hi) {
  wavefront_t* w = &a->wf[comp][score_index(a, score)];
  if (!wavefront_allocate(a, w, a->historic_min_lo, a->historic_max_hi)) return NULL;
  w->present = true;
  w->null = false;
  w->max_ak = 0;
  w->saw_oob = false;
  w->lo = lo;
  w->hi = hi;
  w->init_min = lo;
  w->init_max = hi;
  return w;
}

/* WFA2: wavefront_compute_affine / wavefront_compute_affine2p (A.3).  Returns false on OOM. */
static bool wf_compute(wf_aligner_t* a, int score) {
  if (!aligner_reserve_scores(a, score)) return false;
  const bool p2 = a->two_piece;
  /* wavefront_compute_fetch_input */
  wavefront_t* m_misms = fetch_wavefront(a, COMP_M, score - a->x);
  wavefront_t* m_open1 = fetch_wavefront(a, COMP_M, score - a->o1 - a->e1);
  wavefront_t* i1_ext = fetch_wavefront(a, COMP_I1, score - a->e1);
  wavefront_t* d1_ext = fetch_wavefront(a, COMP_D1, score - a->e1);
  wavefront_t* m_open2 = p2 ? fetch_wavefront(a, COMP_M, score - a->o2 - a->e2) : &a->wavefront_null;
  wavefront_t* i2_ext = p2 ? fetch_wavefront(a, COMP_I2, score - a->e2) : &a->wavefront_null;
  wavefront_t* d2_ext = p2 ? fetch_wavefront(a, COMP_D2, score - a->e2) : &a->wavefront_null;
  /* the shared null wavefront always reads lo=1, hi=-1 */
  a->wavefront_null.lo = 1;
  a->wavefront_null.hi = -1;
  a->wavefront_null.null = true;
  /* null step */
  bool all_null = m_misms->null && m_open1->null && i1_ext->null && d1_ext->null;
  if (p2) all_null = all_null && m_open2->null && i2_ext->null && d2_ext->null;
  if (all_null) {
    allocate_output_null(a, score);
    return true;
  }
  /* wavefront_compute_limits_input (A.3 limits) */
  int lo = m_misms->lo, hi = m_misms->hi;
  if (lo > m_open1->lo - 1) lo = m_open1->lo - 1;
  if (hi < m_open1->hi + 1) hi = m_open1->hi + 1;
  if (lo > i1_ext->lo + 1) lo = i1_ext->lo + 1;
  if (hi < i1_ext->hi + 1) hi = i1_ext->hi + 1;
  if (lo > d1_ext->lo - 1) lo = d1_ext->lo - 1;
  if (hi < d1_ext->hi - 1) hi = d1_ext->hi - 1;
  if (p2) {
    if (lo > m_open2->lo - 1) lo = m_open2->lo - 1;
    if (hi < m_open2->hi + 1) hi = m_open2->hi + 1;
    if (lo > i2_ext->lo + 1) lo = i2_ext->lo + 1;
    if (hi < i2_ext->hi + 1) hi = i2_ext->hi + 1;
    if (lo > d2_ext->lo - 1) lo = d2_ext->lo - 1;
    if (hi < d2_ext->hi - 1) hi = d2_ext->hi - 1;
  }
  /* wavefront_compute_allocate_output: effective (padded) dims tracked historically */
  const int eff_lo = lo - (a->max_score_scope + 1), eff_hi = hi + (a->max_score_scope + 1);
  if (a->historic_min_lo > eff_lo) a->historic_min_lo = eff_lo;
  if (a->historic_max_hi < eff_hi) a->historic_max_hi = eff_hi;
  wavefront_t* out_m = allocate_output(a, COMP_M, score, lo, hi);
  wavefront_t* out_i1 = allocate_output(a, COMP_I1, score, lo, hi);
  wavefront_t* out_d1 = allocate_output(a, COMP_D1, score, lo, hi);
  wavefront_t* out_i2 = p2 ? allocate_output(a, COMP_I2, score, lo, hi) : NULL;
  wavefront_t* out_d2 = p2 ? allocate_output(a, COMP_D2, score, lo, hi) : NULL;
  if (!out_m || !out_i1 || !out_d1 || (p2 && (!out_i2 || !out_d2))) return false;
  if (!p2) {
    a->wf[COMP_I2][score_index(a, score)].present = false;
    a->wf[COMP_D2][score_index(a, score)].present = false;
  }
  /* wavefront_compute_init_ends */
  bool ok = wavefront_init_ends(a, m_misms, lo, hi) && wavefront_init_ends(a, m_open1, lo - 1, hi + 1) &&
            wavefront_init_ends(a, i1_ext, lo - 1, hi) && wavefront_init_ends(a, d1_ext, lo, hi + 1);
  if (p2)
    ok = ok && wavefront_init_ends(a, m_open2, lo - 1, hi + 1) && wavefront_init_ends(a, i2_ext, lo - 1, hi) &&
         wavefront_init_ends(a, d2_ext, lo, hi + 1);
  if (!ok) return false;
  /* kernel: wavefront_compute_affine{,2p}_idm */
  const uint32_t plen = (uint32_t)a->plen, tlen = (uint32_t)a->tlen;
  const wf_offset_t* pm_misms = m_misms->mem - m_misms->alo;
  const wf_offset_t* pm_open1 = m_open1->mem - m_open1->alo;
  const wf_offset_t* pi1_ext = i1_ext->mem - i1_ext->alo;
  const wf_offset_t* pd1_ext = d1_ext->mem - d1_ext->alo;
  wf_offset_t* po_m = out_m->mem - out_m->alo;
  wf_offset_t* po_i1 = out_i1->mem - out_i1->alo;
  wf_offset_t* po_d1 = out_d1->mem - out_d1->alo;
  if (!p2) {
#pragma GCC ivdep
    for (int k = lo; k <= hi; ++k) {
      const wf_offset_t ins1 = MAXI(pm_open1[k - 1], pi1_ext[k - 1]) + 1;
      po_i1[k] = ins1;
      const wf_offset_t del1 = MAXI(pm_open1[k + 1], pd1_ext[k + 1]);
      po_d1[k] = del1;
      const wf_offset_t misms = pm_misms[k] + 1;
      wf_offset_t max = MAXI(del1, MAXI(misms, ins1));
      const uint32_t h = (uint32_t)WF_H(k, max), v = (uint32_t)WF_V(k, max);
      if (h > tlen) max = WF_OFFSET_NULL;
      if (v > plen) max = WF_OFFSET_NULL;
      po_m[k] = max;
    }
  } else {
    const wf_offset_t* pm_open2 = m_open2->mem - m_open2->alo;
    const wf_offset_t* pi2_ext = i2_ext->mem - i2_ext->alo;
    const wf_offset_t* pd2_ext = d2_ext->mem - d2_ext->alo;
    wf_offset_t* po_i2 = out_i2->mem - out_i2->alo;
    wf_offset_t* po_d2 = out_d2->mem - out_d2->alo;
#pragma GCC ivdep
    for (int k = lo; k <= hi; ++k) {
      const wf_offset_t ins1 = MAXI(pm_open1[k - 1], pi1_ext[k - 1]) + 1;
      po_i1[k] = ins1;
      const wf_offset_t ins2 = MAXI(pm_open2[k - 1], pi2_ext[k - 1]) + 1;
      po_i2[k] = ins2;
      const wf_offset_t ins = MAXI(ins1, ins2);
      const wf_offset_t del1 = MAXI(pm_open1[k + 1], pd1_ext[k + 1]);
      po_d1[k] = del1;
      const wf_offset_t del2 = MAXI(pm_open2[k + 1], pd2_ext[k + 1]);
      po_d2[k] = del2;
      const wf_offset_t del = MAXI(del1, del2);
      const wf_offset_t misms = pm_misms[k] + 1;
      wf_offset_t max = MAXI(del, MAXI(misms, ins));
      const uint32_t h = (uint32_t)WF_H(k, max), v = (uint32_t)WF_V(k, max);
      if (h > tlen) max = WF_OFFSET_NULL;
      if (v > plen) max = WF_OFFSET_NULL;
      po_m[k] = max;
    }
  }
  if (a->fast_overlap) { /* any non-NULL value out of bounds at this score? (the M candidate is the max of all) */
    int oob = 0;
    const wf_offset_t* pi1 = po_i1; const wf_offset_t* pd1 = po_d1;
    const wf_offset_t* pi2 = p2 ? out_i2->mem - out_i2->alo : pi1;
    const wf_offset_t* pd2 = p2 ? out_d2->mem - out_d2->alo : pd1;
#pragma GCC ivdep
    for (int k = lo; k <= hi; ++k) {
      const int hmax = MINI((int)tlen, (int)plen + k);
      wf_offset_t mx = MAXI(MAXI(pi1[k], pd1[k]), MAXI(pi2[k], pd2[k]));
      const wf_offset_t mm = pm_misms[k] + 1;
      mx = MAXI(mx, mm);
      oob |= (mx >= 0) & (mx > hmax);
    }
    out_m->saw_oob = oob != 0;
    out_m->max_ak = 0;
  }
  /* wavefront_compute_process_ends */
  wavefront_trim_ends(a, out_m);
  wavefront_trim_ends(a, out_i1);
  wavefront_trim_ends(a, out_d1);
  if (p2) {
    wavefront_trim_ends(a, out_i2);
    wavefront_trim_ends(a, out_d2);
  }
  if (a->stats) {
    a->stats->cell_steps += (uint64_t)(hi - lo + 1);
    if ((uint32_t)(hi - lo + 1) > a->stats->max_width) a->stats->max_width = (uint32_t)(hi - lo + 1);
  }
  return true;
}

/* WFA2: wavefront_extend_matches_packed_kernel (A.4): 8 bytes at a time; the buffers carry
 * distinct sentinels past each (sub)sequence end.  The explicit clamp keeps the result the
 * true bounded LCP even when a raw byte equals the other sequence's sentinel. */
static inline wf_offset_t extend_matches_packed(const wf_aligner_t* a, int k, wf_offset_t offset,
                                                uint64_t* bytes) {
  const int v0 = WF_V(k, offset), h0 = WF_H(k, offset);
  const int rem = MINI(a->plen - v0, a->tlen - h0);
  const char* p = a->pattern + v0;
  const char* t = a->text + h0;
  int n = 0;
  uint64_t pw, tw;
  memcpy(&pw, p, 8);
  memcpy(&tw, t, 8);
  uint64_t cmp = pw ^ tw;
  while (__builtin_expect(cmp == 0, 0)) {
    n += 8;
    if (n >= rem) break;
    memcpy(&pw, p + n, 8);
    memcpy(&tw, t + n, 8);
    cmp = pw ^ tw;
  }
  if (cmp != 0) n += __builtin_ctzll(cmp) / 8;
  if (n > rem) n = rem;
  *bytes += (uint64_t)n + 1;
  return offset + n;
}

/* WFA2: wavefront_termination_end2end */
static bool wf_termination_end2end(wf_aligner_t* a, wavefront_t* mwf, int score) {
  const int alignment_k = a->tlen - a->plen;
  const wf_offset_t alignment_offset = a->tlen;
  wavefront_t* w = mwf;
  if (a->component_end != COMP_M) {
    w = &a->wf[a->component_end][score_index(a, score)];
    if (!w->present) return false;
  }
  if (w->lo > alignment_k || alignment_k > w->hi) return false;
  if (WF_AT(w, alignment_k) < alignment_offset) return false;
  a->end_score = score;
  a->end_k = alignment_k;
  a->end_offset = alignment_offset;
  return true;
}

/* WFA2: wavefront_extend_end2end / wavefront_extend_end2end_max (A.4).
 * Returns 1 when the end was reached (status set), else 0.  *max_ak (nullable) gets the max
 * antidiagonal reached by this wavefront (0 if none). */
static int wf_extend_end2end(wf_aligner_t* a, int score, int* max_ak) {
  if (max_ak) *max_ak = 0;
  if (!a->modular && score >= a->nrows) return 0;
  wavefront_t* mwf = &a->wf[COMP_M][score_index(a, score)];
  if (!mwf->present) return 0;
  uint64_t bytes = 0;
  wf_offset_t max_antidiag = 0;
  wf_offset_t* off = mwf->mem - mwf->alo;
  for (int k = mwf->lo; k <= mwf->hi; ++k) {
    const wf_offset_t o = off[k];
    if (o < 0) continue;
    const wf_offset_t e = extend_matches_packed(a, k, o, &bytes);
    off[k] = e;
    const wf_offset_t ad = WF_ANTIDIAGONAL(k, e);
    if (max_antidiag < ad) max_antidiag = ad;
  }
  if (a->stats) a->stats->extend_bytes += bytes;
  mwf->max_ak = max_antidiag;
  if (max_ak) *max_ak = max_antidiag;
  if (wf_termination_end2end(a, mwf, score)) {
    a->status = WF_STATUS_END_REACHED;
    return 1;
  }
  return 0;
}

/* WFA2: wavefront_unialign_init + initial wavefronts by begin component (A.6 recursion) */
static bool wf_unialign_init(wf_aligner_t* a, int component_begin, int component_end) {
  a->component_begin = component_begin;
  a->component_end = component_end;
  a->status = WF_STATUS_OK;
  a->end_score = -1;
  a->end_k = 0;
  a->end_offset = 0;
  const int pad = a->max_score_scope + 1;
  a->historic_min_lo = -pad;
  a->historic_max_hi = pad;
  if (a->modular) {
    for (int c = 0; c < NCOMP; ++c)
      for (int i = 0; i < a->nrows; ++i) a->wf[c][i].present = false;
  } else {
    arena_reset(&a->arena);
    for (int c = 0; c < NCOMP; ++c)
      if (a->wf[c]) memset(a->wf[c], 0, (size_t)a->nrows * sizeof(wavefront_t));
    if (!aligner_reserve_scores(a, 0)) return false;
  }
  if (!a->two_piece && (component_begin == COMP_I2 || component_begin == COMP_D2)) return false;
  wavefront_t* w = allocate_output(a, component_begin, 0, 0, 0);
  if (!w) return false;
  WF_AT(w, 0) = 0;
  w->max_ak = 0;
  w->saw_oob = false;
  return true;
}

/* ------------------------------------------------------------------------------------------
 * sequences (A.4): padded copies, reversed copies, temporary sentinels at sub-range ends
 * ------------------------------------------------------------------------------------------ */
static bool seqs_load(awo_aligner_t* A, const uint8_t* pattern, int plen, const uint8_t* text, int tlen) {
  size_t pn = (size_t)plen + 2 * SEQ_PAD, tn = (size_t)tlen + 2 * SEQ_PAD;
  if (A->pcap < pn) {
    free(A->pbuf); free(A->prbuf);
    A->pcap = pn * 2;
    A->pbuf = (char*)malloc(A->pcap);
    A->prbuf = (char*)malloc(A->pcap);
  }
  if (A->tcap < tn) {
    free(A->tbuf); free(A->trbuf);
    A->tcap = tn * 2;
    A->tbuf = (char*)malloc(A->tcap);
    A->trbuf = (char*)malloc(A->tcap);
  }
  if (!A->pbuf || !A->prbuf || !A->tbuf || !A->trbuf) return false;
  memset(A->pbuf, PATTERN_EOS, pn);
  memset(A->prbuf, PATTERN_EOS, pn);
  memset(A->tbuf, TEXT_EOS, tn);
  memset(A->trbuf, TEXT_EOS, tn);
  memcpy(A->pbuf + SEQ_PAD, pattern, (size_t)plen);
  memcpy(A->tbuf + SEQ_PAD, text, (size_t)tlen);
  for (int i = 0; i < plen; ++i) A->prbuf[SEQ_PAD + i] = (char)pattern[plen - 1 - i];
  for (int i = 0; i < tlen; ++i) A->trbuf[SEQ_PAD + i] = (char)text[tlen - 1 - i];
  A->plen_total = plen;
  A->tlen_total = tlen;
  A->bounds_set = false;
  return true;
}

/* WFA2: wavefront_bialigner_set_sequences_bounds */
static void seqs_set_bounds(awo_aligner_t* A, int pb, int pe, int tb, int te) {
  const int P = A->plen_total, T = A->tlen_total;
  if (A->bounds_set) { /* restore the chars under the previous sentinels */
    A->pbuf[SEQ_PAD + A->pe] = A->saved[0];
    A->tbuf[SEQ_PAD + A->te] = A->saved[1];
    A->prbuf[SEQ_PAD + (P - A->pb)] = A->saved[2];
    A->trbuf[SEQ_PAD + (T - A->tb)] = A->saved[3];
  }
  A->pb = pb; A->pe = pe; A->tb = tb; A->te = te;
  A->saved[0] = A->pbuf[SEQ_PAD + pe];
  A->saved[1] = A->tbuf[SEQ_PAD + te];
  A->saved[2] = A->prbuf[SEQ_PAD + (P - pb)];
  A->saved[3] = A->trbuf[SEQ_PAD + (T - tb)];
  A->pbuf[SEQ_PAD + pe] = PATTERN_EOS;
  A->tbuf[SEQ_PAD + te] = TEXT_EOS;
  A->prbuf[SEQ_PAD + (P - pb)] = PATTERN_EOS;
  A->trbuf[SEQ_PAD + (T - tb)] = TEXT_EOS;
  A->bounds_set = true;
  const int plen = pe - pb, tlen = te - tb;
  wf_aligner_t* f = &A->fwd; wf_aligner_t* r = &A->rev; wf_aligner_t* s = &A->sub;
  f->pattern = s->pattern = A->pbuf + SEQ_PAD + pb;
  f->text = s->text = A->tbuf + SEQ_PAD + tb;
  r->pattern = A->prbuf + SEQ_PAD + (P - pe);
  r->text = A->trbuf + SEQ_PAD + (T - te);
  f->plen = r->plen = s->plen = plen;
  f->tlen = r->tlen = s->tlen = tlen;
}

/* ------------------------------------------------------------------------------------------
 * CIGAR helpers
 * ------------------------------------------------------------------------------------------ */
static void cigar_append(awo_aligner_t* A, uint8_t op, int n) {
  if (A->cigar_n + n > A->cigar_cap) { A->error = AWO_ERR_CAPACITY; return; }
  memset(A->cigar + A->cigar_n, op, (size_t)n);
  A->cigar_n += n;
}

/* ------------------------------------------------------------------------------------------
 * Backtrace (A.5).  WFA2: wavefront_backtrace_affine with the piggyback packing
 * (offset << 4) | type; the max wins, so ties resolve by type priority.
 * ------------------------------------------------------------------------------------------ */
enum {
  BT_I1_OPEN = 1, BT_I1_EXT = 2, BT_I2_OPEN = 3, BT_I2_EXT = 4,
  BT_D1_OPEN = 5, BT_D1_EXT = 6, BT_D2_OPEN = 7, BT_D2_EXT = 8, BT_M = 9
};
#define BT_SET(off, type) ((((int64_t)(off)) << 4) | (type))
#define BT_TYPE(v) ((int)((v) & 0xF))
#define BT_OFFSET(v) ((wf_offset_t)((v) >> 4))

static int64_t bt_fetch(wf_aligner_t* a, int comp, int score, int k, int add, int type) {
  if (score < 0 || score >= a->nrows) return WF_OFFSET_NULL;
  wavefront_t* w = &a->wf[comp][score];
  if (w->present && w->lo <= k && k <= w->hi) return BT_SET(WF_AT(w, k) + add, type);
  return WF_OFFSET_NULL;
}

static int wf_backtrace_affine(awo_aligner_t* A, wf_aligner_t* a) {
  const int plen = a->plen, tlen = a->tlen;
  if (A->bt_cap < plen + tlen + 2) {
    free(A->bt_buf);
    A->bt_cap = 2 * (plen + tlen + 2);
    A->bt_buf = (uint8_t*)malloc((size_t)A->bt_cap);
    if (!A->bt_buf) return AWO_ERR_INTERNAL;
  }
  uint8_t* ops = A->bt_buf;
  int begin = A->bt_cap - 1; /* ops[begin+1 .. bt_cap-1] hold the CIGAR */
#define BT_PUSH(c) do { if (begin < 0) return AWO_ERR_CAPACITY; ops[begin--] = (uint8_t)(c); } while (0)
  int matrix_type = a->component_end;
  int score = a->end_score;
  int k = a->end_k;
  wf_offset_t offset = a->end_offset;
  int h = WF_H(k, offset), v = WF_V(k, offset);
  while (v > 0 && h > 0 && score > 0) {
    const int mismatch = score - a->x;
    const int gap_open1 = score - a->o1 - a->e1;
    const int gap_extend1 = score - a->e1;
    const int gap_open2 = score - a->o2 - a->e2;
    const int gap_extend2 = score - a->e2;
    int64_t max_all;
    switch (matrix_type) {
      case COMP_D2: {
        const int64_t ext = bt_fetch(a, COMP_D2, gap_extend2, k + 1, 0, BT_D2_EXT);
        const int64_t opn = bt_fetch(a, COMP_M, gap_open2, k + 1, 0, BT_D2_OPEN);
        max_all = MAXI(ext, opn);
        break;
      }
      case COMP_D1: {
        const int64_t ext = bt_fetch(a, COMP_D1, gap_extend1, k + 1, 0, BT_D1_EXT);
        const int64_t opn = bt_fetch(a, COMP_M, gap_open1, k + 1, 0, BT_D1_OPEN);
        max_all = MAXI(ext, opn);
        break;
      }
      case COMP_I2: {
        const int64_t ext = bt_fetch(a, COMP_I2, gap_extend2, k - 1, 1, BT_I2_EXT);
        const int64_t opn = bt_fetch(a, COMP_M, gap_open2, k - 1, 1, BT_I2_OPEN);
        max_all = MAXI(ext, opn);
        break;
      }
      case COMP_I1: {
        const int64_t ext = bt_fetch(a, COMP_I1, gap_extend1, k - 1, 1, BT_I1_EXT);
        const int64_t opn = bt_fetch(a, COMP_M, gap_open1, k - 1, 1, BT_I1_OPEN);
        max_all = MAXI(ext, opn);
        break;
      }
      default: { /* COMP_M */
        const int64_t misms = bt_fetch(a, COMP_M, mismatch, k, 1, BT_M);
        const int64_t ins1_open = bt_fetch(a, COMP_M, gap_open1, k - 1, 1, BT_I1_OPEN);
        const int64_t ins1_ext = bt_fetch(a, COMP_I1, gap_extend1, k - 1, 1, BT_I1_EXT);
        const int64_t max_ins1 = MAXI(ins1_open, ins1_ext);
        const int64_t del1_open = bt_fetch(a, COMP_M, gap_open1, k + 1, 0, BT_D1_OPEN);
        const int64_t del1_ext = bt_fetch(a, COMP_D1, gap_extend1, k + 1, 0, BT_D1_EXT);
        const int64_t max_del1 = MAXI(del1_open, del1_ext);
        if (!a->two_piece) {
          max_all = MAXI(misms, MAXI(max_ins1, max_del1));
          break;
        }
        const int64_t ins2_open = bt_fetch(a, COMP_M, gap_open2, k - 1, 1, BT_I2_OPEN);
        const int64_t ins2_ext = bt_fetch(a, COMP_I2, gap_extend2, k - 1, 1, BT_I2_EXT);
        const int64_t max_ins2 = MAXI(ins2_open, ins2_ext);
        const int64_t del2_open = bt_fetch(a, COMP_M, gap_open2, k + 1, 0, BT_D2_OPEN);
        const int64_t del2_ext = bt_fetch(a, COMP_D2, gap_extend2, k + 1, 0, BT_D2_EXT);
        const int64_t max_del2 = MAXI(del2_open, del2_ext);
        const int64_t max_ins = MAXI(max_ins1, max_ins2);
        const int64_t max_del = MAXI(max_del1, max_del2);
        max_all = MAXI(misms, MAXI(max_ins, max_del));
        break;
      }
    }
    if (matrix_type == COMP_M) { /* traceback matches */
      const wf_offset_t max_offset = BT_OFFSET(max_all);
      const int num_matches = offset - max_offset;
      if (num_matches < 0 || max_all == WF_OFFSET_NULL) return AWO_ERR_INTERNAL;
      for (int i = 0; i < num_matches; ++i) BT_PUSH('M');
      offset = max_offset;
      v = WF_V(k, offset);
      h = WF_H(k, offset);
      if (v <= 0 || h <= 0) break;
    }
    const int bt = BT_TYPE(max_all);
    switch (bt) {
      case BT_M: score = mismatch; matrix_type = COMP_M; break;
      case BT_I1_OPEN: score = gap_open1; matrix_type = COMP_M; break;
      case BT_I1_EXT: score = gap_extend1; matrix_type = COMP_I1; break;
      case BT_I2_OPEN: score = gap_open2; matrix_type = COMP_M; break;
      case BT_I2_EXT: score = gap_extend2; matrix_type = COMP_I2; break;
      case BT_D1_OPEN: score = gap_open1; matrix_type = COMP_M; break;
      case BT_D1_EXT: score = gap_extend1; matrix_type = COMP_D1; break;
      case BT_D2_OPEN: score = gap_open2; matrix_type = COMP_M; break;
      case BT_D2_EXT: score = gap_extend2; matrix_type = COMP_D2; break;
      default: return AWO_ERR_INTERNAL;
    }
    switch (bt) {
      case BT_M: BT_PUSH('X'); --offset; break;
      case BT_I1_OPEN: case BT_I1_EXT: case BT_I2_OPEN: case BT_I2_EXT:
        BT_PUSH('I'); --k; --offset; break;
      default: BT_PUSH('D'); ++k; break;
    }
    v = WF_V(k, offset);
    h = WF_H(k, offset);
  }
  /* account for the beginning of the alignment */
  if (matrix_type == COMP_M) {
    if (v > 0 && h > 0) { /* score == 0: leading run of matches */
      const int num_matches = MINI(v, h);
      for (int i = 0; i < num_matches; ++i) BT_PUSH('M');
      v -= num_matches;
      h -= num_matches;
    }
    while (v > 0) { BT_PUSH('D'); --v; }
    while (h > 0) { BT_PUSH('I'); --h; }
  } else if (v != 0 || h != 0 || score != 0) {
    return AWO_ERR_INTERNAL; /* WFA2: "I?/D?-Beginning backtrace error" */
  }
#undef BT_PUSH
  const int n = A->bt_cap - 1 - begin;
  if (A->cigar_n + n > A->cigar_cap) return AWO_ERR_CAPACITY;
  memcpy(A->cigar + A->cigar_n, ops + begin + 1, (size_t)n); /* cigar_append_forward */
  A->cigar_n += n;
  return AWO_OK;
}

/* WFA2: wavefront_unialign (A.4 main loop: extend(s) -> terminated? -> ++s -> compute(s)).
 * Leaves a->end_* set; returns the penalty or a negative error. */
static int wf_unialign(wf_aligner_t* a, long max_steps) {
  int score = 0;
  for (long step = 0;; ++step) {
    if (wf_extend_end2end(a, score, NULL)) return score;
    if (step > max_steps) return AWO_ERR_INTERNAL;
    ++score;
    if (!wf_compute(a, score)) return AWO_ERR_INTERNAL;
  }
}

static long max_steps_bound(const wf_aligner_t* a) {
  /* any end-to-end alignment costs at most (all gaps): generous loop guard */
  long g = (long)a->o1 + a->o2 + 2L * (a->e1 + a->e2) + a->x;
  return g * ((long)a->plen + a->tlen + 4) + 1024;
}

/* WFA2: wavefront_bialign_base -- plain WFA + backtrace on the subsidiary aligner (A.6) */
static int bialign_base(awo_aligner_t* A, int cb, int ce, int* penalty) {
  wf_aligner_t* s = &A->sub;
  s->stats = A->stats;
  if (!wf_unialign_init(s, cb, ce)) return AWO_ERR_INTERNAL;
  const int score = wf_unialign(s, max_steps_bound(s));
  if (score < 0) return score;
  if (A->stats) A->stats->n_base++;
  if (penalty) *penalty = score;
  if (getenv("AWO_DEBUG")) fprintf(stderr, "[awo]   base score %d end_k %d\n", score, s->end_k);
  return wf_backtrace_affine(A, s);
}

/* ------------------------------------------------------------------------------------------
 * BiWFA breakpoint search (A.6)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int score, score_forward, score_reverse;
  int k_forward, k_reverse;
  wf_offset_t offset_forward, offset_reverse;
  int component;
} bialign_breakpoint_t;

/* WFA2: wavefront_bialign_breakpoint_indel2indel */
static void bialign_breakpoint_indel2indel(wf_aligner_t* a0, bool breakpoint_forward, int score_0, int score_1,
                                           wavefront_t* dwf_0, wavefront_t* dwf_1, int component,
                                           bialign_breakpoint_t* bp) {
  const int tlen = a0->tlen, plen = a0->plen;
  const int gap_open = (component == COMP_I1 || component == COMP_D1) ? a0->o1 : a0->o2;
  const int lo_0 = dwf_0->lo, hi_0 = dwf_0->hi;
  const int lo_1 = WF_K_INVERSE(dwf_1->hi, plen, tlen);
  const int hi_1 = WF_K_INVERSE(dwf_1->lo, plen, tlen);
  if (hi_1 < lo_0 || hi_0 < lo_1) return;
  const int min_hi = MINI(hi_0, hi_1), max_lo = MAXI(lo_0, lo_1);
  if (a0->stats) a0->stats->overlap_rows++;
  for (int k_0 = max_lo; k_0 <= min_hi; ++k_0) {
    const int k_1 = WF_K_INVERSE(k_0, plen, tlen);
    const wf_offset_t doffset_0 = WF_AT(dwf_0, k_0);
    const wf_offset_t doffset_1 = WF_AT(dwf_1, k_1);
    const int dh_0 = WF_H(k_0, doffset_0);
    const int dh_1 = WF_H(k_1, doffset_1);
    if (dh_0 + dh_1 >= tlen && score_0 + score_1 - gap_open < bp->score) {
      if (breakpoint_forward) {
        const int v = WF_V(k_0, dh_0), h = WF_H(k_0, dh_0);
        if (v > plen || h > tlen) continue; /* out-of-bounds coordinates */
        bp->score_forward = score_0;
        bp->score_reverse = score_1;
        bp->k_forward = k_0;
        bp->k_reverse = k_1;
        bp->offset_forward = dh_0;
        bp->offset_reverse = dh_1;
      } else {
        const int v = WF_V(k_1, dh_1), h = WF_H(k_1, dh_1);
        if (v > plen || h > tlen) continue;
        bp->score_forward = score_1;
        bp->score_reverse = score_0;
        bp->k_forward = k_1;
        bp->k_reverse = k_0;
        bp->offset_forward = dh_1;
        bp->offset_reverse = dh_0;
      }
      bp->score = score_0 + score_1 - gap_open;
      bp->component = component;
      return; /* no need to keep searching */
    }
  }
}

/* WFA2: wavefront_bialign_breakpoint_m2m */
static void bialign_breakpoint_m2m(wf_aligner_t* a0, bool breakpoint_forward, int score_0, int score_1,
                                   wavefront_t* mwf_0, wavefront_t* mwf_1, bialign_breakpoint_t* bp) {
  const int tlen = a0->tlen, plen = a0->plen;
  const int lo_0 = mwf_0->lo, hi_0 = mwf_0->hi;
  const int lo_1 = WF_K_INVERSE(mwf_1->hi, plen, tlen);
  const int hi_1 = WF_K_INVERSE(mwf_1->lo, plen, tlen);
  if (hi_1 < lo_0 || hi_0 < lo_1) return;
  const int min_hi = MINI(hi_0, hi_1), max_lo = MAXI(lo_0, lo_1);
  if (a0->stats) a0->stats->overlap_rows++;
  for (int k_0 = max_lo; k_0 <= min_hi; ++k_0) {
    const int k_1 = WF_K_INVERSE(k_0, plen, tlen);
    const wf_offset_t moffset_0 = WF_AT(mwf_0, k_0);
    const wf_offset_t moffset_1 = WF_AT(mwf_1, k_1);
    const int mh_0 = WF_H(k_0, moffset_0);
    const int mh_1 = WF_H(k_1, moffset_1);
    if (mh_0 + mh_1 >= tlen && score_0 + score_1 < bp->score) {
      if (breakpoint_forward) {
        bp->score_forward = score_0;
        bp->score_reverse = score_1;
        bp->k_forward = k_0;
        bp->k_reverse = k_1;
        bp->offset_forward = moffset_0;
        bp->offset_reverse = moffset_1;
      } else {
        bp->score_forward = score_1;
        bp->score_reverse = score_0;
        bp->k_forward = k_1;
        bp->k_reverse = k_0;
        bp->offset_forward = moffset_1;
        bp->offset_reverse = moffset_0;
      }
      bp->score = score_0 + score_1;
      bp->component = COMP_M;
      return;
    }
  }
}

static wavefront_t* present_or_null(wf_aligner_t* a, int comp, int score_mod) {
  wavefront_t* w = &a->wf[comp][score_mod];
  return w->present ? w : NULL;
}

/* WFA2: wavefront_bialign_overlap (A.6 order: per i, D2, I2, D1, I1, then M) */
static void bialign_overlap(wf_aligner_t* a0, wf_aligner_t* a1, int score_0, int score_1,
                            bool breakpoint_forward, bialign_breakpoint_t* bp) {
  const int max_score_scope = a0->max_score_scope;
  const bool p2 = a0->two_piece;
  const int score_mod_0 = score_0 % max_score_scope;
  wavefront_t* mwf_0 = present_or_null(a0, COMP_M, score_mod_0);
  if (mwf_0 == NULL) return;
  wavefront_t* d1wf_0 = present_or_null(a0, COMP_D1, score_mod_0);
  wavefront_t* i1wf_0 = present_or_null(a0, COMP_I1, score_mod_0);
  wavefront_t* d2wf_0 = p2 ? present_or_null(a0, COMP_D2, score_mod_0) : NULL;
  wavefront_t* i2wf_0 = p2 ? present_or_null(a0, COMP_I2, score_mod_0) : NULL;
  for (int i = 0; i < max_score_scope; ++i) {
    const int score_i = score_1 - i;
    if (score_i < 0) break;
    const int score_mod_i = score_i % max_score_scope;
    if (a0->fast_overlap) { /* exact: every in-bounds cell of any component at a score is <= the extended M cell */
      wavefront_t* m1 = present_or_null(a1, COMP_M, score_mod_i);
      if (m1 == NULL) continue; /* null step: nothing to overlap with */
      if (!mwf_0->saw_oob && !m1->saw_oob && mwf_0->max_ak + m1->max_ak < a0->plen + a0->tlen) continue;
    }
    if (p2 && score_0 + score_i - a0->o2 < bp->score) {
      wavefront_t* d2wf_1 = present_or_null(a1, COMP_D2, score_mod_i);
      if (d2wf_0 != NULL && d2wf_1 != NULL)
        bialign_breakpoint_indel2indel(a0, breakpoint_forward, score_0, score_i, d2wf_0, d2wf_1, COMP_D2, bp);
      wavefront_t* i2wf_1 = present_or_null(a1, COMP_I2, score_mod_i);
      if (i2wf_0 != NULL && i2wf_1 != NULL)
        bialign_breakpoint_indel2indel(a0, breakpoint_forward, score_0, score_i, i2wf_0, i2wf_1, COMP_I2, bp);
    }
    if (score_0 + score_i - a0->o1 < bp->score) {
      wavefront_t* d1wf_1 = present_or_null(a1, COMP_D1, score_mod_i);
      if (d1wf_0 != NULL && d1wf_1 != NULL)
        bialign_breakpoint_indel2indel(a0, breakpoint_forward, score_0, score_i, d1wf_0, d1wf_1, COMP_D1, bp);
      wavefront_t* i1wf_1 = present_or_null(a1, COMP_I1, score_mod_i);
      if (i1wf_0 != NULL && i1wf_1 != NULL)
        bialign_breakpoint_indel2indel(a0, breakpoint_forward, score_0, score_i, i1wf_0, i1wf_1, COMP_I1, bp);
    }
    if (score_0 + score_i >= bp->score) continue;
    wavefront_t* mwf_1 = present_or_null(a1, COMP_M, score_mod_i);
    if (mwf_1 != NULL) bialign_breakpoint_m2m(a0, breakpoint_forward, score_0, score_i, mwf_0, mwf_1, bp);
  }
}

/* WFA2: wavefront_bialign_find_breakpoint (A.6).  Returns WF_STATUS_OK with *bp set,
 * WF_STATUS_END_REACHED if one direction finished alone at score 0, or an error. */
/* known_score: the sub-problem's optimal score when a parent's breakpoint handed it down -- its share
 * (forward / reverse score at that breakpoint) minus the gap open of the breakpoint's component, which
 * the child has pre-paid at its begin or end; INT_MAX at the top level.  Only the CPU-baseline mode looks
 * at it: WFA2 only ever replaces the breakpoint by a strictly better one, so the search may stop as soon
 * as that score is reached -- same result (tests/test_oracle.py checks it against the plain search). */
static int bialign_find_breakpoint(awo_aligner_t* A, int cb, int ce, int known_score, bialign_breakpoint_t* bp) {
  wf_aligner_t* f = &A->fwd;
  wf_aligner_t* r = &A->rev;
  f->stats = r->stats = A->stats;
  if (!wf_unialign_init(f, cb, ce) || !wf_unialign_init(r, ce, cb)) return WF_STATUS_ERROR;
  const int plen = f->plen, tlen = f->tlen;
  const int max_antidiagonal = plen + tlen - 1;
  int score_forward = 0, score_reverse = 0, forward_max_ak = 0, reverse_max_ak = 0;
  bp->score = INT_MAX;
  if (wf_extend_end2end(f, score_forward, &forward_max_ak)) return WF_STATUS_END_REACHED;
  if (wf_extend_end2end(r, score_reverse, &reverse_max_ak)) return WF_STATUS_END_REACHED;
  const long max_steps = max_steps_bound(f);
  long steps = 0;
  /* phase 1: advance until the furthest points can collide */
  int max_ak = 0;
  bool last_wf_forward = false;
  while (true) {
    if (forward_max_ak + reverse_max_ak >= max_antidiagonal) break;
    ++score_forward;
    if (!wf_compute(f, score_forward)) return WF_STATUS_ERROR;
    wf_extend_end2end(f, score_forward, &max_ak);
    if (forward_max_ak < max_ak) forward_max_ak = max_ak;
    last_wf_forward = true;
    if (forward_max_ak + reverse_max_ak >= max_antidiagonal) break;
    ++score_reverse;
    if (!wf_compute(r, score_reverse)) return WF_STATUS_ERROR;
    wf_extend_end2end(r, score_reverse, &max_ak);
    if (reverse_max_ak < max_ak) reverse_max_ak = max_ak;
    last_wf_forward = false;
    if (++steps > max_steps) return WF_STATUS_ERROR;
  }
  /* phase 2: advance until no better overlap is possible */
  const int max_score_scope = f->max_score_scope;
  const int gap_opening = f->two_piece ? MAXI(f->o1, f->o2) : f->o1;
  const bool known_optimum = f->fast_overlap && known_score != INT_MAX;
  while (true) {
    if (last_wf_forward) {
      const int min_score_reverse = (score_reverse > max_score_scope - 1) ? score_reverse - (max_score_scope - 1) : 0;
      if (score_forward + min_score_reverse - gap_opening >= bp->score) break;
      bialign_overlap(f, r, score_forward, score_reverse, true, bp);
      if (known_optimum && bp->score == known_score) break;
      ++score_reverse;
      if (!wf_compute(r, score_reverse)) return WF_STATUS_ERROR;
      wf_extend_end2end(r, score_reverse, NULL);
    }
    const int min_score_forward = (score_forward > max_score_scope - 1) ? score_forward - (max_score_scope - 1) : 0;
    if (min_score_forward + score_reverse - gap_opening >= bp->score) break;
    bialign_overlap(r, f, score_reverse, score_forward, false, bp);
    if (known_optimum && bp->score == known_score) break;
    ++score_forward;
    if (!wf_compute(f, score_forward)) return WF_STATUS_ERROR;
    wf_extend_end2end(f, score_forward, NULL);
    if (++steps > max_steps) return WF_STATUS_ERROR;
    last_wf_forward = true;
  }
  if (A->stats) A->stats->n_breakpoints++;
  if (getenv("AWO_DEBUG")) fprintf(stderr, "[awo]   bp score %d sf %d sr %d kf %d off_f %d comp %d (final scores f %d r %d)\n", bp->score, bp->score_forward, bp->score_reverse, bp->k_forward, bp->offset_forward, bp->component, score_forward, score_reverse);
  return WF_STATUS_OK;
}

/* WFA2: wavefront_bialign_alignment (A.6 recursion) */
static int bialign_alignment(awo_aligner_t* A, int pb, int pe, int tb, int te, int cb, int ce,
                             int score_remaining, int known_score, int level, int* penalty) {
  const int plen = pe - pb, tlen = te - tb;
  if (A->stats && (uint32_t)level > A->stats->max_level) A->stats->max_level = (uint32_t)level;
  /* trivial cases */
  if (tlen == 0) {
    cigar_append(A, 'D', plen);
    if (A->stats) A->stats->n_trivial++;
    if (penalty) *penalty = -1; /* not tracked for trivial halves */
    return A->error;
  } else if (plen == 0) {
    cigar_append(A, 'I', tlen);
    if (A->stats) A->stats->n_trivial++;
    if (penalty) *penalty = -1;
    return A->error;
  }
  seqs_set_bounds(A, pb, pe, tb, te);
  if (getenv("AWO_DEBUG")) fprintf(stderr, "[awo] level %d p[%d,%d) t[%d,%d) cb %d ce %d score_remaining %d\n", level, pb, pe, tb, te, cb, ce, score_remaining);
  /* fall back to regular WFA */
  if (score_remaining <= WF_BIALIGN_FALLBACK_MIN_SCORE) return bialign_base(A, cb, ce, penalty);
  bialign_breakpoint_t bp;
  const int st = bialign_find_breakpoint(A, cb, ce, known_score, &bp);
  if (st == WF_STATUS_END_REACHED) return bialign_base(A, cb, ce, penalty); /* wavefront_bialign_exception */
  if (st != WF_STATUS_OK || bp.score == INT_MAX) return AWO_ERR_INTERNAL;
  const int bh = WF_H(bp.k_forward, bp.offset_forward);
  const int bv = WF_V(bp.k_forward, bp.offset_forward);
  if (bh < 0 || bh > tlen || bv < 0 || bv > plen) return AWO_ERR_INTERNAL;
  const int open_c = bp.component == COMP_M ? 0
                     : ((bp.component == COMP_I1 || bp.component == COMP_D1) ? A->pen.gap_open1 : A->pen.gap_open2);
  int rc = bialign_alignment(A, pb, pb + bv, tb, tb + bh, cb, bp.component, bp.score_forward, bp.score_forward - open_c, level + 1, NULL);
  if (rc != AWO_OK) return rc;
  rc = bialign_alignment(A, pb + bv, pe, tb + bh, te, bp.component, ce, bp.score_reverse, bp.score_reverse - open_c, level + 1, NULL);
  if (rc != AWO_OK) return rc;
  if (penalty) *penalty = bp.score;
  return AWO_OK;
}

/* ------------------------------------------------------------------------------------------
 * public API
 * ------------------------------------------------------------------------------------------ */
static int penalties_check(const awo_penalties_t* pen) {
  if (pen->match != 0) return AWO_ERR_PENALTIES; /* A.2: match != 0 is out of scope */
  if (pen->mismatch <= 0 || pen->gap_open1 < 0 || pen->gap_ext1 <= 0) return AWO_ERR_PENALTIES;
  if (pen->two_piece && (pen->gap_open2 < 0 || pen->gap_ext2 <= 0)) return AWO_ERR_PENALTIES;
  return AWO_OK;
}

awo_aligner_t* awo_aligner_new(const awo_penalties_t* pen) {
  if (penalties_check(pen) != AWO_OK) return NULL;
  awo_aligner_t* A = (awo_aligner_t*)calloc(1, sizeof(awo_aligner_t));
  if (!A) return NULL;
  A->pen = *pen;
  aligner_init(&A->fwd, pen, true);
  aligner_init(&A->rev, pen, true);
  aligner_init(&A->sub, pen, false);
  return A;
}

void awo_aligner_set_fast_overlap(awo_aligner_t* A, int on) {
  if (!A) return;
  A->fwd.fast_overlap = A->rev.fast_overlap = on != 0;
}

void awo_aligner_delete(awo_aligner_t* A) {
  if (!A) return;
  aligner_destroy(&A->fwd);
  aligner_destroy(&A->rev);
  aligner_destroy(&A->sub);
  free(A->pbuf); free(A->tbuf); free(A->prbuf); free(A->trbuf); free(A->bt_buf);
  free(A);
}

static int align_prepare(awo_aligner_t* A, const uint8_t* pattern, int plen, const uint8_t* text, int tlen,
                         uint8_t* cigar_out, int cigar_cap, awo_stats_t* stats) {
  if (!A || plen < 0 || tlen < 0 || cigar_cap < plen + tlen) return AWO_ERR_CAPACITY;
  if (!seqs_load(A, pattern, plen, text, tlen)) return AWO_ERR_INTERNAL;
  A->cigar = cigar_out;
  A->cigar_cap = cigar_cap;
  A->cigar_n = 0;
  A->stats = stats;
  A->error = AWO_OK;
  return AWO_OK;
}

int awo_align(awo_aligner_t* A, const uint8_t* pattern, int plen, const uint8_t* text, int tlen,
              uint8_t* cigar_out, int cigar_cap, int* cigar_len, int* penalty, awo_stats_t* stats) {
  int rc = align_prepare(A, pattern, plen, text, tlen, cigar_out, cigar_cap, stats);
  if (rc != AWO_OK) return rc;
  /* WFA2: wavefront_bialign -- short sequences fall back to plain WFA (score_remaining = 0) */
  const bool min_length = MAXI(plen, tlen) <= WF_BIALIGN_FALLBACK_MIN_LENGTH;
  int pen = -1;
  rc = bialign_alignment(A, 0, plen, 0, tlen, COMP_M, COMP_M, min_length ? 0 : INT_MAX, INT_MAX, 0, &pen);
  if (rc != AWO_OK) return rc;
  if (pen < 0) { /* trivial top-level problem: one all-gap run */
    const int n = plen + tlen;
    pen = 0;
    if (n > 0) {
      const int g1 = A->pen.gap_open1 + n * A->pen.gap_ext1;
      const int g2 = A->pen.two_piece ? A->pen.gap_open2 + n * A->pen.gap_ext2 : g1;
      pen = MINI(g1, g2);
    }
  }
  if (cigar_len) *cigar_len = A->cigar_n;
  if (penalty) *penalty = pen;
  return AWO_OK;
}

int awo_align_unidirectional(awo_aligner_t* A, const uint8_t* pattern, int plen, const uint8_t* text, int tlen,
                             uint8_t* cigar_out, int cigar_cap, int* cigar_len, int* penalty,
                             awo_stats_t* stats) {
  int rc = align_prepare(A, pattern, plen, text, tlen, cigar_out, cigar_cap, stats);
  if (rc != AWO_OK) return rc;
  int pen = 0;
  if (tlen == 0 || plen == 0) {
    cigar_append(A, tlen == 0 ? 'D' : 'I', plen + tlen);
    const int n = plen + tlen;
    if (n > 0) {
      const int g1 = A->pen.gap_open1 + n * A->pen.gap_ext1;
      const int g2 = A->pen.two_piece ? A->pen.gap_open2 + n * A->pen.gap_ext2 : g1;
      pen = MINI(g1, g2);
    }
  } else {
    seqs_set_bounds(A, 0, plen, 0, tlen);
    rc = bialign_base(A, COMP_M, COMP_M, &pen);
    if (rc != AWO_OK) return rc;
  }
  if (cigar_len) *cigar_len = A->cigar_n;
  if (penalty) *penalty = pen;
  return AWO_OK;
}
