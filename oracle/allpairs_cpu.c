/*
 * allpairs_cpu.c -- thread-pool all-pairs driver over the CPU restatement.
 *
 * TEST INFRASTRUCTURE / REPORTED CPU BASELINE ONLY (see biwfa_oracle.h): bench.py times this
 * beside the GPU number as cpu_baseline.kind = "port"; it is never the measured product.
 *
 * Mirrors how the reference drives the path: a work-stealing parallel loop over the pair list
 * (/root/reference/src/iterator.rs:222-233) with one cached aligner per worker thread
 * (/root/reference/src/alignment.rs:19-22,210-221), per pair: align (alignment.rs:231), copy
 * the op bytes (:236), count operations (:292-344) and optionally format the PAF record
 * (/root/reference/src/lib.rs:71-112) into a memory sink.
 */
#include "biwfa_oracle.h"

#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
  const uint8_t* seqs;
  const uint64_t* offsets;
  int nseq;
  const int32_t* pairs;
  int64_t npairs;
  const awo_penalties_t* pen;
  awo_pair_result_t* results;
  atomic_llong* cursor;
  int want_paf;
  int fast_overlap;
  /* per-thread outputs */
  awo_stats_t stats;
  uint64_t paf_bytes;
  int failed;
} worker_t;

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* lib.rs:95-111 record layout; cigar_bytes_to_string of alignment.rs:347-376 (RLE, M->'=',
 * I<->D swap).  Returns the number of bytes written (line + '\n'). */
static size_t format_paf(char* out, size_t cap, int qi, int ti, int qlen, int tlen, const uint8_t* cigar, int n,
                         int num_matches, int alignment_length, int q_end, int t_end) {
  const int block_len = t_end > q_end ? t_end : q_end;
  const double identity = alignment_length > 0 ? (double)num_matches / (double)alignment_length : 0.0;
  size_t w = (size_t)snprintf(out, cap, "s%05d\t%d\t0\t%d\t+\ts%05d\t%d\t0\t%d\t%d\t%d\t60\tgi:f:%.6f\tcg:Z:", qi, qlen,
                              q_end, ti, tlen, t_end, num_matches, block_len, identity);
  int i = 0;
  while (i < n && w + 16 < cap) {
    const uint8_t op = cigar[i];
    int j = i;
    while (j < n && cigar[j] == op) ++j;
    const char c = op == 'M' ? '=' : op == 'X' ? 'X' : op == 'I' ? 'D' : op == 'D' ? 'I' : '?';
    w += (size_t)snprintf(out + w, cap - w, "%d%c", j - i, c);
    i = j;
  }
  out[w++] = '\n';
  return w;
}

static void* worker_main(void* arg) {
  worker_t* w = (worker_t*)arg;
  awo_aligner_t* A = awo_aligner_new(w->pen);
  if (!A) { w->failed = 1; return NULL; }
  awo_aligner_set_fast_overlap(A, w->fast_overlap);
  size_t ccap = 1 << 16, pcap = 1 << 17;
  uint8_t* cigar = (uint8_t*)malloc(ccap);
  char* paf = w->want_paf ? (char*)malloc(pcap) : NULL;
  for (;;) {
    const long long i = atomic_fetch_add(w->cursor, 1);
    if (i >= w->npairs) break;
    const int qi = w->pairs[2 * i], ti = w->pairs[2 * i + 1];
    const uint8_t* q = w->seqs + w->offsets[qi];
    const uint8_t* t = w->seqs + w->offsets[ti];
    const int qlen = (int)(w->offsets[qi + 1] - w->offsets[qi]);
    const int tlen = (int)(w->offsets[ti + 1] - w->offsets[ti]);
    if ((size_t)(qlen + tlen) > ccap) {
      ccap = 2 * (size_t)(qlen + tlen);
      cigar = (uint8_t*)realloc(cigar, ccap);
      if (paf) { pcap = 4 * ccap + 256; paf = (char*)realloc(paf, pcap); }
    }
    awo_pair_result_t* r = &w->results[i];
    memset(r, 0, sizeof(*r));
    int n = 0, penalty = 0;
    r->status = awo_align(A, q, qlen, t, tlen, cigar, (int)ccap, &n, &penalty, &w->stats);
    if (r->status != AWO_OK) continue;
    r->penalty = penalty;
    r->cigar_len = n;
    uint64_t hsh = 1469598103934665603ULL;
    int nm = 0, nx = 0, ni = 0, nd = 0;
    for (int c = 0; c < n; ++c) {
      const uint8_t op = cigar[c];
      hsh = (hsh ^ op) * 1099511628211ULL;
      nm += op == 'M';
      nx += op == 'X';
      ni += op == 'I';
      nd += op == 'D';
    }
    r->num_matches = nm;
    r->num_mismatches = nx;
    r->num_ins_text = ni;
    r->num_del_pattern = nd;
    r->cigar_hash = hsh;
    if (paf) /* alignment.rs:320-344: query_end = #M+#X+#D, target_end = #M+#X+#I */
      w->paf_bytes += format_paf(paf, pcap, qi, ti, qlen, tlen, cigar, n, nm, nm + nx, nm + nx + nd, nm + nx + ni);
  }
  free(cigar);
  free(paf);
  awo_aligner_delete(A);
  return NULL;
}

static double all_pairs_impl(const uint8_t* seqs, const uint64_t* offsets, int nseq, const int32_t* pairs, int64_t npairs,
                             const awo_penalties_t* pen, int nthreads, awo_pair_result_t* results, awo_stats_t* stats_total,
                             uint64_t* paf_sink_bytes, int fast_overlap) {
  if (nthreads < 1) nthreads = 1;
  atomic_llong cursor;
  atomic_init(&cursor, 0);
  worker_t* ws = (worker_t*)calloc((size_t)nthreads, sizeof(worker_t));
  pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
  const double t0 = now_s();
  for (int i = 0; i < nthreads; ++i) {
    ws[i].seqs = seqs; ws[i].offsets = offsets; ws[i].nseq = nseq;
    ws[i].pairs = pairs; ws[i].npairs = npairs; ws[i].pen = pen;
    ws[i].results = results; ws[i].cursor = &cursor; ws[i].want_paf = paf_sink_bytes != NULL;
    ws[i].fast_overlap = fast_overlap;
    pthread_create(&th[i], NULL, worker_main, &ws[i]);
  }
  for (int i = 0; i < nthreads; ++i) pthread_join(th[i], NULL);
  const double t1 = now_s();
  if (stats_total) memset(stats_total, 0, sizeof(*stats_total));
  uint64_t paf = 0;
  int failed = 0;
  for (int i = 0; i < nthreads; ++i) {
    failed |= ws[i].failed;
    paf += ws[i].paf_bytes;
    if (stats_total) {
      stats_total->cell_steps += ws[i].stats.cell_steps;
      stats_total->extend_bytes += ws[i].stats.extend_bytes;
      stats_total->n_breakpoints += ws[i].stats.n_breakpoints;
      stats_total->n_base += ws[i].stats.n_base;
      stats_total->n_trivial += ws[i].stats.n_trivial;
      stats_total->overlap_rows += ws[i].stats.overlap_rows;
      if (ws[i].stats.max_level > stats_total->max_level) stats_total->max_level = ws[i].stats.max_level;
      if (ws[i].stats.max_width > stats_total->max_width) stats_total->max_width = ws[i].stats.max_width;
    }
  }
  if (paf_sink_bytes) *paf_sink_bytes = paf;
  free(ws);
  free(th);
  return failed ? -1.0 : t1 - t0;
}

double awo_all_pairs(const uint8_t* seqs, const uint64_t* offsets, int nseq, const int32_t* pairs, int64_t npairs,
                     const awo_penalties_t* pen, int nthreads, awo_pair_result_t* results, awo_stats_t* stats_total,
                     uint64_t* paf_sink_bytes) {
  return all_pairs_impl(seqs, offsets, nseq, pairs, npairs, pen, nthreads, results, stats_total, paf_sink_bytes, 0);
}

double awo_all_pairs_fast(const uint8_t* seqs, const uint64_t* offsets, int nseq, const int32_t* pairs, int64_t npairs,
                          const awo_penalties_t* pen, int nthreads, awo_pair_result_t* results, awo_stats_t* stats_total,
                          uint64_t* paf_sink_bytes) {
  return all_pairs_impl(seqs, offsets, nseq, pairs, npairs, pen, nthreads, results, stats_total, paf_sink_bytes, 1);
}
