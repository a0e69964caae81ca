/*
 * cigar_check.c -- CIGAR validator + re-scorer.
 *
 * TEST INFRASTRUCTURE ONLY (see biwfa_oracle.h).  Restates the checks the reference applies
 * to WFA2 op bytes: full consumption of both sequences with WFA2's I/D meaning
 * (/root/reference/src/wfa.rs:105-176: 'I' consumes the reference/text, 'D' consumes the
 * query/pattern) and that every match column really matches
 * (/root/reference/src/validation_simple.rs:73-161).  Adds: every 'X' column really differs,
 * and the penalty of the op string under the given penalties (a maximal gap run of length L
 * costs min(o1+L*e1, o2+L*e2)).
 */
#include "biwfa_oracle.h"

int awo_cigar_check(const uint8_t* cigar, int n, const uint8_t* pattern, int plen, const uint8_t* text,
                    int tlen, const awo_penalties_t* pen, int64_t* rescored) {
  int q = 0, r = 0;
  int64_t score = 0;
  int i = 0;
  while (i < n) {
    const uint8_t op = cigar[i];
    int j = i;
    while (j < n && cigar[j] == op) ++j;
    const int64_t len = j - i;
    switch (op) {
      case 'M':
        for (int t = 0; t < len; ++t) {
          if (q >= plen || r >= tlen) return -2;
          if (pattern[q] != text[r]) return -3;
          ++q; ++r;
        }
        break;
      case 'X':
        for (int t = 0; t < len; ++t) {
          if (q >= plen || r >= tlen) return -2;
          if (pattern[q] == text[r]) return -4;
          ++q; ++r;
        }
        score += len * pen->mismatch;
        break;
      case 'I':
      case 'D': {
        if (op == 'I') { r += (int)len; if (r > tlen) return -2; }
        else { q += (int)len; if (q > plen) return -2; }
        int64_t g = (int64_t)pen->gap_open1 + len * pen->gap_ext1;
        if (pen->two_piece) {
          const int64_t g2 = (int64_t)pen->gap_open2 + len * pen->gap_ext2;
          if (g2 < g) g = g2;
        }
        score += g;
        break;
      }
      default:
        return -5;
    }
    i = j;
  }
  if (q != plen) return -6;
  if (r != tlen) return -7;
  if (rescored) *rescored = score;
  return 0;
}

/* FNV-1a over op bytes: the hash awo_all_pairs reports per pair (allpairs_cpu.c), exposed so that a
 * test can hash the bytes another implementation produced and compare without a Python byte loop. */
uint64_t awo_fnv1a(const uint8_t* p, int64_t n) {
  uint64_t h = 1469598103934665603ULL;
  for (int64_t i = 0; i < n; ++i) h = (h ^ p[i]) * 1099511628211ULL;
  return h;
}
