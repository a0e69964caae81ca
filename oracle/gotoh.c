/*
 * gotoh.c -- independent full-DP optimum for gap-affine / 2-piece gap-affine global alignment.
 *
 * TEST INFRASTRUCTURE ONLY (see biwfa_oracle.h).  This is the anchor that pins the oracle's
 * penalty: an exact WFA with no heuristic (alignment.rs:228 sets HeuristicStrategy::None)
 * returns the optimum, so WFA2-lib's penalty == this DP's by definition.
 * Gap of length L costs min(o1 + L*e1, o2 + L*e2); match = 0, mismatch = x.
 * O(plen * tlen) time, O(tlen) memory -- for cross-checks at <= a few kbp.
 */
#include "biwfa_oracle.h"

#include <stdlib.h>

#define INF ((int64_t)1 << 50)
static inline int64_t min2(int64_t a, int64_t b) { return a < b ? a : b; }

int64_t awo_gotoh_penalty(const uint8_t* pattern, int plen, const uint8_t* text, int tlen,
                          const awo_penalties_t* pen) {
  const int64_t x = pen->mismatch, o1 = pen->gap_open1, e1 = pen->gap_ext1;
  const int64_t o2 = pen->two_piece ? pen->gap_open2 : INF, e2 = pen->two_piece ? pen->gap_ext2 : 0;
  const size_t n = (size_t)tlen + 1;
  /* rows over j (text); i (pattern) advances row by row */
  int64_t* M = (int64_t*)malloc(5 * n * sizeof(int64_t));
  if (!M) return -1;
  int64_t* I1 = M + n; /* gap consuming text (horizontal) */
  int64_t* I2 = I1 + n;
  int64_t* D1 = I2 + n; /* gap consuming pattern (vertical) */
  int64_t* D2 = D1 + n;
  M[0] = 0;
  I1[0] = I2[0] = D1[0] = D2[0] = INF;
  for (int j = 1; j <= tlen; ++j) {
    I1[j] = min2(M[j - 1] + o1 + e1, I1[j - 1] + e1);
    I2[j] = min2(M[j - 1] + o2 + e2, I2[j - 1] + e2);
    D1[j] = D2[j] = INF;
    M[j] = min2(I1[j], I2[j]);
  }
  for (int i = 1; i <= plen; ++i) {
    int64_t diag = M[0]; /* M[i-1][j-1] */
    D1[0] = min2(M[0] + o1 + e1, D1[0] + e1);
    D2[0] = min2(M[0] + o2 + e2, D2[0] + e2);
    I1[0] = I2[0] = INF;
    M[0] = min2(D1[0], D2[0]);
    const uint8_t pc = pattern[i - 1];
    for (int j = 1; j <= tlen; ++j) {
      const int64_t up = M[j]; /* M[i-1][j] */
      const int64_t d1 = min2(up + o1 + e1, D1[j] + e1);
      const int64_t d2 = min2(up + o2 + e2, D2[j] + e2);
      const int64_t i1 = min2(M[j - 1] + o1 + e1, I1[j - 1] + e1);
      const int64_t i2 = min2(M[j - 1] + o2 + e2, I2[j - 1] + e2);
      int64_t m = diag + (pc == text[j - 1] ? 0 : x);
      m = min2(m, min2(min2(d1, d2), min2(i1, i2)));
      diag = up;
      D1[j] = d1;
      D2[j] = d2;
      I1[j] = i1;
      I2[j] = i2;
      M[j] = m;
    }
  }
  const int64_t r = M[tlen];
  free(M);
  return r;
}
