// Suffix-array construction by induced sorting (SA-IS; Nong, Zhang & Chan, DCC 2009).
//
// Host-side, one-off index construction.  The reference builds its
// suffix array inside sdsl::construct (reference include/psi/fmindex.hpp:257-271), a
// third-party library that is not available here; this is an implementation of the published
// algorithm.  Its decomposition (bucket counts / bucket ends, two induction sweeps, LMS-substring
// naming, recursion on the reduced string) follows the structure of the well-known public
// reference implementation sais-lite (Yuta Mori, 2008-2010, MIT licence) -- not part of the
// reference tree.  Indices are int32 (text length < 2^31).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace psigpu {
namespace sais_detail {

template <typename CharT>
static void get_counts(const CharT* T, int32_t* C, int32_t n, int32_t K)
{
  for (int32_t i = 0; i < K; ++i) C[i] = 0;
  for (int32_t i = 0; i < n; ++i) ++C[T[i]];
}

static inline void get_buckets(const int32_t* C, int32_t* B, int32_t K, bool end)
{
  int32_t sum = 0;
  if (end) for (int32_t i = 0; i < K; ++i) { sum += C[i]; B[i] = sum; }
  else for (int32_t i = 0; i < K; ++i) { sum += C[i]; B[i] = sum - C[i]; }
}

#define PSIGPU_TGET(i) ((t[(i) >> 3] >> ((i) & 7)) & 1)
#define PSIGPU_TSET(i, b) (t[(i) >> 3] = (uint8_t)((b) ? (t[(i) >> 3] | (1u << ((i) & 7))) \
                                                       : (t[(i) >> 3] & ~(1u << ((i) & 7)))))
#define PSIGPU_ISLMS(i) ((i) > 0 && PSIGPU_TGET(i) && !PSIGPU_TGET((i) - 1))

template <typename CharT>
static void induce_l(const uint8_t* t, int32_t* SA, const CharT* T, const int32_t* C,
                     int32_t* B, int32_t n, int32_t K)
{
  get_buckets(C, B, K, false);
  for (int32_t i = 0; i < n; ++i) {
    int32_t j = SA[i] - 1;
    if (j >= 0 && !PSIGPU_TGET(j)) SA[B[T[j]]++] = j;
  }
}

template <typename CharT>
static void induce_s(const uint8_t* t, int32_t* SA, const CharT* T, const int32_t* C,
                     int32_t* B, int32_t n, int32_t K)
{
  get_buckets(C, B, K, true);
  for (int32_t i = n - 1; i >= 0; --i) {
    int32_t j = SA[i] - 1;
    if (j >= 0 && PSIGPU_TGET(j)) SA[--B[T[j]]] = j;
  }
}

// T[n-1] must be the unique smallest symbol (0).
template <typename CharT>
static void sais_main(const CharT* T, int32_t* SA, int32_t n, int32_t K)
{
  std::vector<uint8_t> tv((size_t)n / 8 + 1, 0);
  uint8_t* t = tv.data();
  // S-type = 1, L-type = 0
  PSIGPU_TSET(n - 1, 1);
  if (n >= 2) PSIGPU_TSET(n - 2, 0);
  for (int32_t i = n - 3; i >= 0; --i)
    PSIGPU_TSET(i, (T[i] < T[i + 1] || (T[i] == T[i + 1] && PSIGPU_TGET(i + 1))) ? 1 : 0);

  std::vector<int32_t> Cv((size_t)K), Bv((size_t)K);
  int32_t* C = Cv.data();
  int32_t* B = Bv.data();
  get_counts(T, C, n, K);

  // stage 1: sort LMS substrings
  get_buckets(C, B, K, true);
  for (int32_t i = 0; i < n; ++i) SA[i] = -1;
  for (int32_t i = 1; i < n; ++i)
    if (PSIGPU_ISLMS(i)) SA[--B[T[i]]] = i;
  induce_l(t, SA, T, C, B, n, K);
  induce_s(t, SA, T, C, B, n, K);

  // compact the sorted LMS substrings into SA[0, n1)
  int32_t n1 = 0;
  for (int32_t i = 0; i < n; ++i)
    if (PSIGPU_ISLMS(SA[i])) SA[n1++] = SA[i];
  for (int32_t i = n1; i < n; ++i) SA[i] = -1;

  // name them
  int32_t name = 0, prev = -1;
  for (int32_t i = 0; i < n1; ++i) {
    int32_t pos = SA[i];
    bool diff = false;
    if (prev < 0) diff = true;
    else {
      for (int32_t d = 0; d < n; ++d) {
        if (T[pos + d] != T[prev + d] || PSIGPU_TGET(pos + d) != PSIGPU_TGET(prev + d)) {
          diff = true;
          break;
        }
        if (d > 0 && (PSIGPU_ISLMS(pos + d) || PSIGPU_ISLMS(prev + d))) break;
      }
    }
    if (diff) { ++name; prev = pos; }
    SA[n1 + (pos >> 1)] = name - 1;
  }
  for (int32_t i = n - 1, j = n - 1; i >= n1; --i)
    if (SA[i] >= 0) SA[j--] = SA[i];

  // stage 2: solve the reduced problem
  int32_t* SA1 = SA;
  int32_t* s1 = SA + n - n1;
  if (name < n1) {
    sais_main<int32_t>(s1, SA1, n1, name);
  } else {
    for (int32_t i = 0; i < n1; ++i) SA1[s1[i]] = i;
  }

  // stage 3: induce the result
  get_buckets(C, B, K, true);
  for (int32_t i = 1, j = 0; i < n; ++i)
    if (PSIGPU_ISLMS(i)) s1[j++] = i;           // LMS positions in text order
  for (int32_t i = 0; i < n1; ++i) SA1[i] = s1[SA1[i]];
  for (int32_t i = n1; i < n; ++i) SA[i] = -1;
  for (int32_t i = n1 - 1; i >= 0; --i) {
    int32_t j = SA[i];
    SA[i] = -1;
    SA[--B[T[j]]] = j;
  }
  induce_l(t, SA, T, C, B, n, K);
  induce_s(t, SA, T, C, B, n, K);
}

#undef PSIGPU_TGET
#undef PSIGPU_TSET
#undef PSIGPU_ISLMS

}  // namespace sais_detail

// Suffix array of T[0,n) over [0,K); T[n-1] must be 0 and occur nowhere else.
inline void suffix_array(const uint8_t* T, int32_t* SA, int32_t n, int32_t K)
{
  if (n == 1) { SA[0] = 0; return; }
  sais_detail::sais_main<uint8_t>(T, SA, n, K);
}

}  // namespace psigpu
