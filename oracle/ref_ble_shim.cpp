// Test-infrastructure shim (own code): C symbols over the reference's FastCherries branch-length
// / site-rate estimation (cherryml/phylogeny_estimation/FastCherries/branch_length_estimation.cpp:
// get_branch_lengths :60-103, get_site_rates :105-144, ble :146-241) and its log-transition bank
// (io_helpers.cpp:150-174).  Compiled together with the reference sources WHERE THEY LIE (see
// oracle/Makefile); nothing of the reference is copied.  Cherries arrive as two int arrays
// [n_cherries][L] (state index, -1 = gap / unknown).
#include <string>
#include <utility>
#include <vector>

#include "branch_length_estimation.h"
#include "io_helpers.h"
#include "types.h"

namespace {
typedef std::vector<std::pair<std::vector<int>, std::vector<int>>> cherries_t;

cherries_t make_cherries(const int *cx, const int *cy, int n, int L) {
  cherries_t c(n);
  for (int i = 0; i < n; ++i) {
    c[i].first.assign(cx + (size_t)i * L, cx + (size_t)(i + 1) * L);
    c[i].second.assign(cy + (size_t)i * L, cy + (size_t)(i + 1) * L);
  }
  return c;
}
transition_matrices make_bank(int S, int T, int R, const double *logP) {
  transition_matrices tm(T, R, S, S);
  for (size_t i = 0; i < tm.matrix.size(); ++i) tm.matrix[i] = logP[i];
  return tm;
}
std::vector<std::vector<int>> valid_by_cherry(const cherries_t &c) {
  std::vector<std::vector<int>> v(c.size());
  for (size_t k = 0; k < c.size(); ++k)
    for (size_t i = 0; i < c[k].first.size(); ++i)
      if (c[k].first[i] != -1 && c[k].second[i] != -1) v[k].push_back((int)i);
  return v;
}
std::vector<std::vector<int>> valid_by_site(const cherries_t &c) {
  const size_t L = c.empty() ? 0 : c[0].first.size();
  std::vector<std::vector<int>> v(L);
  for (size_t s = 0; s < L; ++s)
    for (size_t k = 0; k < c.size(); ++k)
      if (c[k].first[s] != -1 && c[k].second[s] != -1) v[s].push_back((int)k);
  return v;
}
}  // namespace

extern "C" int ref_log_bank(const char *matrix_path, int S, const double *grid, int T, const double *rates, int R,
                            double *out) {
  std::vector<double> g(grid, grid + T), r(rates, rates + R);
  transition_matrices tm = read_rate_compute_log_transition_matrices(matrix_path, g, r, S);
  for (size_t i = 0; i < tm.matrix.size(); ++i) out[i] = tm.matrix[i];
  return 0;
}

extern "C" int ref_get_branch_lengths(int S, int T, int R, const double *logP, const int *cx, const int *cy, int n,
                                      int L, const double *grid, const int *site_to_rate, int *out) {
  cherries_t c = make_cherries(cx, cy, n, L);
  std::vector<int> res = get_branch_lengths(c, make_bank(S, T, R, logP), std::vector<double>(grid, grid + T),
                                            std::vector<int>(site_to_rate, site_to_rate + L), valid_by_cherry(c));
  for (int i = 0; i < n; ++i) out[i] = res[i];
  return 0;
}

extern "C" int ref_get_site_rates(int S, int T, int R, const double *logP, const int *cx, const int *cy, int n, int L,
                                  const int *lengths_index, const double *priors, int *out) {
  cherries_t c = make_cherries(cx, cy, n, L);
  std::vector<int> res = get_site_rates(c, make_bank(S, T, R, logP), std::vector<int>(lengths_index, lengths_index + n),
                                        std::vector<double>(priors, priors + R), valid_by_site(c));
  for (int i = 0; i < L; ++i) out[i] = res[i];
  return 0;
}

extern "C" int ref_ble(int S, int T, int R, const double *logP, const int *cx, const int *cy, int n, int L,
                       const int *all_seqs, int n_seqs, const double *grid, const double *rates, const double *weights,
                       int max_iters, double *lengths_out, double *rates_out) {
  cherries_t c = make_cherries(cx, cy, n, L);
  std::vector<std::vector<int>> seqs(n_seqs);
  for (int i = 0; i < n_seqs; ++i) seqs[i].assign(all_seqs + (size_t)i * L, all_seqs + (size_t)(i + 1) * L);
  length_and_rates res = ble(c, seqs, make_bank(S, T, R, logP), std::vector<double>(grid, grid + T),
                             std::vector<double>(rates, rates + R), std::vector<double>(weights, weights + R), max_iters);
  for (int i = 0; i < n; ++i) lengths_out[i] = res.lengths[i];
  for (int i = 0; i < L; ++i) rates_out[i] = res.rates[i];
  return 0;
}
